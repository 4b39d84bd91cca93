// Implementation of tdv_registration.hpp on the C ABI.  One tdv_ctx per host thread (the reference
// calls these statics from up to 8 pool threads, include/thread_pool.hpp:17-33); ABI status codes
// become std::runtime_error so the caller's try/catch structure keeps working
// (src/pipeline.cpp:108-121 falls back to the CPU ICP on any exception from the GPU ICP;
// :146-149 turns any std::exception into "instance skipped").
#include "tdv_registration.hpp"
#include <algorithm>
#include <filesystem>
#include "tdv_hip.h"

#include <cstring>
#include <iostream>
#include <stdexcept>

namespace industry_picking {
namespace hip {

namespace {
struct ThreadCtx {
    tdv_ctx* ctx = nullptr;
    float zmax = 10.0f;
    ~ThreadCtx() { if (ctx) tdv_ctx_destroy(ctx); }
};
thread_local ThreadCtx g_tls;

bool device_available() {
    int n = 0;
    return tdv_device_count(&n) == TDV_OK && n > 0;
}
tdv_ctx* ctx_or_throw() {
    if (!g_tls.ctx) {
        if (!device_available()) throw std::runtime_error("CUDA not available");  // src/gpu_impl.cpp:64,258
        int st = tdv_ctx_create(0, &g_tls.ctx);
        if (st != TDV_OK) throw std::runtime_error(std::string("tdv_ctx_create: ") + tdv_status_string(st));
    }
    return g_tls.ctx;
}
void check(int st, const char* what) {
    if (st != TDV_OK)
        throw std::runtime_error(std::string(what) + ": " + tdv_status_string(st) + " " + tdv_last_error(g_tls.ctx));
}
const float* fp(const std::vector<Vec3f>& v) { return v.empty() ? nullptr : v[0].v; }
float* fp(std::vector<Vec3f>& v) { return v.empty() ? nullptr : v[0].v; }
}  // namespace

bool GPUDepth::isCudaAvailable() { return device_available(); }
bool GPURegistration::isCudaAvailable() { return device_available(); }
void GPUPointCloud::setMaxDepth(float zmax) { g_tls.zmax = zmax; }

Image GPUDepth::preprocess(const Image& raw_depth, const Image& mask, float scale) {
    tdv_ctx* c = ctx_or_throw();
    Image out = Image::create(raw_depth.rows, raw_depth.cols, 1, 4);
    // mask semantics follow the CPU branch (threshold > 10, src/pipeline.cpp:51-52), the parity oracle
    check(tdv_depth_preprocess(c, raw_depth.ptr<uint16_t>(), mask.empty() ? nullptr : mask.ptr<uint8_t>(), raw_depth.cols,
                               raw_depth.rows, scale, TDV_MASK_THRESHOLD10, out.ptr<float>()), "GPUDepth::preprocess");
    return out;
}

PointCloud GPUPointCloud::generate(const Image& depth, const Image& rgb, float fx, float fy, float cx, float cy) {
    if (!device_available()) return {};  // src/gpu_impl.cpp:126
    tdv_ctx* c = ctx_or_throw();
    PointCloud pcd;
    // exact output size from one pass over the host image (the test of src/pipeline.cpp:71), so that the by-value result
    // is not a 22 MB zero-filled worst-case buffer
    int cap = 0;
    {
        const float* z = depth.ptr<float>();
        const size_t px = (size_t)depth.rows * depth.cols;
        for (size_t i = 0; i < px; ++i) cap += (z[i] > 0.f && z[i] <= g_tls.zmax) ? 1 : 0;
    }
    if (cap == 0) return pcd;
    pcd.points.resize(cap);
    if (!rgb.empty()) pcd.colors.resize(cap);
    int n = 0;
    check(tdv_deproject(c, depth.ptr<float>(), rgb.empty() ? nullptr : rgb.ptr<uint8_t>(), depth.cols, depth.rows, fx, fy, cx, cy,
                        g_tls.zmax, fp(pcd.points), rgb.empty() ? nullptr : fp(pcd.colors), cap, &n), "GPUPointCloud::generate");
    pcd.points.resize(n);
    if (!rgb.empty()) pcd.colors.resize(n);
    return pcd;
}

RegistrationResult GPURegistration::icpRefine(const PointCloud& source, const PointCloud& target, const Mat4f& T0,
                                              float distance_threshold, int max_iterations) {
    // the reference GPU entry point has no mode flag (gpu_registration.hpp:10-16): point-to-plane
    // when the target carries normals, otherwise Kabsch — Registration::icpRefine's default
    return Registration::icpRefine(source, target, T0, distance_threshold, max_iterations, true);
}

PointCloud Registration::voxelDownsample(const PointCloud& cloud, float voxel_size) {
    tdv_ctx* c = ctx_or_throw();
    PointCloud out;
    const int n = (int)cloud.size();
    out.points.resize(n);
    const bool col = cloud.hasColors() && n > 0;
    if (col) out.colors.resize(n);
    int m = 0;
    check(tdv_voxel_downsample(c, fp(cloud.points), col ? fp(cloud.colors) : nullptr, n, voxel_size, TDV_VOXEL_ORDER_REFERENCE,
                               fp(out.points), col ? fp(out.colors) : nullptr, n, &m), "Registration::voxelDownsample");
    out.points.resize(m);
    if (col) out.colors.resize(m);
    std::cout << "Voxel downsample: " << cloud.size() << " \xe2\x86\x92 " << out.size() << " points\n";  // registration.cpp:58
    return out;
}

void Registration::estimateNormals(PointCloud& cloud, int k) {
    tdv_ctx* c = ctx_or_throw();
    cloud.normals.resize(cloud.size());
    check(tdv_estimate_normals(c, fp(cloud.points), (int)cloud.size(), k, fp(cloud.normals), nullptr), "Registration::estimateNormals");
    std::cout << "Estimated normals for " << cloud.size() << " points\n";  // registration.cpp:129
}

FPFHFeatures Registration::computeFPFH(const PointCloud& cloud, float radius) {
    tdv_ctx* c = ctx_or_throw();
    FPFHFeatures f;
    f.descriptors.resize(cloud.size());
    check(tdv_compute_fpfh(c, fp(cloud.points), fp(cloud.normals), (int)cloud.size(), radius,
                           f.descriptors.empty() ? nullptr : f.descriptors[0].data(), nullptr, nullptr), "Registration::computeFPFH");
    std::cout << "Computed FPFH features for " << cloud.size() << " points\n";  // registration.cpp:199
    return f;
}

RegistrationResult Registration::ransacRegistration(const PointCloud& source, const PointCloud& target, const FPFHFeatures& sf,
                                                    const FPFHFeatures& tf, float voxel_size, int max_iterations, float confidence) {
    tdv_ctx* c = ctx_or_throw();
    std::cout << "RANSAC registration (threshold=" << voxel_size * 1.5f << ", max_iter=" << max_iterations << ")\n";  // :214
    tdv_ransac_result r;
    check(tdv_ransac(c, fp(source.points), (int)source.size(), fp(target.points), (int)target.size(),
                     sf.descriptors.empty() ? nullptr : sf.descriptors[0].data(), tf.descriptors.empty() ? nullptr : tf.descriptors[0].data(),
                     nullptr, voxel_size, max_iterations, confidence, 42u, &r, nullptr), "Registration::ransacRegistration");
    RegistrationResult out;
    std::memcpy(out.transformation.m, r.T, 64);
    out.fitness = r.fitness; out.rmse = r.rmse;
    std::cout << "RANSAC result: fitness=" << out.fitness << ", RMSE=" << out.rmse << "\n";  // :293
    return out;
}

RegistrationResult Registration::icpRefine(const PointCloud& source, const PointCloud& target, const Mat4f& T0, float distance_threshold,
                                           int max_iterations, bool point_to_plane) {
    tdv_ctx* c = ctx_or_throw();
    std::cout << "ICP refinement (threshold=" << distance_threshold << ", max_iter=" << max_iterations << ", mode="
              << (point_to_plane ? "point-to-plane" : "point-to-point") << ")\n";  // :305-307
    tdv_icp_result r;
    check(tdv_icp(c, fp(source.points), (int)source.size(), fp(target.points), target.hasNormals() ? fp(target.normals) : nullptr,
                  (int)target.size(), T0.data(), distance_threshold, max_iterations, point_to_plane ? 1 : 0, &r), "Registration::icpRefine");
    RegistrationResult out;
    std::memcpy(out.transformation.m, r.T, 64);
    out.fitness = r.fitness; out.rmse = r.rmse;
    std::cout << "ICP result: fitness=" << out.fitness << ", RMSE=" << out.rmse << "\n";  // :412
    return out;
}

PointCloud Registration::loadReferenceModel(const std::string& path) {   // src/registration.cpp:416-461
    PointCloud cloud;
    int n = 0, has_color = 0;
    if (tdv_load_ply_ascii(path.c_str(), nullptr, nullptr, 0, &n, &has_color) != TDV_OK) {   // count query; fails only if the file cannot be opened
        std::cerr << "Cannot open reference model: " << path << "\n";                       // :420-423
        return cloud;
    }
    cloud.points.resize(std::max(n, 0));
    if (has_color) cloud.colors.resize(std::max(n, 0));
    if (n > 0 && tdv_load_ply_ascii(path.c_str(), fp(cloud.points), has_color ? fp(cloud.colors) : nullptr, n, &n, &has_color) != TDV_OK)
        throw std::runtime_error("Registration::loadReferenceModel: " + path + " changed while it was read");
    std::cout << "Loaded reference model: " << cloud.size() << " points from " << path << "\n";   // :459
    return cloud;
}

std::vector<Image> Segmentation::loadMasksFromDir(const std::string& masks_dir) {  // src/segmentation.cpp:12-42
    namespace fs = std::filesystem;
    std::vector<Image> masks;
    std::error_code ec;
    if (!fs::is_directory(masks_dir, ec)) {
        std::cerr << "Mask directory not found: " << masks_dir << "\n";
        return masks;
    }
    std::vector<fs::path> files;
    for (const auto& entry : fs::directory_iterator(masks_dir, ec)) {
        std::string ext = entry.path().extension().string();
        std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
        if (ext == ".png" || ext == ".jpg" || ext == ".jpeg") files.push_back(entry.path());
    }
    std::sort(files.begin(), files.end());
    for (const auto& file : files) {
        int w = 0, h = 0;
        if (tdv_load_mask_png(file.string().c_str(), nullptr, 0, &w, &h) != TDV_OK || w <= 0 || h <= 0) continue;  // like an empty imread
        Image m = Image::create(h, w, 1, 1);
        if (tdv_load_mask_png(file.string().c_str(), m.ptr<uint8_t>(), (long long)w * h, &w, &h) != TDV_OK) continue;
        masks.push_back(std::move(m));
    }
    std::cout << "Loaded " << masks.size() << " masks from " << masks_dir << "\n";
    return masks;
}

Mat4f composePose(const Mat4f& extrinsics, const Mat4f& refined) {
    Mat4f out;
    if (tdv_pose_compose(extrinsics.data(), refined.data(), out.data()) != TDV_OK) throw std::runtime_error("composePose: singular transform");
    return out;
}

}  // namespace hip
}  // namespace industry_picking
