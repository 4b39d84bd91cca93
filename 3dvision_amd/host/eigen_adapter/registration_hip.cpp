// Link-time substitute for the reference's src/registration.cpp: defines the Registration:: statics
// declared in include/registration.hpp:32-60 on top of the HIP backend, so that the unchanged
// src/pipeline.cpp (which calls them directly, :92-102, :291-294) runs the non-ICP stages on the GPU too.
// Link EITHER src/registration.cpp OR this file: every static of the class is defined here, including
// Registration::loadReferenceModel (registration.hpp:59, called at src/pipeline.cpp:284) on tdv_load_ply_ascii.
// Not BUILT in this repository (no Eigen in the image); PARSED against include/registration.hpp by tests/test_adapter_syntax.py
// (g++ -fsyntax-only -Werror with stand-in Eigen declarations); mirrors ../tdv_registration.cpp, which runs on the GPU box.
#include "registration.hpp"
#include "tdv_hip.h"

#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>

namespace industry_picking {

namespace {
struct ThreadCtx { tdv_ctx* ctx = nullptr; ~ThreadCtx() { if (ctx) tdv_ctx_destroy(ctx); } };
thread_local ThreadCtx g_tls;
tdv_ctx* ctx_or_throw() {
    if (!g_tls.ctx) {
        int n = 0;
        if (tdv_device_count(&n) != TDV_OK || n <= 0) throw std::runtime_error("HIP device not available");
        if (tdv_ctx_create(0, &g_tls.ctx) != TDV_OK) throw std::runtime_error("tdv_ctx_create failed");
    }
    return g_tls.ctx;
}
void check(int st, const char* what) {
    if (st != TDV_OK) throw std::runtime_error(std::string(what) + ": " + tdv_status_string(st) + " " + tdv_last_error(g_tls.ctx));
}
const float* fp(const std::vector<Eigen::Vector3f>& v) { return v.empty() ? nullptr : v[0].data(); }
float* fp(std::vector<Eigen::Vector3f>& v) { return v.empty() ? nullptr : v[0].data(); }
}  // namespace

PointCloud Registration::voxelDownsample(const PointCloud& cloud, float voxel_size) {
    tdv_ctx* c = ctx_or_throw();
    PointCloud out;
    const int n = (int)cloud.size();
    const bool col = cloud.hasColors() && n > 0;
    out.points.resize(n);
    if (col) out.colors.resize(n);
    int m = 0;
    check(tdv_voxel_downsample(c, fp(cloud.points), col ? fp(cloud.colors) : nullptr, n, voxel_size, TDV_VOXEL_ORDER_REFERENCE,
                               fp(out.points), col ? fp(out.colors) : nullptr, n, &m), "Registration::voxelDownsample");
    out.points.resize(m);
    if (col) out.colors.resize(m);
    std::cout << "Voxel downsample: " << cloud.size() << " \xe2\x86\x92 " << out.size() << " points\n";
    return out;
}

void Registration::estimateNormals(PointCloud& cloud, int k) {
    tdv_ctx* c = ctx_or_throw();
    cloud.normals.resize(cloud.size());
    check(tdv_estimate_normals(c, fp(cloud.points), (int)cloud.size(), k, fp(cloud.normals), nullptr), "Registration::estimateNormals");
    std::cout << "Estimated normals for " << cloud.size() << " points\n";
}

FPFHFeatures Registration::computeFPFH(const PointCloud& cloud, float radius) {
    tdv_ctx* c = ctx_or_throw();
    FPFHFeatures f;
    f.descriptors.resize(cloud.size());
    check(tdv_compute_fpfh(c, fp(cloud.points), fp(cloud.normals), (int)cloud.size(), radius,
                           f.descriptors.empty() ? nullptr : f.descriptors[0].data(), nullptr, nullptr), "Registration::computeFPFH");
    std::cout << "Computed FPFH features for " << cloud.size() << " points\n";
    return f;
}

RegistrationResult Registration::ransacRegistration(const PointCloud& source, const PointCloud& target,
                                                    const FPFHFeatures& source_features, const FPFHFeatures& target_features,
                                                    float voxel_size, int max_iterations, float confidence) {
    tdv_ctx* c = ctx_or_throw();
    std::cout << "RANSAC registration (threshold=" << voxel_size * 1.5f << ", max_iter=" << max_iterations << ")\n";
    tdv_ransac_result r;
    check(tdv_ransac(c, fp(source.points), (int)source.size(), fp(target.points), (int)target.size(),
                     source_features.descriptors.empty() ? nullptr : source_features.descriptors[0].data(),
                     target_features.descriptors.empty() ? nullptr : target_features.descriptors[0].data(),
                     nullptr, voxel_size, max_iterations, confidence, 42u, &r, nullptr), "Registration::ransacRegistration");
    RegistrationResult out;
    std::memcpy(out.transformation.data(), r.T, 64);
    out.fitness = r.fitness; out.rmse = r.rmse;
    std::cout << "RANSAC result: fitness=" << out.fitness << ", RMSE=" << out.rmse << "\n";
    return out;
}

RegistrationResult Registration::icpRefine(const PointCloud& source, const PointCloud& target, const Eigen::Matrix4f& initial_transform,
                                           float distance_threshold, int max_iterations, bool point_to_plane) {
    tdv_ctx* c = ctx_or_throw();
    std::cout << "ICP refinement (threshold=" << distance_threshold << ", max_iter=" << max_iterations << ", mode="
              << (point_to_plane ? "point-to-plane" : "point-to-point") << ")\n";
    tdv_icp_result r;
    check(tdv_icp(c, fp(source.points), (int)source.size(), fp(target.points), target.hasNormals() ? fp(target.normals) : nullptr,
                  (int)target.size(), initial_transform.data(), distance_threshold, max_iterations, point_to_plane ? 1 : 0, &r),
          "Registration::icpRefine");
    RegistrationResult out;
    std::memcpy(out.transformation.data(), r.T, 64);
    out.fitness = r.fitness; out.rmse = r.rmse;
    std::cout << "ICP result: fitness=" << out.fitness << ", RMSE=" << out.rmse << "\n";
    return out;
}

PointCloud Registration::loadReferenceModel(const std::string& path) {   // src/registration.cpp:416-461
    PointCloud cloud;
    int n = 0, has_color = 0;
    if (tdv_load_ply_ascii(path.c_str(), nullptr, nullptr, 0, &n, &has_color) != TDV_OK) {   // count query; fails only if the file cannot be opened
        std::cerr << "Cannot open reference model: " << path << "\n";
        return cloud;
    }
    cloud.points.resize(n > 0 ? n : 0);
    if (has_color) cloud.colors.resize(n > 0 ? n : 0);
    if (n > 0 && tdv_load_ply_ascii(path.c_str(), fp(cloud.points), has_color ? fp(cloud.colors) : nullptr, n, &n, &has_color) != TDV_OK)
        throw std::runtime_error("Registration::loadReferenceModel: " + path + " changed while it was read");
    std::cout << "Loaded reference model: " << cloud.size() << " points from " << path << "\n";
    return cloud;
}

}  // namespace industry_picking
