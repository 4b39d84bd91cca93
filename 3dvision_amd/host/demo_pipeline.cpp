// Demo-mode driver: the reference's demo run (use_camera:false, use_robot:false, visualization none)
// reproduced over the HIP backend's host mirror — the same stage sequence, parameters and console lines
// as Pipeline::run / Pipeline::processInstance (/root/reference/src/pipeline.cpp:183-380, :25-150),
// without the peripherals (camera, SAM, robot, GL viewer) that need libraries this image lacks.
// The reference's pipeline.cpp itself cannot be built here (OpenCV, Eigen, yaml-cpp, GLFW, GLEW, glm,
// librealsense2 are all REQUIRED by its CMakeLists.txt:26-46); with those present a maintainer links the
// unchanged pipeline.cpp against eigen_adapter/*.cpp instead.
//
//   demo_pipeline [voxel_size=0.001] [ransac_max_iterations=100000] [icp_max_iterations=200]
#include "tdv_registration.hpp"

#include <chrono>
#include <vector>
#include <utility>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <optional>

using namespace industry_picking::hip;

struct Config {  // include/pipeline_config.hpp defaults + config/pipeline_config.yaml
    int width = 1280, height = 720;
    float scale_to_meters = 1000.0f, clipping_max = 1.5f;
    float voxel_size = 0.001f;
    int ransac_max_iterations = 100000;
    float ransac_confidence = 0.999f, icp_distance_factor = 0.4f;
    int icp_max_iterations = 200;
    float min_fitness = 0.3f;
    bool use_point_to_plane = true;
    Mat4f camera_extrinsics = Mat4f::Identity();
};

static std::optional<Mat4f> processInstance(const Config& cfg, const Image& mask, const Image& depth, const Image& rgb,
                                            float fx, float fy, float cx, float cy, const PointCloud& ref_cloud,
                                            const FPFHFeatures& ref_features, int instance_id) {
    auto t0 = std::chrono::high_resolution_clock::now();
    // per-stage wall time, printed on one extra line when TDV_DEMO_STAGES is set
    const bool stages = std::getenv("TDV_DEMO_STAGES") != nullptr;
    std::vector<std::pair<const char*, float>> stage_ms;
    auto mark_t = t0;
    auto mark = [&](const char* name) {
        auto now = std::chrono::high_resolution_clock::now();
        stage_ms.emplace_back(name, std::chrono::duration<float, std::milli>(now - mark_t).count());
        mark_t = now;
    };
    std::cout << "\n--- Processing instance " << instance_id << " ---\n";
    try {
        Image scaled_depth = GPUDepth::preprocess(depth, mask, cfg.scale_to_meters);             // pipeline.cpp:43-44
        mark("preprocess");
        size_t nonzero = 0;
        for (size_t i = 0; i < (size_t)scaled_depth.rows * scaled_depth.cols; ++i) nonzero += scaled_depth.ptr<float>()[i] != 0.f;
        if (nonzero == 0) { std::cerr << "Instance " << instance_id << ": empty depth after masking\n"; return std::nullopt; }
        GPUPointCloud::setMaxDepth(cfg.clipping_max);                                             // CPU-branch clipping (pipeline.cpp:71)
        PointCloud pcd = GPUPointCloud::generate(scaled_depth, rgb, fx, fy, cx, cy);              // :65-66
        mark("unproject");
        if (pcd.empty()) { std::cerr << "Instance " << instance_id << ": empty point cloud\n"; return std::nullopt; }
        std::cout << "Instance " << instance_id << ": " << pcd.size() << " points\n";
        PointCloud source_down = Registration::voxelDownsample(pcd, cfg.voxel_size);              // :92
        mark("voxel");
        Registration::estimateNormals(source_down, 30);                                           // :93
        mark("normals");
        FPFHFeatures source_features = Registration::computeFPFH(source_down, cfg.voxel_size * 5.0f);  // :94-95
        mark("fpfh");
        RegistrationResult coarse = Registration::ransacRegistration(source_down, ref_cloud, source_features, ref_features,
                                                                     cfg.voxel_size, cfg.ransac_max_iterations, cfg.ransac_confidence);
        mark("ransac");
        float icp_threshold = cfg.voxel_size * cfg.icp_distance_factor;                           // :104
        RegistrationResult refined;
        if (GPURegistration::isCudaAvailable()) {
            try {
                refined = GPURegistration::icpRefine(source_down, ref_cloud, coarse.transformation, icp_threshold, cfg.icp_max_iterations);
            } catch (...) {
                throw;  // the reference falls back to its CPU ICP here (:114-120); this backend has no CPU path
            }
        }
        mark("icp");
        if (refined.fitness < cfg.min_fitness) std::cerr << "Instance " << instance_id << ": low fitness " << refined.fitness << "\n";
        Mat4f T_world_object = composePose(cfg.camera_extrinsics, refined.transformation);        // :136-137
        auto t1 = std::chrono::high_resolution_clock::now();
        float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
        std::cout << "Instance " << instance_id << " done in " << ms << " ms (fitness=" << refined.fitness << ")\n";
        if (stages) {
            std::cout << "Stages [ms]:";
            for (auto& st : stage_ms) std::cout << " " << st.first << "=" << st.second;
            std::cout << "\n";
        }
        return T_world_object;
    } catch (const std::exception& e) {
        std::cerr << "Instance " << instance_id << " error: " << e.what() << "\n";
        return std::nullopt;
    }
}

int main(int argc, char** argv) {
    Config cfg;
    if (argc > 1) cfg.voxel_size = (float)std::atof(argv[1]);
    if (argc > 2) cfg.ransac_max_iterations = std::atoi(argv[2]);
    if (argc > 3) cfg.icp_max_iterations = std::atoi(argv[3]);
    const float ext[16] = {0.00705456f, 0.99996948f, -0.00335601f, 0.43244419f, 0.99984465f, -0.00710781f, -0.01612942f, -0.03129219f,
                           -0.01615278f, -0.0032417f, -0.99986428f, 0.39502932f, 0.f, 0.f, 0.f, 1.f};  // config/pipeline_config.yaml:41-57 (row-major)
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) cfg.camera_extrinsics(r, c) = ext[r * 4 + c];
    auto pipeline_start = std::chrono::high_resolution_clock::now();
    std::cout << "\n=== Starting Pipeline ===\n\n[1/5] Using dummy data...\nGenerating procedural test scene...\n";
    const int w = cfg.width, h = cfg.height;
    Image rgb = Image::create(h, w, 3, 1), depth = Image::create(h, w, 1, 2);
    float fx = 900, fy = 900, cx = w / 2.0f, cy = h / 2.0f;
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) {  // pipeline.cpp:223-239
            uint8_t* px = rgb.ptr<uint8_t>() + ((size_t)v * w + u) * 3;
            px[0] = px[1] = px[2] = 50;
            float z = 1.0f;
            if (std::abs(u - cx) < 100 && std::abs(v - cy) < 100) { z = 0.8f; px[0] = 0; px[1] = 0; px[2] = 255; }
            else if (((u / 50) + (v / 50)) % 2 == 0) { px[0] = px[1] = px[2] = 200; }
            depth.ptr<uint16_t>()[(size_t)v * w + u] = static_cast<unsigned short>(z * cfg.scale_to_meters);
        }
    std::cout << "\n[2/5] Segmentation...\nGenerating dummy mask for box...\n";
    Image mask = Image::create(h, w, 1, 1);
    for (int v = h / 2 - 100; v <= h / 2 + 100; ++v)
        for (int u = w / 2 - 100; u <= w / 2 + 100; ++u) mask.ptr<uint8_t>()[(size_t)v * w + u] = 255;  // pipeline.cpp:251-257
    std::cout << "Found 1 masks\n\n[3/5] Loading reference model...\nGenerating dummy reference model...\n";
    PointCloud ref_cloud;
    for (float x = -0.1f; x <= 0.1f; x += 0.005f)
        for (float y = -0.1f; y <= 0.1f; y += 0.005f) { ref_cloud.points.emplace_back(x, y, 0.0f); ref_cloud.normals.emplace_back(0.f, 0.f, 1.f); }
    if (!GPUDepth::isCudaAvailable()) { std::cerr << "No HIP device: this backend has no CPU path.\n"; return 2; }
    PointCloud ref_down = Registration::voxelDownsample(ref_cloud, cfg.voxel_size);               // pipeline.cpp:291
    Registration::estimateNormals(ref_down, 30);
    FPFHFeatures ref_features = Registration::computeFPFH(ref_down, cfg.voxel_size * 5.0f);
    std::cout << "\n[4/5] Processing 1 instances (parallel)...\n";
    auto proc_start = std::chrono::high_resolution_clock::now();
    auto result = processInstance(cfg, mask, depth, rgb, fx, fy, cx, cy, ref_down, ref_features, 0);
    float proc_ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - proc_start).count();
    std::cout << "\nAll instances processed in " << proc_ms << " ms\n";
    std::cout << "Filtered: " << (result ? 1 : 0) << " \xe2\x86\x92 " << (result ? 1 : 0) << " waypoints\n";
    std::cout << "\n[5/5] Robot execution skipped (use_robot=false)\nComputed " << (result ? 1 : 0) << " pick poses.\n";
    if (result) {
        std::cout << "T_world_object =\n";
        for (int r = 0; r < 4; ++r) { for (int c = 0; c < 4; ++c) std::cout << (*result)(r, c) << (c < 3 ? " " : "\n"); }
    }
    float total_ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - pipeline_start).count();
    std::cout << "\n=== Pipeline complete: " << total_ms << " ms ===\n";
    return result ? 0 : 1;
}
