// Host-side C++ mirror of the reference's operator API for the registration hot path, served by the
// HIP backend through the C ABI (include/tdv_hip.h).  Same class names, method names, argument
// order/meaning and error behaviour as
//   /root/reference/include/registration.hpp:10-60      PointCloud, FPFHFeatures, RegistrationResult, Registration
//   /root/reference/include/gpu_registration.hpp:8-19   GPURegistration
//   /root/reference/include/gpu_depth.hpp:9-22          GPUDepth, GPUPointCloud
// but free of Eigen and OpenCV (neither exists in this image): Vec3f / Mat4f are plain structs that
// are layout-compatible with Eigen::Vector3f / Eigen::Matrix4f (column-major), and Image stands in for
// a continuous cv::Mat.  The TUs in eigen_adapter/ bind the very same entry points to the reference's
// real types for a build inside the reference tree.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace industry_picking {
namespace hip {

struct Vec3f {
    float v[3];
    Vec3f() : v{0, 0, 0} {}
    Vec3f(float x, float y, float z) : v{x, y, z} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};
static_assert(sizeof(Vec3f) == 12, "Vec3f must match Eigen::Vector3f");

struct Mat4f {  // column-major, like Eigen::Matrix4f
    float m[16];
    static Mat4f Identity() { Mat4f a; for (int i = 0; i < 16; ++i) a.m[i] = (i % 5 == 0) ? 1.f : 0.f; return a; }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    const float* data() const { return m; }
    float* data() { return m; }
};

struct PointCloud {  // registration.hpp:10-19
    std::vector<Vec3f> points, normals, colors;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    bool hasNormals() const { return normals.size() == points.size(); }
    bool hasColors() const { return colors.size() == points.size(); }
};

struct FPFHFeatures {  // registration.hpp:21-24
    std::vector<std::array<float, 33>> descriptors;
    size_t size() const { return descriptors.size(); }
};

struct RegistrationResult {  // registration.hpp:26-30
    Mat4f transformation = Mat4f::Identity();
    float fitness = 0.0f;
    float rmse = 0.0f;
};

// Continuous single- or three-channel image (what the reference reads through cv::Mat::data).
struct Image {
    int rows = 0, cols = 0, channels = 1, elem_size = 1;  // elem_size per channel: 1 (u8), 2 (u16), 4 (f32)
    std::vector<uint8_t> bytes;
    bool empty() const { return rows == 0 || cols == 0; }
    template <class T> T* ptr() { return reinterpret_cast<T*>(bytes.data()); }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(bytes.data()); }
    static Image create(int rows, int cols, int channels, int elem_size) {
        Image im; im.rows = rows; im.cols = cols; im.channels = channels; im.elem_size = elem_size;
        im.bytes.assign((size_t)rows * cols * channels * elem_size, 0);
        return im;
    }
};

class Registration {  // registration.hpp:32-60
public:
    static PointCloud voxelDownsample(const PointCloud& cloud, float voxel_size);
    static void estimateNormals(PointCloud& cloud, int k = 30);
    static FPFHFeatures computeFPFH(const PointCloud& cloud, float radius);
    static RegistrationResult ransacRegistration(const PointCloud& source, const PointCloud& target,
                                                 const FPFHFeatures& source_features, const FPFHFeatures& target_features,
                                                 float voxel_size, int max_iterations = 100000, float confidence = 0.999f);
    static RegistrationResult icpRefine(const PointCloud& source, const PointCloud& target, const Mat4f& initial_transform,
                                        float distance_threshold, int max_iterations = 200, bool point_to_plane = true);
    // registration.hpp:59 / src/registration.cpp:416-461 (ASCII PLY, incl. the skipped first vertex); caller: src/pipeline.cpp:284
    static PointCloud loadReferenceModel(const std::string& path);
};

class GPURegistration {  // gpu_registration.hpp:8-19
public:
    static RegistrationResult icpRefine(const PointCloud& source, const PointCloud& target, const Mat4f& initial_transform,
                                        float distance_threshold, int max_iterations = 200);
    static bool isCudaAvailable();
};

class GPUDepth {  // gpu_depth.hpp:9-13
public:
    static Image preprocess(const Image& raw_depth /*u16*/, const Image& mask /*u8 or empty*/, float scale);
    static bool isCudaAvailable();
};

class GPUPointCloud {  // gpu_depth.hpp:15-22
public:
    static PointCloud generate(const Image& depth /*f32*/, const Image& rgb /*u8x3 BGR or empty*/, float fx, float fy, float cx, float cy);
    // the reference dispatch hard-codes max_depth = 10 (src/gpu_impl.cpp:97); the CPU branch uses
    // config.depth.clipping_max (src/pipeline.cpp:71).  Settable per thread; default 10.
    static void setMaxDepth(float zmax);
};

class Segmentation {  // segmentation.hpp:10-13 (the mask-directory source; the SAM client is out of scope)
public:
    // every .png/.jpg/.jpeg of the directory in sorted order as a binary u8 mask (> 10 -> 255); files the backend cannot
    // decode (colour / palette PNG, JPEG) are skipped like unreadable files are in the reference
    static std::vector<Image> loadMasksFromDir(const std::string& masks_dir);
};

// Pose composition of src/pipeline.cpp:136-137: extrinsics * T^-1.
Mat4f composePose(const Mat4f& camera_extrinsics, const Mat4f& refined);

}  // namespace hip
}  // namespace industry_picking
