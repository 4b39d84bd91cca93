// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under 3dvision_amd/ (the product)
// may include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use the oracle, and only as the checker / the
// reported CPU baseline — never as the thing measured or shipped.
//
// CPU restatement of the reference's hot path, Eigen/OpenCV-free, flat arrays:
//   /root/reference/src/registration.cpp:15-60    voxelDownsample (+ key/hash)
//   /root/reference/src/registration.cpp:63-130   findKNN + estimateNormals
//   /root/reference/src/registration.cpp:83-102,133-201  findRadiusNN + computeFPFH
//   /root/reference/src/registration.cpp:204-295  ransacRegistration
//   /root/reference/src/registration.cpp:297-414  icpRefine
//   /root/reference/src/pipeline.cpp:46-54        depth scale + mask (CPU branch)
//   /root/reference/src/pipeline.cpp:61-84        unprojection (CPU branch)
//   /root/reference/src/pipeline.cpp:211-241,251-257,275-282  demo scene / mask / model
//   /root/reference/src/pipeline.cpp:136-137      pose composition
// The reference TU itself cannot be compiled in this image (Eigen and OpenCV
// are absent), and the reference holds no result-pinning tests:
// PARITY UNPINNED by the reference.  Solver semantics follow Eigen 3.4.0 as
// restated in small_linalg.hpp.  std::mt19937, std::uniform_int_distribution
// <size_t> and std::unordered_map are used here exactly as the reference uses
// them (libstdc++ of g++ 11, the toolchain of this image).
//
// Layouts: points/normals/colours are AoS float[n*3] (bit-identical to
// std::vector<Eigen::Vector3f>); FPFH is float[n*33]; 4x4 transforms are
// COLUMN-MAJOR float[16] (Eigen::Matrix4f::data()).
//
// Build: g++ -O3 -DNDEBUG -ffp-contract=off -fPIC -shared (see oracle/Makefile).
#include "small_linalg.hpp"

#include <cstdint>
#include <cstring>
#include <cstdio>
#include <vector>
#include <array>
#include <random>
#include <unordered_map>
#include <algorithm>
#include <numeric>
#include <limits>
#include <cmath>
#include <fstream>
#include <string>

using namespace orc;

namespace {

struct VoxelKey {  // registration.cpp:15-18
    int x, y, z;
    bool operator==(const VoxelKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelKeyHash {  // registration.cpp:20-27
    size_t operator()(const VoxelKey& k) const {
        size_t h = std::hash<int>()(k.x);
        h ^= std::hash<int>()(k.y) + 0x9e3779b9 + (h << 6) + (h >> 2);
        h ^= std::hash<int>()(k.z) + 0x9e3779b9 + (h << 6) + (h >> 2);
        return h;
    }
};

inline float sqnorm_diff(const float* a, const float* b) {  // (a - b).squaredNorm()
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sum3(dx * dx, dy * dy, dz * dz);
}

// registration.cpp:63-81
void find_knn(const float* pts, int n, const float* query, int k, std::vector<size_t>& out) {
    std::vector<std::pair<float, size_t>> dists;
    dists.reserve(n);
    for (int i = 0; i < n; ++i) dists.emplace_back(sqnorm_diff(pts + 3 * i, query), (size_t)i);
    int kk = std::min(k, (int)dists.size());
    std::partial_sort(dists.begin(), dists.begin() + kk, dists.end());
    out.clear();
    for (int i = 0; i < kk; ++i) out.push_back(dists[i].second);
}

// registration.cpp:83-102
void find_radius_nn(const float* pts, int n, const float* query, float radius, int max_nn, std::vector<size_t>& out) {
    float r2 = radius * radius;
    std::vector<std::pair<float, size_t>> dists;
    for (int i = 0; i < n; ++i) {
        float d2 = sqnorm_diff(pts + 3 * i, query);
        if (d2 <= r2) dists.emplace_back(d2, (size_t)i);
    }
    std::sort(dists.begin(), dists.end());
    out.clear();
    for (int i = 0; i < std::min(max_nn, (int)dists.size()); ++i) out.push_back(dists[i].second);
}

inline void set_identity44(float* T) { for (int i = 0; i < 16; ++i) T[i] = 0.f; T[0] = T[5] = T[10] = T[15] = 1.f; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------- solvers (for unit tests)
void orc_jacobi_svd3(const float* A_colmajor, float* U, float* S, float* V) {
    M3 A; std::memcpy(A.m, A_colmajor, 36);
    SVD3 r = jacobi_svd3(A);
    std::memcpy(U, r.U.m, 36); std::memcpy(V, r.V.m, 36); std::memcpy(S, r.s, 12);
}
void orc_kabsch_rotation(const float* H_colmajor, float* R) {
    M3 H; std::memcpy(H.m, H_colmajor, 36);
    M3 r = kabsch_rotation(H);
    std::memcpy(R, r.m, 36);
}
int orc_self_adjoint_eig3(const float* A_colmajor, float* w, float* V) {
    M3 A; std::memcpy(A.m, A_colmajor, 36);
    Eig3 r = self_adjoint_eig3(A);
    std::memcpy(w, r.w, 12); std::memcpy(V, r.V.m, 36);
    return r.ok ? 0 : 1;
}
void orc_ldlt6_solve(const float* A_rowmajor, const float* b, float* x) { ldlt6_solve(A_rowmajor, b, x); }
void orc_euler_xyz_matrix(float a, float b, float g, float* R_colmajor) {
    M3 r = euler_xyz_matrix(a, b, g); std::memcpy(R_colmajor, r.m, 36);
}

// ---------------------------------------------------------------- depth + unprojection
// pipeline.cpp:46-54.  cv::Mat::convertTo(CV_32FC1, alpha) computes in float with
// alpha rounded to float (OpenCV cvtScale 16u->32f, work type float); alpha is the
// DOUBLE quotient 1.0 / scale.  Mask: threshold(mask,10,255,BINARY); setTo(0, ==0).
// mask may be NULL (apply_mask false / empty mask).
void orc_depth_preprocess(const uint16_t* raw, const uint8_t* mask, int width, int height, float scale, float* out) {
    const float a = (float)(1.0 / (double)scale);
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        float v = (float)raw[i] * a;
        if (mask && !(mask[i] > 10)) v = 0.f;
        out[i] = v;
    }
}
// pipeline.cpp:57 cv::countNonZero
// cv::resize(mask, resized_mask, depth.size(), 0, 0, cv::INTER_NEAREST), /root/reference/src/pipeline.cpp:38-41.  OpenCV is an
// un-vendored, un-pinned dependency (CMakeLists.txt:26); restated from its published resizeNN (modules/imgproc/src/resize.cpp,
// 4.x): resize() passes inv_scale_x = (double)dsize.width / ssize.width, resizeNN computes ifx = 1. / inv_scale_x and
// x_ofs[x] = min(cvFloor(x * ifx), ssize.width - 1); rows likewise with sy = min(cvFloor(y * ify), ssize.height - 1).
// (INTER_NEAREST, not INTER_NEAREST_EXACT: no half-pixel centre.)  Integer index arithmetic on a double product; pure-integer
// form floor(x * sw / dw) is the cross-check in tests/test_oracle_golden.py wherever the double product is exact.
void orc_mask_resize_nearest(const uint8_t* src, int sw, int sh, int dw, int dh, uint8_t* dst) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double ifx = 1. / inv_scale_x, ify = 1. / inv_scale_y;
    for (int y = 0; y < dh; ++y) {
        const int sy = std::min((int)std::floor(y * ify), sh - 1);
        for (int x = 0; x < dw; ++x) {
            const int sx = std::min((int)std::floor(x * ifx), sw - 1);
            dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
        }
    }
}

int orc_count_nonzero(const float* depth, int n) { int c = 0; for (int i = 0; i < n; ++i) c += depth[i] != 0.f; return c; }

// pipeline.cpp:68-83.  bgr may be NULL (rgb.empty()).  Returns the number of points.
int orc_unproject(const float* depth, const uint8_t* bgr, int width, int height,
                  float fx, float fy, float cx, float cy, float clipping_max,
                  float* xyz, float* rgb) {
    int n = 0;
    for (int v = 0; v < height; ++v) {
        for (int u = 0; u < width; ++u) {
            float z = depth[(size_t)v * width + u];
            if (z <= 0 || z > clipping_max) continue;
            float x = (u - cx) * z / fx;
            float y = (v - cy) * z / fy;
            xyz[3 * n + 0] = x; xyz[3 * n + 1] = y; xyz[3 * n + 2] = z;
            if (bgr && rgb) {
                const uint8_t* p = bgr + ((size_t)v * width + u) * 3;
                rgb[3 * n + 0] = p[2] / 255.0f; rgb[3 * n + 1] = p[1] / 255.0f; rgb[3 * n + 2] = p[0] / 255.0f;
            }
            ++n;
        }
    }
    return n;
}

// ---------------------------------------------------------------- demo inputs
// pipeline.cpp:211-241 (scene), :251-257 (mask), :275-282 (model).
void orc_demo_scene(int w, int h, float scale_to_meters, uint16_t* depth, uint8_t* bgr) {
    float cx = w / 2.0f, cy = h / 2.0f;
    float floor_z = 1.0f, box_z = 0.8f;
    for (int v = 0; v < h; ++v) {
        for (int u = 0; u < w; ++u) {
            uint8_t* px = bgr + ((size_t)v * w + u) * 3;
            px[0] = px[1] = px[2] = 50;
            float z = floor_z;
            if (std::abs(u - cx) < 100 && std::abs(v - cy) < 100) {
                z = box_z;
                px[0] = 0; px[1] = 0; px[2] = 255;
            } else if (((u / 50) + (v / 50)) % 2 == 0) {
                px[0] = px[1] = px[2] = 200;
            }
            depth[(size_t)v * w + u] = static_cast<unsigned short>(z * scale_to_meters);
        }
    }
}
void orc_demo_mask(int w, int h, uint8_t* mask) {  // cv::rectangle filled, both corners inclusive
    std::memset(mask, 0, (size_t)w * h);
    int cx = w / 2, cy = h / 2;
    for (int v = cy - 100; v <= cy + 100; ++v)
        for (int u = cx - 100; u <= cx + 100; ++u)
            if (u >= 0 && u < w && v >= 0 && v < h) mask[(size_t)v * w + u] = 255;
}
int orc_demo_model(float* xyz, float* normals, int capacity) {
    int n = 0;
    for (float x = -0.1f; x <= 0.1f; x += 0.005f)
        for (float y = -0.1f; y <= 0.1f; y += 0.005f) {
            if (n < capacity) {
                xyz[3 * n] = x; xyz[3 * n + 1] = y; xyz[3 * n + 2] = 0.0f;
                if (normals) { normals[3 * n] = 0; normals[3 * n + 1] = 0; normals[3 * n + 2] = 1; }
            }
            ++n;
        }
    return n;
}

// ---------------------------------------------------------------- voxelDownsample
// registration.cpp:29-60.  Output order = libstdc++ unordered_map iteration order.
// out_first_index (optional) receives, per output voxel, the smallest input index in it
// (lets a test match voxels between differently-ordered outputs).  Returns the voxel count.
int orc_voxel_downsample(const float* xyz, const float* rgb, int n, float voxel_size,
                         float* out_xyz, float* out_rgb, int* out_first_index, int capacity) {
    std::unordered_map<VoxelKey, std::vector<size_t>, VoxelKeyHash> grid;
    float inv = 1.0f / voxel_size;
    for (int i = 0; i < n; ++i) {
        VoxelKey key{static_cast<int>(std::floor(xyz[3 * i] * inv)),
                     static_cast<int>(std::floor(xyz[3 * i + 1] * inv)),
                     static_cast<int>(std::floor(xyz[3 * i + 2] * inv))};
        grid[key].push_back(i);
    }
    int m = 0;
    for (auto& kv : grid) {
        auto& indices = kv.second;
        float ap[3] = {0, 0, 0}, ac[3] = {0, 0, 0};
        for (size_t idx : indices) {
            for (int c = 0; c < 3; ++c) ap[c] += xyz[3 * idx + c];
            if (rgb) for (int c = 0; c < 3; ++c) ac[c] += rgb[3 * idx + c];
        }
        float cnt = static_cast<float>(indices.size());
        if (m < capacity) {
            for (int c = 0; c < 3; ++c) out_xyz[3 * m + c] = ap[c] / cnt;
            if (rgb && out_rgb) for (int c = 0; c < 3; ++c) out_rgb[3 * m + c] = ac[c] / cnt;
            if (out_first_index) out_first_index[m] = (int)indices[0];
        }
        ++m;
    }
    return m;
}

// ---------------------------------------------------------------- normals
// registration.cpp:105-130.  knn_out (optional) receives n*k neighbour indices
// (padded with -1 when n < k) in (d2, idx) order.
void orc_estimate_normals(const float* xyz, int n, int k, float* normals, int* knn_out) {
    std::vector<size_t> nb;
    for (int i = 0; i < n; ++i) {
        find_knn(xyz, n, xyz + 3 * i, k, nb);
        if (knn_out) for (int j = 0; j < k; ++j) knn_out[(size_t)i * k + j] = j < (int)nb.size() ? (int)nb[j] : -1;
        float c[3] = {0, 0, 0};
        for (size_t idx : nb) for (int a = 0; a < 3; ++a) c[a] += xyz[3 * idx + a];
        float cnt = static_cast<float>(nb.size());
        for (int a = 0; a < 3; ++a) c[a] /= cnt;
        M3 cov; for (int a = 0; a < 9; ++a) cov.m[a] = 0.f;
        for (size_t idx : nb) {
            float d[3] = {xyz[3 * idx] - c[0], xyz[3 * idx + 1] - c[1], xyz[3 * idx + 2] - c[2]};
            for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) cov(r, cc) += d[r] * d[cc];
        }
        for (int a = 0; a < 9; ++a) cov.m[a] /= cnt;
        Eig3 e = self_adjoint_eig3(cov);
        float nrm[3] = {e.V(0, 0), e.V(1, 0), e.V(2, 0)};
        float mp[3] = {-xyz[3 * i], -xyz[3 * i + 1], -xyz[3 * i + 2]};
        if (dot3(nrm, mp) < 0) for (int a = 0; a < 3; ++a) nrm[a] = -nrm[a];
        for (int a = 0; a < 3; ++a) normals[3 * i + a] = nrm[a];
    }
}

// ---------------------------------------------------------------- FPFH
// registration.cpp:133-201.  nbr_out/nbr_cnt (optional): n*100 radius-neighbour lists.
void orc_compute_fpfh(const float* xyz, const float* normals, int n, float radius, float* desc, int* nbr_out, int* nbr_cnt) {
    std::vector<std::array<float, 33>> spfh(n);
    std::vector<size_t> nb;
    std::vector<std::vector<size_t>> lists(n);
    for (int idx = 0; idx < n; ++idx) {
        std::array<float, 33> hist{};
        find_radius_nn(xyz, n, xyz + 3 * idx, radius, 100, nb);
        lists[idx] = nb;  // the reference repeats the identical search at :177
        if (nbr_out) {
            for (int j = 0; j < 100; ++j) nbr_out[(size_t)idx * 100 + j] = j < (int)nb.size() ? (int)nb[j] : -1;
            nbr_cnt[idx] = (int)nb.size();
        }
        for (size_t ni : nb) {
            if ((int)ni == idx) continue;
            float diff[3] = {xyz[3 * ni] - xyz[3 * idx], xyz[3 * ni + 1] - xyz[3 * idx + 1], xyz[3 * ni + 2] - xyz[3 * idx + 2]};
            float dist = std::sqrt(sum3(diff[0] * diff[0], diff[1] * diff[1], diff[2] * diff[2]));
            if (dist < 1e-8f) continue;
            const float* u = normals + 3 * idx;
            const float* nj = normals + 3 * ni;
            float dn[3] = {diff[0] / dist, diff[1] / dist, diff[2] / dist};
            float v[3] = {u[1] * dn[2] - u[2] * dn[1], u[2] * dn[0] - u[0] * dn[2], u[0] * dn[1] - u[1] * dn[0]};
            float w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
            float alpha = dot3(v, nj);
            float phi = dot3(u, dn);
            float theta = std::atan2(dot3(w, nj), dot3(u, nj));
            int bin_a = std::clamp(static_cast<int>((alpha + 1.0f) * 5.5f), 0, 10);
            int bin_p = std::clamp(static_cast<int>((phi + 1.0f) * 5.5f), 0, 10);
            int bin_t = std::clamp(static_cast<int>((theta / M_PI + 1.0f) * 5.5f), 0, 10);
            hist[bin_a] += 1.0f; hist[11 + bin_p] += 1.0f; hist[22 + bin_t] += 1.0f;
        }
        float sum = 0;
        for (float v : hist) sum += v;
        if (sum > 0) for (float& v : hist) v /= sum;
        spfh[idx] = hist;
    }
    for (int i = 0; i < n; ++i) {
        std::array<float, 33> f{};
        for (int d = 0; d < 33; ++d) f[d] = spfh[i][d];
        for (size_t ni : lists[i]) {
            if ((int)ni == i) continue;
            float diff[3] = {xyz[3 * ni] - xyz[3 * i], xyz[3 * ni + 1] - xyz[3 * i + 1], xyz[3 * ni + 2] - xyz[3 * i + 2]};
            float dist = std::sqrt(sum3(diff[0] * diff[0], diff[1] * diff[1], diff[2] * diff[2]));
            if (dist < 1e-8f) continue;
            float weight = 1.0f / dist;
            for (int d = 0; d < 33; ++d) f[d] += weight * spfh[ni][d];
        }
        float sum = 0;
        for (float v : f) sum += v;
        if (sum > 0) for (float& v : f) v /= sum;
        for (int d = 0; d < 33; ++d) desc[(size_t)i * 33 + d] = f[d];
    }
}

// ---------------------------------------------------------------- RANSAC
// registration.cpp:216-232
void orc_feature_match(const float* fs, int ns, const float* ft, int nt, int* corr) {
    for (int i = 0; i < ns; ++i) {
        float best = std::numeric_limits<float>::max();
        size_t bi = 0;
        for (int j = 0; j < nt; ++j) {
            float dist = 0;
            for (int d = 0; d < 33; ++d) {
                float diff = fs[(size_t)i * 33 + d] - ft[(size_t)j * 33 + d];
                dist += diff * diff;
            }
            if (dist < best) { best = dist; bi = j; }
        }
        corr[i] = (int)bi;
    }
}

// registration.cpp:235-239: the index stream.  out holds count*3 draws.
void orc_sample_triples(uint32_t seed, uint64_t n, int count, uint64_t* out) {
    std::mt19937 rng(seed);
    std::uniform_int_distribution<size_t> dist(0, n - 1);
    for (int i = 0; i < count; ++i) { out[3 * i] = dist(rng); out[3 * i + 1] = dist(rng); out[3 * i + 2] = dist(rng); }
}

// registration.cpp:242-268: hypothesis from 3 pairs.  T column-major.
static void hypothesis_from_pairs(const float* s0, const float* s1, const float* s2,
                                  const float* t0, const float* t1, const float* t2, M3& R, float* t) {
    const float* sp[3] = {s0, s1, s2};
    const float* tp[3] = {t0, t1, t2};
    float sc[3], tc[3];
    for (int r = 0; r < 3; ++r) {  // rowwise().mean() = (c0 + (c1 + c2)) / 3
        sc[r] = sum3(sp[0][r], sp[1][r], sp[2][r]) / 3.0f;
        tc[r] = sum3(tp[0][r], tp[1][r], tp[2][r]) / 3.0f;
    }
    M3 S, Tm;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { S(r, c) = sp[c][r] - sc[r]; Tm(r, c) = tp[c][r] - tc[r]; }
    M3 H = mul(S, transpose(Tm));
    R = kabsch_rotation(H);
    float Rs[3]; mulv(R, sc, Rs);
    for (int r = 0; r < 3; ++r) t[r] = tc[r] - Rs[r];
}
void orc_hypothesis_from_pairs(const float* s3x3, const float* t3x3, float* T) {
    M3 R; float t[3];
    hypothesis_from_pairs(s3x3, s3x3 + 3, s3x3 + 6, t3x3, t3x3 + 3, t3x3 + 6, R, t);
    set_identity44(T);
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) T[c * 4 + r] = R(r, c);
    for (int r = 0; r < 3; ++r) T[12 + r] = t[r];
}

// registration.cpp:204-295.  corr_in (optional): use these correspondences instead of
// running the feature match (lets tests isolate the hypothesis loop).  Optional traces:
// trace_inliers[max_iterations] (-1 for skipped iterations, untouched past an early exit),
// out_corr[ns], out_best_iter, out_iters_run.
void orc_ransac(const float* src, int ns, const float* tgt, int nt,
                const float* fs, const float* ft, const int* corr_in,
                float voxel_size, int max_iterations, float confidence,
                float* T_out, float* fitness_out, float* rmse_out,
                int* trace_inliers, int* out_corr, int* out_best_iter, int* out_iters_run) {
    float distance_threshold = voxel_size * 1.5f;
    std::vector<size_t> corr(ns);
    if (corr_in) for (int i = 0; i < ns; ++i) corr[i] = corr_in[i];
    else {
        std::vector<int> c(ns);
        orc_feature_match(fs, ns, ft, nt, c.data());
        for (int i = 0; i < ns; ++i) corr[i] = c[i];
    }
    if (out_corr) for (int i = 0; i < ns; ++i) out_corr[i] = (int)corr[i];

    float bestT[16]; set_identity44(bestT);
    float best_fitness = 0.f, best_rmse = 0.f;
    int best_iter = -1, iters_run = 0;
    std::mt19937 rng(42);
    std::uniform_int_distribution<size_t> dist(0, ns - 1);
    for (int iter = 0; iter < max_iterations; ++iter) {
        iters_run = iter + 1;
        size_t i0 = dist(rng), i1 = dist(rng), i2 = dist(rng);
        if (i0 == i1 || i1 == i2 || i0 == i2) { if (trace_inliers) trace_inliers[iter] = -1; continue; }
        M3 R; float t[3];
        hypothesis_from_pairs(src + 3 * i0, src + 3 * i1, src + 3 * i2,
                              tgt + 3 * corr[i0], tgt + 3 * corr[i1], tgt + 3 * corr[i2], R, t);
        int inliers = 0;
        float total_error = 0;
        for (int i = 0; i < ns; ++i) {
            float p[3]; mulv(R, src + 3 * i, p);
            for (int a = 0; a < 3; ++a) p[a] += t[a];
            float err = std::sqrt(sqnorm_diff(p, tgt + 3 * corr[i]));
            if (err < distance_threshold) { ++inliers; total_error += err * err; }
        }
        if (trace_inliers) trace_inliers[iter] = inliers;
        float fitness = static_cast<float>(inliers) / ns;
        float rmse = inliers > 0 ? std::sqrt(total_error / inliers) : 999.0f;
        if (fitness > best_fitness) {
            set_identity44(bestT);
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) bestT[c * 4 + r] = R(r, c);
            for (int r = 0; r < 3; ++r) bestT[12 + r] = t[r];
            best_fitness = fitness; best_rmse = rmse; best_iter = iter;
        }
        if (fitness > confidence) break;
    }
    std::memcpy(T_out, bestT, 64);
    *fitness_out = best_fitness; *rmse_out = best_rmse;
    if (out_best_iter) *out_best_iter = best_iter;
    if (out_iters_run) *out_iters_run = iters_run;
}

// ---------------------------------------------------------------- ICP
// One pass of registration.cpp:314-359 for a given T (column-major): nearest neighbours,
// acceptance, normal equations.  corr[i] = best index (always written), d2[i] = best_dist2,
// accepted[i] = 1 if d <= threshold.  ATA row-major 36, ATb 6 (zero in point-to-point mode).
void orc_icp_correspondences(const float* src, int ns, const float* tgt, const float* tgt_normals, int nt,
                             const float* T, float distance_threshold, int point_to_plane,
                             int* corr, float* d2_out, uint8_t* accepted,
                             int* n_corr_out, float* total_error_out, float* ATA, float* ATb) {
    M3 R; float t[3];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R(r, c) = T[c * 4 + r];
    for (int r = 0; r < 3; ++r) t[r] = T[12 + r];
    int n_corr = 0; float total_error = 0;
    for (int i = 0; i < 36; ++i) ATA[i] = 0; for (int i = 0; i < 6; ++i) ATb[i] = 0;
    bool p2pl = point_to_plane && tgt_normals;
    for (int i = 0; i < ns; ++i) {
        float p[3]; mulv(R, src + 3 * i, p);
        for (int a = 0; a < 3; ++a) p[a] += t[a];
        float best = std::numeric_limits<float>::max(); size_t bi = 0;
        for (int j = 0; j < nt; ++j) {
            float d2 = sqnorm_diff(p, tgt + 3 * j);
            if (d2 < best) { best = d2; bi = j; }
        }
        if (corr) corr[i] = (int)bi;
        if (d2_out) d2_out[i] = best;
        float d = std::sqrt(best);
        bool acc = !(d > distance_threshold);
        if (accepted) accepted[i] = acc;
        if (!acc) continue;
        ++n_corr; total_error += best;
        if (p2pl) {
            const float* q = tgt + 3 * bi; const float* n = tgt_normals + 3 * bi;
            float J[6] = {p[1] * n[2] - p[2] * n[1], p[2] * n[0] - p[0] * n[2], p[0] * n[1] - p[1] * n[0], n[0], n[1], n[2]};
            float pq[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
            float residual = dot3(pq, n);
            for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) ATA[r * 6 + c] += J[r] * J[c]; ATb[r] += J[r] * residual; }
        }
    }
    *n_corr_out = n_corr; *total_error_out = total_error;
}

// registration.cpp:297-414.  tgt_normals may be NULL (hasNormals() false).
// trace (optional): per iteration 20 floats = T after the update (16, column-major), rmse,
// fitness, n_corr, 0.  Returns the number of iterations whose update was applied.
int orc_icp(const float* src, int ns, const float* tgt, const float* tgt_normals, int nt,
            const float* T0, float distance_threshold, int max_iterations, int point_to_plane,
            float* T_out, float* fitness_out, float* rmse_out, float* trace) {
    float T[16]; std::memcpy(T, T0, 64);
    float res_T[16]; std::memcpy(res_T, T, 64);
    float res_fitness = 0.f, res_rmse = 0.f;
    int applied = 0;
    bool p2pl = point_to_plane && tgt_normals;
    std::vector<float> sc, tc;
    for (int iter = 0; iter < max_iterations; ++iter) {
        M3 R; float t[3];
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R(r, c) = T[c * 4 + r];
        for (int r = 0; r < 3; ++r) t[r] = T[12 + r];
        int n_corr = 0; float total_error = 0;
        float ATA[36] = {0}, ATb[6] = {0};
        sc.clear(); tc.clear();
        for (int i = 0; i < ns; ++i) {
            float p[3]; mulv(R, src + 3 * i, p);
            for (int a = 0; a < 3; ++a) p[a] += t[a];
            float best = std::numeric_limits<float>::max(); size_t bi = 0;
            for (int j = 0; j < nt; ++j) {
                float d2 = sqnorm_diff(p, tgt + 3 * j);
                if (d2 < best) { best = d2; bi = j; }
            }
            float d = std::sqrt(best);
            if (d > distance_threshold) continue;
            ++n_corr; total_error += best;
            if (p2pl) {
                const float* q = tgt + 3 * bi; const float* n = tgt_normals + 3 * bi;
                float J[6] = {p[1] * n[2] - p[2] * n[1], p[2] * n[0] - p[0] * n[2], p[0] * n[1] - p[1] * n[0], n[0], n[1], n[2]};
                float pq[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
                float residual = dot3(pq, n);
                for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) ATA[r * 6 + c] += J[r] * J[c]; ATb[r] += J[r] * residual; }
            } else {
                for (int a = 0; a < 3; ++a) { sc.push_back(p[a]); tc.push_back(tgt[3 * bi + a]); }
            }
        }
        if (n_corr < 3) break;
        float delta[16]; set_identity44(delta);
        if (p2pl) {
            float nb[6], x[6];
            for (int i = 0; i < 6; ++i) nb[i] = -ATb[i];
            ldlt6_solve(ATA, nb, x);
            M3 dR = euler_xyz_matrix(x[0], x[1], x[2]);
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) delta[c * 4 + r] = dR(r, c);
            for (int r = 0; r < 3; ++r) delta[12 + r] = x[3 + r];
        } else {
            size_t m = sc.size() / 3;
            float sm[3] = {0, 0, 0}, tm[3] = {0, 0, 0};
            for (size_t i = 0; i < m; ++i) for (int a = 0; a < 3; ++a) { sm[a] += sc[3 * i + a]; tm[a] += tc[3 * i + a]; }
            for (int a = 0; a < 3; ++a) { sm[a] /= static_cast<float>(m); tm[a] /= static_cast<float>(m); }
            M3 H; for (int a = 0; a < 9; ++a) H.m[a] = 0.f;
            for (size_t i = 0; i < m; ++i) {
                float a3[3] = {sc[3 * i] - sm[0], sc[3 * i + 1] - sm[1], sc[3 * i + 2] - sm[2]};
                float b3[3] = {tc[3 * i] - tm[0], tc[3 * i + 1] - tm[1], tc[3 * i + 2] - tm[2]};
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) H(r, c) += a3[r] * b3[c];
            }
            M3 dR = kabsch_rotation(H);
            float Rs[3]; mulv(dR, sm, Rs);
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) delta[c * 4 + r] = dR(r, c);
            for (int r = 0; r < 3; ++r) delta[12 + r] = tm[r] - Rs[r];
        }
        mul44(delta, T, T);
        float prev_rmse = res_rmse;
        res_rmse = std::sqrt(total_error / n_corr);
        res_fitness = static_cast<float>(n_corr) / ns;
        std::memcpy(res_T, T, 64);
        applied = iter + 1;
        if (trace) {
            float* tr = trace + (size_t)iter * 20;
            std::memcpy(tr, T, 64); tr[16] = res_rmse; tr[17] = res_fitness; tr[18] = (float)n_corr; tr[19] = 0.f;
        }
        if (iter > 0 && std::abs(prev_rmse - res_rmse) < 1e-6f) break;
    }
    std::memcpy(T_out, res_T, 64);
    *fitness_out = res_fitness; *rmse_out = res_rmse;
    return applied;
}

// ---------------------------------------------------------------- neighbours of the path
// cuda/depth_processing.cu:62-155 (bilateral filter; the reference has no CPU version and never calls it).
void orc_bilateral_filter(const float* in, float* out, int width, int height, float sigma_spatial, float sigma_range) {
    int radius = static_cast<int>(2.0f * sigma_spatial + 0.5f);
    if (radius > 5) radius = 5;
    const float inv_spatial2 = -0.5f / (sigma_spatial * sigma_spatial);
    const float inv_range2 = -0.5f / (sigma_range * sigma_range);
    auto at = [&](int x, int y) -> float { return (x >= 0 && x < width && y >= 0 && y < height) ? in[(size_t)y * width + x] : 0.0f; };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            float center = at(x, y);
            if (center <= 0.0f) { out[(size_t)y * width + x] = 0.0f; continue; }
            float sum_w = 0.0f, sum_v = 0.0f;
            for (int dy = -radius; dy <= radius; ++dy)
                for (int dx = -radius; dx <= radius; ++dx) {
                    float nb = at(x + dx, y + dy);
                    if (nb <= 0.0f) continue;
                    float rd = nb - center;
                    float w = expf(static_cast<float>(dx * dx + dy * dy) * inv_spatial2 + rd * rd * inv_range2);
                    sum_w += w; sum_v += w * nb;
                }
            out[(size_t)y * width + x] = (sum_w > 0.0f) ? (sum_v / sum_w) : center;
        }
}

// pipeline.cpp:153-180.  poses: n column-major 4x4.  Returns the kept count.
int orc_filter_duplicates(const float* poses, int n, float min_distance, float* out) {
    std::vector<std::array<float, 16>> filtered;
    auto norm3 = [](float x, float y, float z) { return std::sqrt(sum3(x * x, y * y, z * z)); };
    for (int w = 0; w < n; ++w) {
        const float* wp = poses + (size_t)w * 16;
        bool is_dup = false;
        for (size_t i = 0; i < filtered.size(); ++i) {
            float dist = norm3(wp[12] - filtered[i][12], wp[13] - filtered[i][13], wp[14] - filtered[i][14]);
            if (dist < min_distance) {
                is_dup = true;
                float existing_dist = norm3(filtered[i][12], filtered[i][13], filtered[i][14]);
                float current_dist = norm3(wp[12], wp[13], wp[14]);
                if (current_dist < existing_dist) std::memcpy(filtered[i].data(), wp, 64);
                break;
            }
        }
        if (!is_dup) { std::array<float, 16> a; std::memcpy(a.data(), wp, 64); filtered.push_back(a); }
    }
    for (size_t i = 0; i < filtered.size(); ++i) std::memcpy(out + i * 16, filtered[i].data(), 64);
    return (int)filtered.size();
}

// registration.cpp:416-461.  The reference's float x, y, z are uninitialised when the last read fails
// (C++11 stores 0 in x; y, z keep indeterminate values): this restatement zero-initialises them.
// Returns the number of points (-1 if the file cannot be opened); *has_color_out as detected.
int orc_load_ply(const char* path, float* xyz, float* rgb, int capacity, int* has_color_out) {
    std::ifstream file(path);
    if (!file.is_open()) return -1;
    std::string line;
    int vertex_count = 0;
    bool has_color = false, in_header = true;
    while (std::getline(file, line) && in_header) {
        if (line.find("element vertex") != std::string::npos) sscanf(line.c_str(), "element vertex %d", &vertex_count);
        if (line.find("red") != std::string::npos || line.find("diffuse_red") != std::string::npos) has_color = true;
        if (line == "end_header") in_header = false;
    }
    *has_color_out = has_color;
    int n = 0;
    for (int i = 0; i < vertex_count; ++i) {
        float x = 0, y = 0, z = 0;
        file >> x >> y >> z;
        if (n < capacity) { xyz[3 * n] = x; xyz[3 * n + 1] = y; xyz[3 * n + 2] = z; }
        if (has_color) {
            float r = 0, g = 0, b = 0;
            file >> r >> g >> b;
            if (r > 1.0f) { r /= 255.0f; g /= 255.0f; b /= 255.0f; }
            if (n < capacity && rgb) { rgb[3 * n] = r; rgb[3 * n + 1] = g; rgb[3 * n + 2] = b; }
        }
        ++n;
        std::getline(file, line);
    }
    return n;
}

// ---------------------------------------------------------------- pose composition
// pipeline.cpp:136-137: T_world_object = extrinsics * refined.inverse().  The reference uses
// Eigen's general 4x4 inverse; this restates it as a cofactor inverse in float (tolerance use only).
void orc_pose_compose(const float* extrinsics, const float* T, float* out) {
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = T[i];
    // column-major general inverse via adjugate (double intermediates, rounded once)
    auto A = [&](int r, int c) { return a[c * 4 + r]; };
    double cof[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        int rr[3], cc[3], k = 0;
        for (int i = 0; i < 4; ++i) if (i != r) rr[k++] = i;
        k = 0; for (int i = 0; i < 4; ++i) if (i != c) cc[k++] = i;
        double m = A(rr[0], cc[0]) * (A(rr[1], cc[1]) * A(rr[2], cc[2]) - A(rr[1], cc[2]) * A(rr[2], cc[1]))
                 - A(rr[0], cc[1]) * (A(rr[1], cc[0]) * A(rr[2], cc[2]) - A(rr[1], cc[2]) * A(rr[2], cc[0]))
                 + A(rr[0], cc[2]) * (A(rr[1], cc[0]) * A(rr[2], cc[1]) - A(rr[1], cc[1]) * A(rr[2], cc[0]));
        cof[c * 4 + r] = ((r + c) & 1) ? -m : m;
    }
    double det = 0; for (int c = 0; c < 4; ++c) det += A(0, c) * cof[c * 4 + 0];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv[c * 4 + r] = cof[r * 4 + c] / det;
    float invf[16]; for (int i = 0; i < 16; ++i) invf[i] = (float)inv[i];
    mul44(extrinsics, invf, out);
}

}  // extern "C"
