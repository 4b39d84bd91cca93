// extern "C" entry points of include/tdv_hip.h: argument checks, host<->device staging through the
// ctx workspace, dispatch to the device pipelines.  This TU replaces the reference's CUDA
// dispatch layer /root/reference/src/gpu_impl.cpp (per-call cudaMalloc/cudaMemcpy/launch/cudaFree)
// with arena-backed staging on one stream per ctx.
#include "tdv_internal.hpp"
#include <cstring>
#include <algorithm>

using namespace tdv;

namespace {

template <class T>
int upload(tdv_ctx* ctx, const T* host, size_t count, T** dev) {
    *dev = nullptr;
    if (!host || count == 0) return TDV_OK;
    TDV_TRY(ws_alloc(ctx, count, dev));
    TDV_HIP(ctx, hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return TDV_OK;
}
template <class T>
int download(tdv_ctx* ctx, T* host, const T* dev, size_t count) {
    if (!host || !dev || count == 0) return TDV_OK;
    TDV_HIP(ctx, hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return TDV_OK;
}
int begin(tdv_ctx* ctx) {
    if (!ctx) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    return ws_reset(ctx);
}
int finish(tdv_ctx* ctx) {
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TDV_OK;
}

}  // namespace

extern "C" {

int tdv_depth_preprocess(tdv_ctx* ctx, const uint16_t* raw, const uint8_t* mask, int width, int height,
                         float scale, int mask_mode, float* out_depth) {
    if (!raw || !out_depth || width < 0 || height < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    const size_t n = (size_t)width * height;
    if (n == 0) return TDV_OK;
    uint16_t* d_raw; uint8_t* d_mask; float* d_out;
    TDV_TRY(upload(ctx, raw, n, &d_raw));
    TDV_TRY(upload(ctx, mask, n, &d_mask));
    TDV_TRY(ws_alloc(ctx, n, &d_out));
    TDV_TRY(depth_preprocess_dev(ctx, d_raw, d_mask, width, height, scale, mask_mode, d_out));
    TDV_TRY(download(ctx, out_depth, d_out, n));
    return finish(ctx);
}

int tdv_bilateral_filter(tdv_ctx* ctx, const float* depth, int width, int height, float sigma_spatial, float sigma_range,
                         float* out_depth) {
    if (!depth || !out_depth || width < 0 || height < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    const size_t n = (size_t)width * height;
    if (n == 0) return TDV_OK;
    float *d_in, *d_out;
    TDV_TRY(upload(ctx, depth, n, &d_in));
    TDV_TRY(ws_alloc(ctx, n, &d_out));
    TDV_TRY(bilateral_filter_dev(ctx, d_in, d_out, width, height, sigma_spatial, sigma_range));
    TDV_TRY(download(ctx, out_depth, d_out, n));
    return finish(ctx);
}

static int voxel_batch_impl(tdv_ctx* ctx, const float* d_xyz, const int* h_cloud_offsets, int n_clouds, float voxel_size, const float* pinhole4,
                            float* d_out_xyz, int* h_voxel_offsets);
int tdv_voxel_downsample_batch_dev(tdv_ctx* ctx, const float* d_xyz, const int* h_cloud_offsets, int n_clouds, float voxel_size,
                                   float* d_out_xyz, int* h_voxel_offsets) {
    return voxel_batch_impl(ctx, d_xyz, h_cloud_offsets, n_clouds, voxel_size, nullptr, d_out_xyz, h_voxel_offsets);
}
int tdv_voxel_downsample_batch_pinhole_dev(tdv_ctx* ctx, const float* d_xyz, const int* h_cloud_offsets, int n_clouds, float voxel_size,
                                           float fx, float fy, float cx, float cy, float* d_out_xyz, int* h_voxel_offsets) {
    if (!(fx > 0.f) || !(fy > 0.f)) return TDV_ERR_BAD_ARG;
    const float cam[4] = {fx, fy, cx, cy};
    return voxel_batch_impl(ctx, d_xyz, h_cloud_offsets, n_clouds, voxel_size, cam, d_out_xyz, h_voxel_offsets);
}
static int voxel_batch_impl(tdv_ctx* ctx, const float* d_xyz, const int* h_cloud_offsets, int n_clouds, float voxel_size, const float* pinhole4,
                            float* d_out_xyz, int* h_voxel_offsets) {
    if (!h_cloud_offsets || !h_voxel_offsets || n_clouds < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    h_voxel_offsets[0] = 0;
    if (n_clouds == 0) return TDV_OK;
    for (int b = 0; b < n_clouds; ++b) if (h_cloud_offsets[b] > h_cloud_offsets[b + 1] || h_cloud_offsets[b] < 0) return TDV_ERR_BAD_ARG;
    const int total = h_cloud_offsets[n_clouds];
    if (h_cloud_offsets[0] != 0 || (total > 0 && (!d_xyz || !d_out_xyz))) return TDV_ERR_BAD_ARG;
    int* d_off;
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds + 1, &d_off));
    TDV_HIP(ctx, hipMemcpyAsync(d_off, h_cloud_offsets, ((size_t)n_clouds + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    int overflowed = 0;
    TDV_TRY(voxel_downsample_batch_dev(ctx, d_xyz, total, d_off, n_clouds, voxel_size, d_out_xyz, nullptr, nullptr, h_voxel_offsets, &overflowed, nullptr,
                                       pinhole4, pinhole4 ? h_cloud_offsets : nullptr));
    if (!overflowed) return TDV_OK;
    // a voxel with more points than the table's member rows hold (a coarse grid): cloud by cloud on the path without that limit
    int at = 0;
    for (int b = 0; b < n_clouds; ++b) {
        const int n = h_cloud_offsets[b + 1] - h_cloud_offsets[b];
        int v = 0;
        h_voxel_offsets[b] = at;
        if (n > 0) TDV_TRY(voxel_downsample_dev(ctx, d_xyz + (size_t)h_cloud_offsets[b] * 3, nullptr, n, voxel_size, TDV_VOXEL_ORDER_FIRST, d_out_xyz + (size_t)at * 3, nullptr, n, &v));
        at += v;
    }
    h_voxel_offsets[n_clouds] = at;
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TDV_OK;
}

int tdv_mask_resize_nearest(tdv_ctx* ctx, const uint8_t* masks, int n_masks, int src_width, int src_height, int dst_width, int dst_height, uint8_t* out) {
    if (n_masks < 0 || src_width <= 0 || src_height <= 0 || dst_width < 0 || dst_height < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    const size_t ns = (size_t)src_width * src_height * n_masks, nd = (size_t)dst_width * dst_height * n_masks;
    if (nd == 0) return TDV_OK;
    if (!masks || !out) return TDV_ERR_BAD_ARG;
    uint8_t *d_in, *d_out;
    TDV_TRY(upload(ctx, masks, ns, &d_in));
    TDV_TRY(ws_alloc(ctx, nd, &d_out));
    TDV_TRY(mask_resize_nearest_dev(ctx, d_in, n_masks, src_width, src_height, dst_width, dst_height, d_out));
    TDV_TRY(download(ctx, out, d_out, nd));
    return finish(ctx);
}

int tdv_mask_resize_nearest_dev(tdv_ctx* ctx, const uint8_t* d_masks, int n_masks, int src_width, int src_height, int dst_width, int dst_height, uint8_t* d_out) {
    TDV_TRY(begin(ctx));
    return mask_resize_nearest_dev(ctx, d_masks, n_masks, src_width, src_height, dst_width, dst_height, d_out);
}

int tdv_deproject(tdv_ctx* ctx, const float* depth, const uint8_t* bgr, int width, int height,
                  float fx, float fy, float cx, float cy, float zmax,
                  float* out_xyz, float* out_rgb, int capacity, int* n_out) {
    if (!depth || !n_out || width < 0 || height < 0 || capacity < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    *n_out = 0;
    const size_t n = (size_t)width * height;
    if (n == 0) return TDV_OK;
    float* d_depth; uint8_t* d_bgr; float *d_xyz = nullptr, *d_rgb = nullptr;
    TDV_TRY(upload(ctx, depth, n, &d_depth));
    TDV_TRY(upload(ctx, bgr, n * 3, &d_bgr));
    const size_t cap = (size_t)std::min<size_t>((size_t)capacity, n);
    if (out_xyz && cap) TDV_TRY(ws_alloc(ctx, cap * 3, &d_xyz));
    if (out_rgb && bgr && cap) TDV_TRY(ws_alloc(ctx, cap * 3, &d_rgb));
    int st = depth_to_cloud_dev(ctx, nullptr, d_depth, nullptr, d_bgr, width, height, 1.f, 0, fx, fy, cx, cy, zmax,
                                d_xyz, d_rgb, (int)cap, n_out);
    if (st != TDV_OK) return st;
    TDV_TRY(download(ctx, out_xyz, d_xyz, (size_t)*n_out * 3));
    TDV_TRY(download(ctx, out_rgb, d_rgb, (size_t)*n_out * 3));
    return finish(ctx);
}

int tdv_depth_to_cloud(tdv_ctx* ctx, const uint16_t* raw, const uint8_t* mask, const uint8_t* bgr,
                       int width, int height, float scale, int mask_mode,
                       float fx, float fy, float cx, float cy, float zmax,
                       float* out_xyz, float* out_rgb, int capacity, int* n_out) {
    if (!raw || !n_out || width < 0 || height < 0 || capacity < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    *n_out = 0;
    const size_t n = (size_t)width * height;
    if (n == 0) return TDV_OK;
    uint16_t* d_raw; uint8_t *d_mask, *d_bgr; float *d_xyz = nullptr, *d_rgb = nullptr;
    TDV_TRY(upload(ctx, raw, n, &d_raw));
    TDV_TRY(upload(ctx, mask, n, &d_mask));
    TDV_TRY(upload(ctx, bgr, n * 3, &d_bgr));
    const size_t cap = (size_t)std::min<size_t>((size_t)capacity, n);
    if (out_xyz && cap) TDV_TRY(ws_alloc(ctx, cap * 3, &d_xyz));
    if (out_rgb && bgr && cap) TDV_TRY(ws_alloc(ctx, cap * 3, &d_rgb));
    int st = depth_to_cloud_dev(ctx, d_raw, nullptr, d_mask, d_bgr, width, height, scale, mask_mode, fx, fy, cx, cy, zmax,
                                d_xyz, d_rgb, (int)cap, n_out);
    if (st != TDV_OK) return st;
    TDV_TRY(download(ctx, out_xyz, d_xyz, (size_t)*n_out * 3));
    TDV_TRY(download(ctx, out_rgb, d_rgb, (size_t)*n_out * 3));
    return finish(ctx);
}

int tdv_voxel_downsample(tdv_ctx* ctx, const float* xyz, const float* rgb, int n, float voxel_size, int order,
                         float* out_xyz, float* out_rgb, int capacity, int* n_out) {
    if (!n_out || n < 0 || capacity < 0 || (n > 0 && (!xyz || !out_xyz)) || !(voxel_size > 0.f)) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    *n_out = 0;
    if (n == 0) return TDV_OK;
    float *d_xyz, *d_rgb, *d_oxyz = nullptr, *d_orgb = nullptr;
    TDV_TRY(upload(ctx, xyz, (size_t)n * 3, &d_xyz));
    TDV_TRY(upload(ctx, rgb, (size_t)n * 3, &d_rgb));
    const int cap = std::min(capacity, n);
    if (cap) TDV_TRY(ws_alloc(ctx, (size_t)cap * 3, &d_oxyz));
    if (rgb && out_rgb && cap) TDV_TRY(ws_alloc(ctx, (size_t)cap * 3, &d_orgb));
    int st = voxel_downsample_dev(ctx, d_xyz, d_rgb, n, voxel_size, order, d_oxyz, d_orgb, cap, n_out);
    if (st != TDV_OK) return st;
    TDV_TRY(download(ctx, out_xyz, d_oxyz, (size_t)*n_out * 3));
    TDV_TRY(download(ctx, out_rgb, d_orgb, (size_t)*n_out * 3));
    return finish(ctx);
}

int tdv_estimate_normals(tdv_ctx* ctx, const float* xyz, int n, int k, float* out_normals, int* out_knn) {
    if (n < 0 || k <= 0 || (n > 0 && (!xyz || !out_normals))) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    if (n == 0) return TDV_OK;
    float *d_xyz, *d_nrm; int* d_knn = nullptr;
    TDV_TRY(upload(ctx, xyz, (size_t)n * 3, &d_xyz));
    TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &d_nrm));
    if (out_knn) TDV_TRY(ws_alloc(ctx, (size_t)n * k, &d_knn));
    TDV_TRY(estimate_normals_dev(ctx, d_xyz, n, k, d_nrm, d_knn));
    TDV_TRY(download(ctx, out_normals, d_nrm, (size_t)n * 3));
    TDV_TRY(download(ctx, out_knn, d_knn, (size_t)n * k));
    return finish(ctx);
}

int tdv_compute_fpfh(tdv_ctx* ctx, const float* xyz, const float* normals, int n, float radius,
                     float* out_desc33, int* out_nbr, int* out_nbr_cnt) {
    if (n < 0 || (n > 0 && (!xyz || !normals || !out_desc33))) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    if (n == 0) return TDV_OK;
    float *d_xyz, *d_nrm, *d_desc; int *d_nbr = nullptr, *d_cnt = nullptr;
    TDV_TRY(upload(ctx, xyz, (size_t)n * 3, &d_xyz));
    TDV_TRY(upload(ctx, normals, (size_t)n * 3, &d_nrm));
    TDV_TRY(ws_alloc(ctx, (size_t)n * 33, &d_desc));
    if (out_nbr) TDV_TRY(ws_alloc(ctx, (size_t)n * 100, &d_nbr));
    if (out_nbr_cnt) TDV_TRY(ws_alloc(ctx, (size_t)n, &d_cnt));
    TDV_TRY(compute_fpfh_dev(ctx, d_xyz, d_nrm, n, radius, d_desc, d_nbr, d_cnt));
    TDV_TRY(download(ctx, out_desc33, d_desc, (size_t)n * 33));
    TDV_TRY(download(ctx, out_nbr, d_nbr, (size_t)n * 100));
    TDV_TRY(download(ctx, out_nbr_cnt, d_cnt, (size_t)n));
    return finish(ctx);
}

int tdv_feature_match(tdv_ctx* ctx, const float* fs, int ns, const float* ft, int nt, int* out_corr) {
    if (ns < 0 || nt < 0 || (ns > 0 && (!fs || !out_corr)) || (nt > 0 && !ft)) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    if (ns == 0) return TDV_OK;
    if (nt == 0) { std::memset(out_corr, 0, (size_t)ns * 4); return TDV_OK; }
    float *d_fs, *d_ft; int* d_corr;
    TDV_TRY(upload(ctx, fs, (size_t)ns * 33, &d_fs));
    TDV_TRY(upload(ctx, ft, (size_t)nt * 33, &d_ft));
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &d_corr));
    TDV_TRY(feature_match_dev(ctx, d_fs, ns, d_ft, nt, d_corr));
    TDV_TRY(download(ctx, out_corr, d_corr, (size_t)ns));
    return finish(ctx);
}

int tdv_ransac(tdv_ctx* ctx, const float* src, int ns, const float* tgt, int nt,
               const float* fs, const float* ft, const int* corr,
               float voxel_size, int max_iterations, float confidence, uint32_t seed,
               tdv_ransac_result* out, int* trace_inliers) {
    if (!out || ns < 0 || nt < 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    float *d_src, *d_tgt, *d_fs = nullptr, *d_ft = nullptr; int* d_corr = nullptr;
    TDV_TRY(upload(ctx, src, (size_t)ns * 3, &d_src));
    TDV_TRY(upload(ctx, tgt, (size_t)nt * 3, &d_tgt));
    if (corr) TDV_TRY(upload(ctx, corr, (size_t)ns, &d_corr));
    else { TDV_TRY(upload(ctx, fs, (size_t)ns * 33, &d_fs)); TDV_TRY(upload(ctx, ft, (size_t)nt * 33, &d_ft)); }
    return ransac_run_dev(ctx, d_src, ns, d_tgt, nt, d_fs, d_ft, d_corr, voxel_size, max_iterations, confidence, seed, out, trace_inliers);
}

int tdv_icp(tdv_ctx* ctx, const float* src, int ns, const float* tgt, const float* tgt_normals, int nt,
            const float* T0, float distance_threshold, int max_iterations, int point_to_plane,
            tdv_icp_result* out) {
    if (!out || !T0 || ns < 0 || nt < 0 || (ns > 0 && !src) || (nt > 0 && !tgt)) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    float *d_src, *d_tgt, *d_nrm;
    TDV_TRY(upload(ctx, src, (size_t)ns * 3, &d_src));
    TDV_TRY(upload(ctx, tgt, (size_t)nt * 3, &d_tgt));
    TDV_TRY(upload(ctx, tgt_normals, (size_t)nt * 3, &d_nrm));
    if (ns == 0 || nt == 0) {
        std::memcpy(out->T, T0, 64); out->fitness = 0.f; out->rmse = 0.f; out->iterations = 0; out->n_corr = 0;
        return TDV_OK;
    }
    return icp_run_dev(ctx, d_src, ns, d_tgt, d_nrm, nt, T0, distance_threshold, max_iterations, point_to_plane, 0, out);
}

int tdv_icp_correspondences(tdv_ctx* ctx, const float* src, int ns, const float* tgt, int nt,
                            const float* T, float distance_threshold,
                            int* out_corr, float* out_d2, uint8_t* out_accepted, int* out_n_corr) {
    if (!src || !tgt || !T || ns <= 0 || nt <= 0) return TDV_ERR_BAD_ARG;
    TDV_TRY(begin(ctx));
    float *d_src, *d_tgt;
    TDV_TRY(upload(ctx, src, (size_t)ns * 3, &d_src));
    TDV_TRY(upload(ctx, tgt, (size_t)nt * 3, &d_tgt));
    IcpOutputs o;
    if (out_corr) TDV_TRY(ws_alloc(ctx, (size_t)ns, &o.corr));
    if (out_d2) TDV_TRY(ws_alloc(ctx, (size_t)ns, &o.d2));
    if (out_accepted) TDV_TRY(ws_alloc(ctx, (size_t)ns, &o.accepted));
    TDV_TRY(icp_correspondences_dev(ctx, d_src, ns, d_tgt, nt, T, distance_threshold, o, out_n_corr));
    TDV_TRY(download(ctx, out_corr, o.corr, (size_t)ns));
    TDV_TRY(download(ctx, out_d2, o.d2, (size_t)ns));
    TDV_TRY(download(ctx, out_accepted, o.accepted, (size_t)ns));
    return finish(ctx);
}

// ---- device-resident entry points
int tdv_icp_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, const float* d_tgt_normals, int nt,
                const float* T0, float distance_threshold, int max_iterations, int point_to_plane,
                int fixed_iterations, tdv_icp_result* out) {
    TDV_TRY(begin(ctx));
    return icp_run_dev(ctx, d_src, ns, d_tgt, d_tgt_normals, nt, T0, distance_threshold, max_iterations, point_to_plane, fixed_iterations, out);
}
int tdv_ransac_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt,
                   const float* d_fs, const float* d_ft, const int* d_corr,
                   float voxel_size, int max_iterations, float confidence, uint32_t seed,
                   tdv_ransac_result* out, int* trace_inliers) {
    TDV_TRY(begin(ctx));
    return ransac_run_dev(ctx, d_src, ns, d_tgt, nt, d_fs, d_ft, d_corr, voxel_size, max_iterations, confidence, seed, out, trace_inliers);
}
int tdv_feature_match_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr) {
    TDV_TRY(begin(ctx));
    TDV_TRY(feature_match_dev(ctx, d_fs, ns, d_ft, nt, d_corr));
    return finish(ctx);
}
int tdv_estimate_normals_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float* d_normals, int* d_knn) {
    TDV_TRY(begin(ctx));
    TDV_TRY(estimate_normals_dev(ctx, d_xyz, n, k, d_normals, d_knn));
    return finish(ctx);
}
int tdv_compute_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, float radius,
                         float* d_desc33, int* d_nbr, int* d_nbr_cnt) {
    TDV_TRY(begin(ctx));
    TDV_TRY(compute_fpfh_dev(ctx, d_xyz, d_normals, n, radius, d_desc33, d_nbr, d_nbr_cnt));
    return finish(ctx);
}
int tdv_normals_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float radius, float* d_normals, float* d_desc33) {
    TDV_TRY(begin(ctx));
    TDV_TRY(normals_fpfh_dev(ctx, d_xyz, n, k, radius, d_normals, d_desc33));
    return finish(ctx);
}
int tdv_radix_sort_pairs_dev(tdv_ctx* ctx, const unsigned long long* d_keys_in, unsigned long long* d_keys_out, const unsigned* d_vals_in,
                             unsigned* d_vals_out, size_t n, int end_bit) {
    TDV_TRY(begin(ctx));
    TDV_TRY(radix_sort_pairs_dev(ctx, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, end_bit));
    return finish(ctx);
}
int tdv_depth_to_cloud_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_mask, const uint8_t* d_bgr,
                           int width, int height, float scale, int mask_mode,
                           float fx, float fy, float cx, float cy, float zmax,
                           float* d_xyz, float* d_rgb, int capacity, int* n_out) {
    TDV_TRY(begin(ctx));
    return depth_to_cloud_dev(ctx, d_raw, nullptr, d_mask, d_bgr, width, height, scale, mask_mode, fx, fy, cx, cy, zmax,
                              d_xyz, d_rgb, capacity, n_out);
}
int tdv_voxel_downsample_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int n, float voxel_size, int order,
                             float* d_out_xyz, float* d_out_rgb, int capacity, int* n_out) {
    TDV_TRY(begin(ctx));
    return voxel_downsample_dev(ctx, d_xyz, d_rgb, n, voxel_size, order, d_out_xyz, d_out_rgb, capacity, n_out);
}

}  // extern "C"
