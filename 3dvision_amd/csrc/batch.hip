// Batched, device-resident Pipeline::processInstance (/root/reference/src/pipeline.cpp:25-150) and the
// model preparation of Pipeline::run (:291-294).  Pure composition of the device pipelines: every
// intermediate (cloud, voxels, normals, FPFH, correspondences) stays in the ctx workspace; per instance
// the host only reads three scalars (point count, voxel count, results).  SURVEY.md 8f N1.
#include "tdv_internal.hpp"
#include <cstring>

namespace tdv {

int register_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_bgr, const uint8_t* d_masks, int n_instances,
                       const tdv_batch_params* prm, const float* d_model_xyz, const float* d_model_normals,
                       const float* d_model_fpfh, int n_model, tdv_instance_result* results) {
    if (!ctx || !d_raw || !d_masks || !prm || !results || n_instances < 0 || n_model < 0) return TDV_ERR_BAD_ARG;
    if (n_model > 0 && (!d_model_xyz || !d_model_fpfh)) return TDV_ERR_BAD_ARG;
    const size_t npx = (size_t)prm->width * prm->height;
    for (int b = 0; b < n_instances; ++b) {
        tdv_instance_result& r = results[b];
        std::memset(&r, 0, sizeof(r));
        for (int i = 0; i < 16; ++i) r.T[i] = (i % 5 == 0) ? 1.f : 0.f;
        const WsMark mark = ws_mark(ctx);
        const uint8_t* mask = d_masks + (size_t)b * npx;
        // pass 1: count (capacity 0) so that the cloud buffer is sized to the instance, not the frame
        int n = 0;
        int st = depth_to_cloud_dev(ctx, d_raw, nullptr, mask, nullptr, prm->width, prm->height, prm->scale_to_meters, prm->mask_mode,
                                    prm->fx, prm->fy, prm->cx, prm->cy, prm->zmax, nullptr, nullptr, 0, &n);
        if (st != TDV_OK && st != TDV_ERR_BAD_ARG) return st;  // BAD_ARG here only means "capacity 0 < n"
        r.n_points = n;
        if (n == 0) { r.status = 2; ws_rewind(ctx, mark); continue; }
        float *xyz, *rgb = nullptr;
        TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &xyz));
        if (d_bgr) TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &rgb));
        TDV_TRY(depth_to_cloud_dev(ctx, d_raw, nullptr, mask, d_bgr, prm->width, prm->height, prm->scale_to_meters, prm->mask_mode,
                                   prm->fx, prm->fy, prm->cx, prm->cy, prm->zmax, xyz, rgb, n, &n));
        float* vx; int v = 0;
        TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &vx));
        TDV_TRY(voxel_downsample_dev(ctx, xyz, nullptr, n, prm->voxel_size, TDV_VOXEL_ORDER_FIRST, nullptr, vx, nullptr, n, &v));
        r.n_voxels = v;
        float *nrm, *fpfh; int* corr;
        TDV_TRY(ws_alloc(ctx, (size_t)v * 3, &nrm));
        TDV_TRY(ws_alloc(ctx, (size_t)v * 33, &fpfh));
        TDV_TRY(ws_alloc(ctx, (size_t)v, &corr));
        TDV_TRY(normals_fpfh_dev(ctx, vx, v, prm->normals_k, prm->voxel_size * prm->fpfh_radius_factor, nrm, fpfh));
        tdv_ransac_result coarse;
        TDV_TRY(feature_match_dev(ctx, fpfh, v, d_model_fpfh, n_model, corr));
        TDV_TRY(ransac_run_dev(ctx, vx, v, d_model_xyz, n_model, nullptr, nullptr, corr, prm->voxel_size, prm->ransac_max_iterations,
                               prm->ransac_confidence, prm->seed, &coarse, nullptr));
        r.coarse_fitness = coarse.fitness; r.coarse_inliers = coarse.inliers;
        tdv_icp_result fine;
        const float thr = prm->voxel_size * prm->icp_distance_factor;  // pipeline.cpp:104
        TDV_TRY(icp_run_dev(ctx, vx, v, d_model_xyz, d_model_normals, n_model, coarse.T, thr, prm->icp_max_iterations, prm->point_to_plane, 0, &fine));
        std::memcpy(r.T, fine.T, 64);
        r.fitness = fine.fitness; r.rmse = fine.rmse; r.icp_iterations = fine.iterations;
        r.status = 0;
        ws_rewind(ctx, mark);
    }
    return TDV_OK;
}

}  // namespace tdv

using namespace tdv;

extern "C" {

int tdv_register_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_bgr, const uint8_t* d_masks, int n_instances,
                           const tdv_batch_params* prm, const float* d_model_xyz, const float* d_model_normals,
                           const float* d_model_fpfh, int n_model, tdv_instance_result* results) {
    if (!ctx) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    TDV_TRY(ws_reset(ctx));
    return register_batch_dev(ctx, d_raw, d_bgr, d_masks, n_instances, prm, d_model_xyz, d_model_normals, d_model_fpfh, n_model, results);
}

int tdv_prepare_model_dev(tdv_ctx* ctx, const float* d_xyz, int n, float voxel_size, int normals_k, float fpfh_radius_factor,
                          float* d_out_xyz, float* d_out_normals, float* d_out_fpfh, int* n_out) {
    if (!ctx || !n_out || n < 0 || (n > 0 && (!d_xyz || !d_out_xyz || !d_out_normals || !d_out_fpfh))) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    TDV_TRY(ws_reset(ctx));
    *n_out = 0;
    if (n == 0) return TDV_OK;
    int v = 0;
    TDV_TRY(voxel_downsample_dev(ctx, d_xyz, nullptr, n, voxel_size, TDV_VOXEL_ORDER_FIRST, nullptr, d_out_xyz, nullptr, n, &v));
    TDV_TRY(normals_fpfh_dev(ctx, d_out_xyz, v, normals_k, voxel_size * fpfh_radius_factor, d_out_normals, d_out_fpfh));
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = v;
    return TDV_OK;
}

}  // extern "C"
