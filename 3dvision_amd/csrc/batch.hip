// Batched, device-resident Pipeline::processInstance (/root/reference/src/pipeline.cpp:25-150) and the
// model preparation of Pipeline::run (:291-294).  Pure composition of the device pipelines: every
// intermediate (cloud, voxels, normals, FPFH, correspondences) stays in the ctx workspace; per instance
// the host only reads three scalars (point count, voxel count, results).  SURVEY.md 8f N1.
#include "tdv_internal.hpp"
#include <algorithm>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace tdv {

__global__ void k_gather_i32(const int* __restrict__ in, const int* __restrict__ idx, int n, int* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = in[idx[p]];
}

int register_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_bgr, const uint8_t* d_masks, int n_instances,
                       const tdv_batch_params* prm, const float* d_model_xyz, const float* d_model_normals,
                       const float* d_model_fpfh, int n_model, tdv_instance_result* results) {
    if (!ctx || !d_raw || !d_masks || !prm || !results || n_instances < 0 || n_model < 0) return TDV_ERR_BAD_ARG;
    if (n_model > 0 && (!d_model_xyz || !d_model_fpfh)) return TDV_ERR_BAD_ARG;
    if (n_instances == 0) return TDV_OK;
    // all clouds of the frame in two launches (count + emit), back to back in one buffer
    if (prm->voxel_order != TDV_VOXEL_ORDER_FIRST && prm->voxel_order != TDV_VOXEL_ORDER_REFERENCE) return TDV_ERR_BAD_ARG;
    std::vector<int> off((size_t)n_instances + 1, 0);
    int* d_off = nullptr;
    const int* d_frame_of = nullptr;
    TDV_TRY(frame_map_dev(ctx, n_instances, prm->n_frames, prm->frame_of_instance, &d_frame_of));
    TDV_TRY(depth_to_cloud_batch_count(ctx, d_raw, d_frame_of, d_masks, n_instances, 1, prm->width, prm->height, prm->scale_to_meters, prm->mask_mode,
                                       prm->zmax, &d_off, off.data()));
    float *all_xyz = nullptr;
    if (off[n_instances] > 0) {
        TDV_TRY(ws_alloc(ctx, (size_t)off[n_instances] * 3, &all_xyz));
        TDV_TRY(depth_to_cloud_batch_emit(ctx, d_raw, d_frame_of, d_masks, nullptr, n_instances, 1, prm->width, prm->height, prm->scale_to_meters,
                                          prm->mask_mode, prm->fx, prm->fy, prm->cx, prm->cy, prm->zmax, d_off, all_xyz, nullptr));
    }
    (void)d_bgr;  // colours do not enter the registration chain (voxelDownsample keeps them, nothing downstream reads them)
    // the model's descriptors are packed once; every instance, on either lane, searches the same read-only index
    FmIndex model_index; bool have_index = false;
    if (n_model >= 2048 && !getenv("TDV_FM_BRUTE") && !getenv("TDV_FM_KEYORDER")) {
        TDV_TRY(fm_index_build(ctx, d_model_fpfh, n_model, &model_index));
        have_index = true;
    }
    // likewise the model's Morton order and boxes for the ICP correspondence search (built when ICP will take the pruned path)
    // (the hash grid at this call's ICP threshold first; the Morton order only if the grid cannot be used)
    const float icp_thr = prm->voxel_size * prm->icp_distance_factor;  // pipeline.cpp:104
    CellGrid model_grid{}; bool have_grid = false;
    SortedCloud model_sorted{}; bool have_sorted = false;
    if (n_model >= 4096 && ctx->icp_search != TDV_ICP_SEARCH_BRUTE) {
        if (ctx->icp_search != TDV_ICP_SEARCH_PRUNED) {
            TDV_TRY(cell_grid_build(ctx, d_model_xyz, n_model, icp_thr, &model_grid));
            have_grid = true;                     // (usable or not: ICP looks at the flag and does not build its own)
        }
        if (!model_grid.usable) {
            TDV_TRY(spatial_sort_cloud(ctx, d_model_xyz, n_model, model_sorted));
            have_sorted = true;
        }
    }
    static const bool coherent_stages = !(getenv("TDV_BATCH_COHERENT") && atoi(getenv("TDV_BATCH_COHERENT")) == 0);   // A/B knob
    // one instance: voxel -> normals + FPFH -> match -> RANSAC -> ICP on context c (its stream, its workspace)
    auto run_instance = [&](tdv_ctx* c, int b) -> int {
        tdv_instance_result& r = results[b];
        std::memset(&r, 0, sizeof(r));
        for (int i = 0; i < 16; ++i) r.T[i] = (i % 5 == 0) ? 1.f : 0.f;
        const WsMark mark = ws_mark(c);
        int n = off[b + 1] - off[b];
        r.n_points = n;
        if (n == 0) { r.status = 2; ws_rewind(c, mark); return TDV_OK; }
        float* xyz = all_xyz + (size_t)off[b] * 3;
        float* vx; int v = 0;
        TDV_TRY(ws_alloc(c, (size_t)n * 3, &vx));
        // The reference's container order scatters neighbouring voxels over the whole array, which costs the per-point stages
        // (radius search, SPFH / FPFH gathers, descriptor search) their locality.  Those stages are per point: they run on the
        // same voxels in first-occurrence order (image order: coherent), with every neighbour list ordered by the reference
        // POSITIONS of its members — so each point's normal, descriptor and match carry the bits they have in the reference
        // order — and only the correspondences are permuted.  RANSAC (which draws points by position) and ICP (whose sums
        // run over positions) see the cloud in the reference's order.
        const bool coherent = prm->voxel_order == TDV_VOXEL_ORDER_REFERENCE && coherent_stages && prm->normals_k <= 100;
        VoxelBothOrders both{nullptr, nullptr, nullptr};
        if (coherent) {
            TDV_TRY(ws_alloc(c, (size_t)n * 3, &both.first_xyz));
            TDV_TRY(ws_alloc(c, (size_t)n, &both.ref2first));
            TDV_TRY(ws_alloc(c, (size_t)n, &both.first2ref));
        }
        TDV_TRY(voxel_downsample_dev(c, xyz, nullptr, n, prm->voxel_size, prm->voxel_order, vx, nullptr, n, &v, coherent ? &both : nullptr));
        r.n_voxels = v;
        float *nrm, *fpfh; int* corr;
        TDV_TRY(ws_alloc(c, (size_t)v * 3, &nrm));
        TDV_TRY(ws_alloc(c, (size_t)v * 33, &fpfh));
        TDV_TRY(ws_alloc(c, (size_t)v, &corr));
        int* corr_stage = corr;
        if (coherent) TDV_TRY(ws_alloc(c, (size_t)v, &corr_stage));
        TDV_TRY(normals_fpfh_dev(c, coherent ? both.first_xyz : vx, v, prm->normals_k, prm->voxel_size * prm->fpfh_radius_factor, nrm, fpfh,
                                 coherent ? both.first2ref : nullptr, coherent ? both.ref2first : nullptr));
        tdv_ransac_result coarse;
        if (have_index && v >= 4096) TDV_TRY(feature_match_indexed_dev(c, fpfh, v, model_index, corr_stage));
        else TDV_TRY(feature_match_dev(c, fpfh, v, d_model_fpfh, n_model, corr_stage));
        if (coherent && v > 0) {
            k_gather_i32<<<(v + 255) / 256, 256, 0, c->stream>>>(corr_stage, both.ref2first, v, corr);
            TDV_CHECK_LAUNCH(c);
        }
        TDV_TRY(ransac_run_dev(c, vx, v, d_model_xyz, n_model, nullptr, nullptr, corr, prm->voxel_size, prm->ransac_max_iterations,
                               prm->ransac_confidence, prm->seed, &coarse, nullptr));
        r.coarse_fitness = coarse.fitness; r.coarse_inliers = coarse.inliers;
        tdv_icp_result fine;
        TDV_TRY(icp_run_dev(c, vx, v, d_model_xyz, d_model_normals, n_model, coarse.T, icp_thr, prm->icp_max_iterations, prm->point_to_plane, 0, &fine,
                            have_sorted ? &model_sorted : nullptr, have_grid ? &model_grid : nullptr));
        std::memcpy(r.T, fine.T, 64);
        r.fitness = fine.fitness; r.rmse = fine.rmse; r.icp_iterations = fine.iterations;
        r.status = 0;
        ws_rewind(c, mark);
        return TDV_OK;
    };
    // Lanes: the calling thread on ctx plus helper threads, each on its own helper ctx (stream + workspace, owned by ctx and
    // chained through ->helper), take the instances in turn, so that one lane's host syncs, its host replay of the voxel
    // order and its small kernels overlap the others' work — the shape of the reference's thread pool
    // (src/pipeline.cpp:321-327), inside one call.  Results do not depend on the lanes.  Measured on C4 (96 instances,
    // instances/s): the reference's voxel order, whose container replay is 3 ms of host time per instance, 1 / 2 / 3 / 4 / 6
    // lanes: 125 / 184 / 211 / 226 / 252; first-occurrence order 2 / 3 / 4 lanes: 265 / 275 / 260 — and at the end of round
    // 2, with every kernel shorter: reference order 4 / 6 / 8 / 10 lanes 395 / 439 / 466 / 463, first-occurrence order
    // 3 / 4 / 6 lanes 525 / 514 / 532.  Hence 8 resp. 3 lanes by default; TDV_BATCH_LANES overrides (1 = the caller's
    // thread only, at most 16).
    static const int lanes_env = getenv("TDV_BATCH_LANES") ? atoi(getenv("TDV_BATCH_LANES")) : 0;
    const int lanes_default = prm->voxel_order == TDV_VOXEL_ORDER_REFERENCE ? 12 : 3;   // (10 / 12 / 14 / 16 lanes at the end of round 2: 502-513 / 508-518 / 509-519 / 507-518 instances/s)
    const int want = std::max(1, std::min(std::min(lanes_env > 0 ? lanes_env : lanes_default, 16), n_instances));
    std::vector<tdv_ctx*> lane_ctx{ctx};
    for (tdv_ctx* c = ctx; (int)lane_ctx.size() < want; c = c->helper) {
        if (!c->helper && tdv_ctx_create(ctx->device, &c->helper) != TDV_OK) { c->helper = nullptr; break; }
        lane_ctx.push_back(c->helper);
    }
    const int L = (int)lane_ctx.size();
    if (L == 1) {
        for (int b = 0; b < n_instances; ++b) TDV_TRY(run_instance(ctx, b));
        return TDV_OK;
    }
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the clouds and the model's index are complete before other streams read them
    std::vector<int> status((size_t)L, TDV_OK);
    std::vector<std::thread> workers;
    for (int l = 1; l < L; ++l) {
        tdv_ctx* h = lane_ctx[l];
        h->timing = ctx->timing; h->icp_search = ctx->icp_search; h->ransac_score_mode = ctx->ransac_score_mode; h->err[0] = 0;
        workers.emplace_back([&, h, l]() {
            try {
                if (hipSetDevice(h->device) != hipSuccess) { status[l] = TDV_ERR_NO_DEVICE; return; }
                status[l] = ws_reset(h);
                for (int b = l; b < n_instances && status[l] == TDV_OK; b += L) status[l] = run_instance(h, b);
                if (status[l] == TDV_OK && hipStreamSynchronize(h->stream) != hipSuccess) status[l] = TDV_ERR_LAUNCH;
            } catch (...) {   // nothing may escape a thread
                std::snprintf(h->err, sizeof(h->err), "exception in a helper lane");
                status[l] = TDV_ERR_INTERNAL;
            }
        });
    }
    try {
        for (int b = 0; b < n_instances && status[0] == TDV_OK; b += L) status[0] = run_instance(ctx, b);
    } catch (...) {   // the workers must be joined whatever happens here
        std::snprintf(ctx->err, sizeof(ctx->err), "exception in the batch lane");
        status[0] = TDV_ERR_INTERNAL;
    }
    for (auto& w : workers) w.join();
    if (status[0] != TDV_OK) return status[0];
    for (int l = 1; l < L; ++l)
        if (status[l] != TDV_OK) { std::snprintf(ctx->err, sizeof(ctx->err), "batch lane %d: %s", l + 1, lane_ctx[l]->err); return status[l]; }
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

int tdv_depth_to_cloud_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_masks, const uint8_t* d_bgr,
                                 int n_instances, int mask_format, int n_frames, const int* h_frame_of_instance,
                                 int width, int height, float scale, int mask_mode,
                                 float fx, float fy, float cx, float cy, float zmax,
                                 float* d_xyz, float* d_rgb, long long capacity, int* h_offsets) {
    if (!ctx || !d_raw || !d_masks || !h_offsets || n_instances < 0 || width < 0 || height < 0 || capacity < 0) return TDV_ERR_BAD_ARG;
    if (mask_format != 0 && n_frames > 1) return TDV_ERR_BAD_ARG;   // a label image belongs to one frame
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    TDV_TRY(ws_reset(ctx));
    h_offsets[0] = 0;
    if (n_instances == 0 || (size_t)width * height == 0) { for (int b = 0; b <= n_instances; ++b) h_offsets[b] = 0; return TDV_OK; }
    int* d_off = nullptr;
    const int* d_frame_of = nullptr;
    TDV_TRY(frame_map_dev(ctx, n_instances, n_frames, h_frame_of_instance, &d_frame_of));
    TDV_TRY(depth_to_cloud_batch_count(ctx, d_raw, d_frame_of, d_masks, n_instances, mask_format == 0, width, height, scale, mask_mode, zmax, &d_off, h_offsets));
    if ((long long)h_offsets[n_instances] > capacity || (h_offsets[n_instances] > 0 && !d_xyz)) return TDV_ERR_BAD_ARG;
    if (h_offsets[n_instances] > 0)
        TDV_TRY(depth_to_cloud_batch_emit(ctx, d_raw, d_frame_of, d_masks, d_bgr, n_instances, mask_format == 0, width, height, scale, mask_mode,
                                          fx, fy, cx, cy, zmax, d_off, d_xyz, d_rgb));
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TDV_OK;
}

int tdv_prepare_model_dev(tdv_ctx* ctx, const float* d_xyz, int n, float voxel_size, int voxel_order, int normals_k, float fpfh_radius_factor,
                          float* d_out_xyz, float* d_out_normals, float* d_out_fpfh, int* n_out) {
    if (!ctx || !n_out || n < 0 || (n > 0 && (!d_xyz || !d_out_xyz || !d_out_normals || !d_out_fpfh))) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    TDV_TRY(ws_reset(ctx));
    *n_out = 0;
    if (n == 0) return TDV_OK;
    int v = 0;
    TDV_TRY(voxel_downsample_dev(ctx, d_xyz, nullptr, n, voxel_size, voxel_order, d_out_xyz, nullptr, n, &v));
    TDV_TRY(normals_fpfh_dev(ctx, d_out_xyz, v, normals_k, voxel_size * fpfh_radius_factor, d_out_normals, d_out_fpfh));
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = v;
    return TDV_OK;
}

}  // extern "C"
