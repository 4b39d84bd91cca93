// Batched, device-resident Pipeline::processInstance (/root/reference/src/pipeline.cpp:25-150) and the
// model preparation of Pipeline::run (:291-294).  Pure composition of the device pipelines: every
// intermediate (cloud, voxels, normals, FPFH, correspondences) stays in the ctx workspace; per instance
// the host only reads three scalars (point count, voxel count, results).  SURVEY.md 8f N1.
#include "tdv_internal.hpp"
#include <algorithm>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace tdv {

// out[P] = local[P] + (first voxel of the instance P belongs to): positions inside an instance -> positions in the batch's arrays
__global__ void k_globalise(const int* __restrict__ local, const int* __restrict__ voff, int n_inst, int total, int* __restrict__ out) {
    const int P = blockIdx.x * blockDim.x + threadIdx.x;
    if (P >= total) return;
    int a = 0, z = n_inst;
    while (z - a > 1) { const int m = (a + z) >> 1; if (voff[m] <= P) a = m; else z = m; }
    out[P] = local[P] + voff[a];
}

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
    if (prm->mask_format < 0 || prm->mask_format > 2) return TDV_ERR_BAD_ARG;
    if (prm->mask_format != 0 && prm->n_frames > 1) return TDV_ERR_BAD_ARG;                     // a label image belongs to one frame
    if (prm->mask_format == 1 && n_instances > 255) return TDV_ERR_BAD_ARG;                     // u8 labels: instance b is label b + 1
    if (prm->mask_format == 2 && n_instances > 65535) return TDV_ERR_BAD_ARG;
    const int layout = prm->mask_format == 0 ? 1 : (prm->mask_format == 1 ? 0 : 2);            // depth.hip's `stacked` argument
    // masks of another size than the frame: nearest-neighbour resize first (src/pipeline.cpp:38-41)
    const int mw = prm->mask_width > 0 ? prm->mask_width : prm->width, mh = prm->mask_height > 0 ? prm->mask_height : prm->height;
    if (mw != prm->width || mh != prm->height) {
        if (prm->mask_format == 2) return TDV_ERR_BAD_ARG;                                      // (a u16 label image is never resized: no such input in the reference)
        uint8_t* resized;
        const int n_masks = prm->mask_format == 0 ? n_instances : 1;
        TDV_TRY(ws_alloc(ctx, (size_t)n_masks * prm->width * prm->height + 16, &resized));
        resized = reinterpret_cast<uint8_t*>(align_up(reinterpret_cast<uintptr_t>(resized), 16));
        TDV_TRY(mask_resize_nearest_dev(ctx, d_masks, n_masks, mw, mh, prm->width, prm->height, resized));
        d_masks = resized;
    }
    std::vector<int> off((size_t)n_instances + 1, 0);
    int* d_off = nullptr;
    const int* d_frame_of = nullptr;
    TDV_TRY(frame_map_dev(ctx, n_instances, prm->n_frames, prm->frame_of_instance, &d_frame_of));
    TDV_TRY(depth_to_cloud_batch_count(ctx, d_raw, d_frame_of, d_masks, n_instances, layout, prm->width, prm->height, prm->scale_to_meters, prm->mask_mode,
                                       prm->zmax, &d_off, off.data()));
    float *all_xyz = nullptr;
    if (off[n_instances] > 0) {
        TDV_TRY(ws_alloc(ctx, (size_t)off[n_instances] * 3, &all_xyz));
        TDV_TRY(depth_to_cloud_batch_emit(ctx, d_raw, d_frame_of, d_masks, nullptr, n_instances, layout, prm->width, prm->height, prm->scale_to_meters,
                                          prm->mask_mode, prm->fx, prm->fy, prm->cx, prm->cy, prm->zmax, d_off, all_xyz, nullptr));
    }
    // instances without a point: status 1 when the masked depth image holds no non-zero value at all (pipeline.cpp:57-60),
    // status 2 when it does but nothing survives the z clip (:86-89)
    std::vector<int> empty_inst, empty_nonzero;
    for (int b = 0; b < n_instances; ++b) if (off[b + 1] == off[b]) empty_inst.push_back(b);
    empty_nonzero.assign(empty_inst.size(), 0);
    TDV_TRY(depth_batch_nonzero_any(ctx, d_raw, d_frame_of, d_masks, layout, prm->width, prm->height, prm->scale_to_meters, prm->mask_mode,
                                    empty_inst.data(), (int)empty_inst.size(), empty_nonzero.data()));
    std::vector<int> empty_status((size_t)n_instances, 0);
    for (size_t e = 0; e < empty_inst.size(); ++e) empty_status[empty_inst[e]] = empty_nonzero[e] ? 2 : 1;
    (void)d_bgr;  // colours do not enter the registration chain (voxelDownsample keeps them, nothing downstream reads them)
    // the model's descriptors are packed once; every instance, on either lane, searches the same read-only index
    FmIndex model_index; bool have_index = false;
    if (n_model >= 2048 && !getenv("TDV_FM_BRUTE") && !study_env("TDV_FM_KEYORDER")) {
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
    const bool coherent_stages = !(study_env("TDV_BATCH_COHERENT") && atoi(study_env("TDV_BATCH_COHERENT")) == 0);   // A/B knob (read per call: the tests switch it)
    // voxels of ALL instances in first-occurrence order with one memset + two launches (voxel.hip, hash-table path); a lane then
    // only finishes its instance's reference order.  A voxel too full for the table's member rows (a very coarse grid) sends the
    // whole batch back to per-instance calls.
    const bool batched_voxel_env = !(getenv("TDV_BATCH_VOXEL") && atoi(getenv("TDV_BATCH_VOXEL")) == 0);          // A/B knob (read per call: the tests switch it)
    const bool want_ref = prm->voxel_order == TDV_VOXEL_ORDER_REFERENCE;
    std::vector<int> voff((size_t)n_instances + 1, 0);
    float* vox_first_all = nullptr; int* vox_rank_all = nullptr; int4* vox_leaders_all = nullptr;
    float* vox_ref_all = nullptr; int *vox_r2f_all = nullptr, *vox_f2r_all = nullptr;
    std::vector<int> ref_failed((size_t)n_instances, 1);        // 1: this instance's reference order is (still) to be made by the host replay
    int* d_voff = nullptr;                                      // voxel offsets of the instances on the device (n_instances + 2 ints)
    bool batched_voxel = false;
    const int total_pts = off[n_instances];
    if (batched_voxel_env && total_pts > 0 && !study_env("TDV_VOXEL_LEGACY") && !study_env("TDV_VOXEL_SORT")) {
        TDV_TRY(ws_alloc(ctx, (size_t)total_pts * 3, &vox_first_all));
        if (want_ref) { TDV_TRY(ws_alloc(ctx, (size_t)total_pts, &vox_rank_all)); TDV_TRY(ws_alloc(ctx, (size_t)total_pts, &vox_leaders_all)); }
        int* d_off_inst;
        TDV_TRY(ws_alloc(ctx, (size_t)n_instances + 1, &d_off_inst));
        TDV_TRY(ws_alloc(ctx, (size_t)n_instances + 2, &d_voff));
        TDV_HIP(ctx, hipMemcpyAsync(d_off_inst, off.data(), ((size_t)n_instances + 1) * 4, hipMemcpyHostToDevice, ctx->stream));   // (off outlives the call's sync below)
        const WsMark vmark = ws_mark(ctx);                   // the table and member rows are scratch: given back after the call
        int overflowed = 0;
        const float pinhole[4] = {prm->fx, prm->fy, prm->cx, prm->cy};     // the clouds come from this call's own unprojection: row-major pixel order, these intrinsics
        TDV_TRY(voxel_downsample_batch_dev(ctx, all_xyz, total_pts, d_off_inst, n_instances, prm->voxel_size, vox_first_all, vox_rank_all, vox_leaders_all,
                                           voff.data(), &overflowed, d_voff, pinhole, off.data()));
        ws_rewind(ctx, vmark);
        batched_voxel = !overflowed;
        // ... and their reference order, on the device too (the host replay stays as the fall-back of a cloud whose hash degenerates)
        const bool device_order_env = !(getenv("TDV_VOXEL_DEVICE_ORDER") && atoi(getenv("TDV_VOXEL_DEVICE_ORDER")) == 0);   // A/B knob (read per call: the tests switch it)
        if (batched_voxel && want_ref && device_order_env && voff[n_instances] > 0) {
            const size_t tv = (size_t)voff[n_instances];
            TDV_TRY(ws_alloc(ctx, tv * 3, &vox_ref_all));
            TDV_TRY(ws_alloc(ctx, tv, &vox_r2f_all));
            TDV_TRY(ws_alloc(ctx, tv, &vox_f2r_all));
            TDV_TRY(voxel_reference_order_batch_dev(ctx, n_instances, voff.data(), d_voff, vox_leaders_all, vox_first_all, vox_ref_all, vox_r2f_all, vox_f2r_all,
                                                    ref_failed.data()));
        }
    }
    // one instance: voxel -> normals + FPFH -> match -> RANSAC -> ICP on context c (its stream, its workspace)
    auto run_instance = [&](tdv_ctx* c, int b) -> int {
        tdv_instance_result& r = results[b];
        std::memset(&r, 0, sizeof(r));
        for (int i = 0; i < 16; ++i) r.T[i] = (i % 5 == 0) ? 1.f : 0.f;
        const WsMark mark = ws_mark(c);
        int n = off[b + 1] - off[b];
        r.n_points = n;
        if (n == 0) { r.status = empty_status[b]; ws_rewind(c, mark); return TDV_OK; }
        float* xyz = all_xyz + (size_t)off[b] * 3;
        float* vx; int v = 0;
        // The reference's container order scatters neighbouring voxels over the whole array, which costs the per-point stages
        // (radius search, SPFH / FPFH gathers, descriptor search) their locality.  Those stages are per point: they run on the
        // same voxels in first-occurrence order (image order: coherent), with every neighbour list ordered by the reference
        // POSITIONS of its members — so each point's normal, descriptor and match carry the bits they have in the reference
        // order — and only the correspondences are permuted.  RANSAC (which draws points by position) and ICP (whose sums
        // run over positions) see the cloud in the reference's order.
        const bool coherent = want_ref && coherent_stages && prm->normals_k <= 100;
        VoxelBothOrders both{nullptr, nullptr, nullptr};
        if (batched_voxel) {
            v = voff[b + 1] - voff[b];
            float* first = vox_first_all + (size_t)voff[b] * 3;
            if (!want_ref) vx = first;
            else if (vox_ref_all && !ref_failed[b]) {           // ordered on the device with the rest of the batch
                vx = vox_ref_all + (size_t)voff[b] * 3;
                if (coherent) both = VoxelBothOrders{first, vox_r2f_all + voff[b], vox_f2r_all + voff[b]};
            } else {
                TDV_TRY(ws_alloc(c, (size_t)v * 3, &vx));
                if (coherent) {
                    both.first_xyz = first;
                    TDV_TRY(ws_alloc(c, (size_t)v, &both.ref2first));
                    TDV_TRY(ws_alloc(c, (size_t)v, &both.first2ref));
                }
                TDV_TRY(voxel_reference_order(c, v, n, vox_leaders_all + voff[b], first, nullptr, vox_rank_all + off[b], voff[b], vx, nullptr,
                                              coherent ? &both : nullptr));
            }
        } else {
            TDV_TRY(ws_alloc(c, (size_t)n * 3, &vx));
            if (coherent) {
                TDV_TRY(ws_alloc(c, (size_t)n * 3, &both.first_xyz));
                TDV_TRY(ws_alloc(c, (size_t)n, &both.ref2first));
                TDV_TRY(ws_alloc(c, (size_t)n, &both.first2ref));
            }
            TDV_TRY(voxel_downsample_dev(c, xyz, nullptr, n, prm->voxel_size, prm->voxel_order, vx, nullptr, n, &v, coherent ? &both : nullptr));
        }
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
    // chained through ->helper), take the instances in turn, so that one lane's host syncs and its small kernels overlap the
    // others' work — the shape of the reference's thread pool (src/pipeline.cpp:321-327), inside one call.  Results do not
    // depend on the lanes.  Round 2 needed 12 lanes in the reference's voxel order to hide 3 ms of host replay per instance
    // (395 / 466 / 513 instances/s of C4 with 4 / 8 / 12 lanes); with the order made on the device large instances no longer
    // care much (round 3, C4, 256 instances: 613 / 603-628 / 631-667 / 625-651 with 2 / 4 / 6 / 8 lanes once the kernels of an instance
    // had shrunk to 1.6 ms; the GPU is busy either way; round 4, three runs each on one box, instance kernels at 1.4 ms: 645-653 / 667-673 /
    // 632-638 / 660-676 with 2 / 3 / 4 / 6 lanes) and take 3: the throughput of 6 with half the host threads and half the lane
    // workspaces (8 caller threads x 6 lanes were 48 threads and ~10 GB of workspace per caller on a real host).  Small instances are chains of
    // launch-bound kernels and host read-backs and still gain from more lanes (C5, 1,024 instances of ~400 voxels:
    // 5,650 / 6,550 / 6,790 instances/s with 8 / 12 / 16): they take 12.  Never more than the host has hardware threads;
    // TDV_BATCH_LANES overrides (1 = the caller's thread only, at most 16).
    const int lanes_env = getenv("TDV_BATCH_LANES") ? atoi(getenv("TDV_BATCH_LANES")) : 0;   // read per call, like the other knobs (the tests switch it)
    const int hw_threads = std::max(1u, std::thread::hardware_concurrency());
    const bool small_instances = off[n_instances] / std::max(n_instances, 1) < 8192;      // points per instance, on average
    const int lanes_default = std::min(small_instances ? 12 : 3, hw_threads);
    const int want = std::max(1, std::min(std::min(lanes_env > 0 ? lanes_env : lanes_default, 16), n_instances));
    std::vector<tdv_ctx*> lane_ctx{ctx};
    for (tdv_ctx* c = ctx; (int)lane_ctx.size() < want; c = c->helper) {
        if (!c->helper && tdv_ctx_create(ctx->device, &c->helper) != TDV_OK) { c->helper = nullptr; break; }
        lane_ctx.push_back(c->helper);
    }
    const int L = (int)lane_ctx.size();
    ctx->last_batch_lanes = L;
    // fn(lane ctx, instance) over all instances, dealt to the lanes in turn; returns when every lane's stream has drained
    auto for_all_instances = [&](const std::function<int(tdv_ctx*, int)>& fn) -> int {
        if (L == 1) {
            int st = TDV_OK;
            for (int b = 0; b < n_instances && st == TDV_OK; ++b) st = fn(ctx, b);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == TDV_OK) st = TDV_ERR_LAUNCH;   // nothing of this pass is in flight when it returns
            return st;
        }
        TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));   // what the pass reads is complete before other streams read it
        std::vector<int> status((size_t)L, TDV_OK);
        std::vector<std::thread> workers;
        for (int l = 1; l < L; ++l) {
            tdv_ctx* h = lane_ctx[l];
            h->timing = ctx->timing; h->icp_search = ctx->icp_search; h->icp_accumulate = ctx->icp_accumulate; h->ransac_score_mode = ctx->ransac_score_mode; h->err[0] = 0;
            workers.emplace_back([&, h, l]() {
                try {
                    if (hipSetDevice(h->device) != hipSuccess) { status[l] = TDV_ERR_NO_DEVICE; return; }
                    status[l] = ws_reset(h);
                    for (int b = l; b < n_instances && status[l] == TDV_OK; b += L) status[l] = fn(h, b);
                    // on failure too: kernels of this lane may still be reading all_xyz and the workspaces the next call reuses
                    if (hipStreamSynchronize(h->stream) != hipSuccess && status[l] == TDV_OK) status[l] = TDV_ERR_LAUNCH;
                } catch (...) {   // nothing may escape a thread
                    std::snprintf(h->err, sizeof(h->err), "exception in a helper lane");
                    status[l] = TDV_ERR_INTERNAL;
                }
            });
        }
        try {
            for (int b = 0; b < n_instances && status[0] == TDV_OK; b += L) status[0] = fn(ctx, b);
        } catch (...) {   // the workers must be joined whatever happens here
            std::snprintf(ctx->err, sizeof(ctx->err), "exception in the batch lane");
            status[0] = TDV_ERR_INTERNAL;
        }
        for (auto& w : workers) w.join();
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && status[0] == TDV_OK) status[0] = TDV_ERR_LAUNCH;
        if (status[0] != TDV_OK) return status[0];
        for (int l = 1; l < L; ++l)
            if (status[l] != TDV_OK) { std::snprintf(ctx->err, sizeof(ctx->err), "batch lane %d: %s", l + 1, lane_ctx[l]->err); return status[l]; }
        return TDV_OK;
    };

    // STAGED batch (small instances whose voxels sit in the batch's own arrays): instead of walking each instance through the
    // whole chain, the descriptor match runs ONCE for all instances:
    //   voxels + reference order (above)  ->  [lanes] normals + FPFH per instance, descriptors into one array
    //   ->  ONE descriptor match for all instances' points against the model  ->  [lanes] RANSAC + ICP per instance.
    // Every point's nearest model descriptor is the same whoever else is in the call (an exact search), so the results are those
    // of the instance-by-instance chain, bit for bit (tests/test_gpu_chain.py, test_gpu_configs.py, test_gpu_c5.py).  For C5's
    // 1,024 instances of ~400 voxels the per-instance match was the largest item of an instance (2 launches, 167 us of 600 us of
    // kernels): 4,716 -> 6,554 instances/s at 12 lanes.  For C4's 147k-voxel instances it LOSES (497 -> 469 instances/s at 4 lanes,
    // 429 at 12): there the lanes overlap kernels of different kinds, which the common stages take away - hence the size rule
    // (TDV_BATCH_STAGED=0 / 1 forces either shape).
    const int staged_env = getenv("TDV_BATCH_STAGED") ? atoi(getenv("TDV_BATCH_STAGED")) : -1;   // A/B knob (read per call: the tests switch it)
    bool staged = (staged_env < 0 ? small_instances : staged_env != 0) && batched_voxel && voff[n_instances] > 0;
    if (staged && want_ref) { if (!vox_ref_all) staged = false; else for (int b = 0; b < n_instances; ++b) if (ref_failed[b] && off[b + 1] > off[b]) staged = false; }
    if (!staged) return for_all_instances(run_instance);

    const size_t tv = (size_t)voff[n_instances];
    const bool coherent = want_ref && coherent_stages && prm->normals_k <= 100;
    float* fpfh_all; int* corr_all;
    TDV_TRY(ws_alloc(ctx, tv * 33, &fpfh_all));
    TDV_TRY(ws_alloc(ctx, tv, &corr_all));
    auto instance_clouds = [&](int b, float*& vx, const float*& stage_xyz) {
        float* first = vox_first_all + (size_t)voff[b] * 3;
        vx = want_ref ? vox_ref_all + (size_t)voff[b] * 3 : first;
        stage_xyz = coherent ? first : vx;
    };
    auto stage_features = [&](tdv_ctx* c, int b) -> int {
        tdv_instance_result& r = results[b];
        std::memset(&r, 0, sizeof(r));
        for (int i = 0; i < 16; ++i) r.T[i] = (i % 5 == 0) ? 1.f : 0.f;
        const int n = off[b + 1] - off[b], v = voff[b + 1] - voff[b];
        r.n_points = n; r.n_voxels = v;
        if (n == 0) { r.status = empty_status[b]; return TDV_OK; }
        const WsMark mark = ws_mark(c);
        float* vx; const float* stage_xyz;
        instance_clouds(b, vx, stage_xyz);
        float* nrm;
        TDV_TRY(ws_alloc(c, (size_t)v * 3, &nrm));
        TDV_TRY(normals_fpfh_dev(c, stage_xyz, v, prm->normals_k, prm->voxel_size * prm->fpfh_radius_factor, nrm, fpfh_all + (size_t)voff[b] * 33,
                                 coherent ? vox_f2r_all + voff[b] : nullptr, coherent ? vox_r2f_all + voff[b] : nullptr));
        ws_rewind(c, mark);
        return TDV_OK;
    };
    // small instances with at least k points each: normals + FPFH of ALL of them in one set of launches (knn.hip), not ~16
    // launches per instance
    const bool batched_features_env = !(getenv("TDV_BATCH_FEATURES") && atoi(getenv("TDV_BATCH_FEATURES")) == 0);   // A/B knob (read per call: the tests switch it)
    bool batched_features = batched_features_env && small_instances && d_voff && prm->normals_k <= 100;
    int *tie = nullptr, *tie_inv = nullptr;     // coherent stages: global first -> reference position and back
    for (int b = 0; b < n_instances && batched_features; ++b) { const int v = voff[b + 1] - voff[b]; if (v > 0 && v < prm->normals_k) batched_features = false; }
    if (batched_features) {
        for (int b = 0; b < n_instances; ++b) {
            tdv_instance_result& r = results[b];
            std::memset(&r, 0, sizeof(r));
            for (int i = 0; i < 16; ++i) r.T[i] = (i % 5 == 0) ? 1.f : 0.f;
            r.n_points = off[b + 1] - off[b]; r.n_voxels = voff[b + 1] - voff[b];
            if (r.n_points == 0) r.status = empty_status[b];
        }
        if (coherent) {      // positions inside an instance -> positions in the batch's arrays (kept: the RANSAC pass below reorders the match with them)
            TDV_TRY(ws_alloc(ctx, tv, &tie)); TDV_TRY(ws_alloc(ctx, tv, &tie_inv));
            k_globalise<<<(unsigned)((tv + 255) / 256), 256, 0, ctx->stream>>>(vox_f2r_all, d_voff, n_instances, (int)tv, tie);
            k_globalise<<<(unsigned)((tv + 255) / 256), 256, 0, ctx->stream>>>(vox_r2f_all, d_voff, n_instances, (int)tv, tie_inv);
            TDV_CHECK_LAUNCH(ctx);
        }
        const WsMark fmark = ws_mark(ctx);
        float* nrm_all;
        TDV_TRY(ws_alloc(ctx, tv * 3, &nrm_all));
        const float* stage_all = coherent ? vox_first_all : (want_ref ? vox_ref_all : vox_first_all);
        TDV_TRY(normals_fpfh_batch_dev(ctx, stage_all, voff.data(), d_voff, n_instances, prm->normals_k, prm->voxel_size * prm->fpfh_radius_factor,
                                       nrm_all, fpfh_all, tie, tie_inv));
        ws_rewind(ctx, fmark);        // (stream order: the descriptor match below is enqueued behind the kernels that used the scratch)
    } else {
        TDV_TRY(for_all_instances(stage_features));
    }
    if (have_index && tv >= 4096) TDV_TRY(feature_match_indexed_dev(ctx, fpfh_all, (int)tv, model_index, corr_all));
    else TDV_TRY(feature_match_dev(ctx, fpfh_all, (int)tv, d_model_fpfh, n_model, corr_all));
    // small instances and a small model: all ICP refinements in ONE launch (icp.hip: k_icp_small, a workgroup per instance) after the
    // RANSAC pass - same kernel as the per-instance call, same bits
    int v_max = 0;
    for (int b = 0; b < n_instances; ++b) v_max = std::max(v_max, voff[b + 1] - voff[b]);
    const bool icp_small_off = getenv("TDV_ICP_SMALL") && atoi(getenv("TDV_ICP_SMALL")) == 0;   // (read per call: the tests switch it)
    const bool batched_icp = !icp_small_off && d_voff && v_max <= icp_small_max_points() && n_model <= icp_small_max_points() && n_model > 0 &&
                             (long long)v_max * n_model <= icp_small_max_pairs_batch() &&
                             (ctx->icp_search == TDV_ICP_SEARCH_AUTO || ctx->icp_search == TDV_ICP_SEARCH_BRUTE);
    std::vector<float> coarse_T(batched_icp ? (size_t)n_instances * 16 : 0, 0.f);
    auto stage_register = [&](tdv_ctx* c, int b) -> int {
        tdv_instance_result& r = results[b];
        const int v = voff[b + 1] - voff[b];
        if (batched_icp) for (int i = 0; i < 16; ++i) coarse_T[(size_t)b * 16 + i] = (i % 5 == 0) ? 1.f : 0.f;
        if (off[b + 1] == off[b]) return TDV_OK;
        const WsMark mark = ws_mark(c);
        float* vx; const float* stage_xyz;
        instance_clouds(b, vx, stage_xyz);
        int* corr = corr_all + voff[b];
        if (coherent) {                                        // the match ran on the coherent ordering: bring it to reference positions
            TDV_TRY(ws_alloc(c, (size_t)v, &corr));
            k_gather_i32<<<(v + 255) / 256, 256, 0, c->stream>>>(corr_all + voff[b], vox_r2f_all + voff[b], v, corr);
            TDV_CHECK_LAUNCH(c);
        }
        tdv_ransac_result coarse;
        TDV_TRY(ransac_run_dev(c, vx, v, d_model_xyz, n_model, nullptr, nullptr, corr, prm->voxel_size, prm->ransac_max_iterations,
                               prm->ransac_confidence, prm->seed, &coarse, nullptr));
        r.coarse_fitness = coarse.fitness; r.coarse_inliers = coarse.inliers;
        if (batched_icp) { std::memcpy(&coarse_T[(size_t)b * 16], coarse.T, 64); ws_rewind(c, mark); return TDV_OK; }
        tdv_icp_result fine;
        TDV_TRY(icp_run_dev(c, vx, v, d_model_xyz, d_model_normals, n_model, coarse.T, icp_thr, prm->icp_max_iterations, prm->point_to_plane, 0, &fine,
                            have_sorted ? &model_sorted : nullptr, have_grid ? &model_grid : nullptr));
        std::memcpy(r.T, fine.T, 64);
        r.fitness = fine.fitness; r.rmse = fine.rmse; r.icp_iterations = fine.iterations;
        r.status = 0;
        ws_rewind(c, mark);
        return TDV_OK;
    };
    // ... and all coarse alignments in a handful of launches (ransac.hip: ransac_small_batch_dev) when the ICP pass is batched too
    bool ransac_done = false;
    if (batched_icp && d_voff && (!coherent || tie_inv)) {
        int* corr_ref = corr_all;
        if (coherent) {                                        // the match ran on the coherent ordering: bring it to reference positions
            TDV_TRY(ws_alloc(ctx, tv, &corr_ref));
            k_gather_i32<<<(unsigned)((tv + 255) / 256), 256, 0, ctx->stream>>>(corr_all, tie_inv, (int)tv, corr_ref);
            TDV_CHECK_LAUNCH(ctx);
        }
        std::vector<tdv_ransac_result> coarse((size_t)n_instances);
        int fell_back = 0;
        TDV_TRY(ransac_small_batch_dev(ctx, want_ref ? vox_ref_all : vox_first_all, voff.data(), d_voff, n_instances, d_model_xyz, n_model, corr_ref, prm->voxel_size,
                                       prm->ransac_max_iterations, prm->ransac_confidence, prm->seed, coarse.data(), &fell_back));
        if (!fell_back) {
            ransac_done = true;
            for (int b = 0; b < n_instances; ++b) {
                std::memcpy(&coarse_T[(size_t)b * 16], coarse[b].T, 64);
                results[b].coarse_fitness = coarse[b].fitness; results[b].coarse_inliers = coarse[b].inliers;
            }
        }
    }
    if (!ransac_done) TDV_TRY(for_all_instances(stage_register));
    if (batched_icp) {
        std::vector<tdv_icp_result> fine((size_t)n_instances);
        TDV_TRY(icp_small_batch_dev(ctx, want_ref ? vox_ref_all : vox_first_all, d_voff, n_instances, d_model_xyz, d_model_normals, n_model, coarse_T.data(), icp_thr,
                                    prm->icp_max_iterations, prm->point_to_plane, fine.data(), v_max));
        for (int b = 0; b < n_instances; ++b) {
            if (off[b + 1] == off[b]) continue;
            tdv_instance_result& r = results[b];
            std::memcpy(r.T, fine[b].T, 64);
            r.fitness = fine[b].fitness; r.rmse = fine[b].rmse; r.icp_iterations = fine[b].iterations;
            r.status = 0;
        }
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

int tdv_depth_to_cloud_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_masks, const uint8_t* d_bgr,
                                 int n_instances, int mask_format, int n_frames, const int* h_frame_of_instance,
                                 int width, int height, float scale, int mask_mode,
                                 float fx, float fy, float cx, float cy, float zmax,
                                 float* d_xyz, float* d_rgb, long long capacity, int* h_offsets) {
    if (!ctx || !d_raw || !d_masks || !h_offsets || n_instances < 0 || width < 0 || height < 0 || capacity < 0) return TDV_ERR_BAD_ARG;
    if (mask_format < 0 || mask_format > 2) return TDV_ERR_BAD_ARG;
    if (mask_format != 0 && n_frames > 1) return TDV_ERR_BAD_ARG;   // a label image belongs to one frame
    if ((mask_format == 1 && n_instances > 255) || (mask_format == 2 && n_instances > 65535)) return TDV_ERR_BAD_ARG;   // instance b is label b + 1
    const int layout = mask_format == 0 ? 1 : (mask_format == 1 ? 0 : 2);
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    TDV_TRY(ws_reset(ctx));
    h_offsets[0] = 0;
    if (n_instances == 0 || (size_t)width * height == 0) { for (int b = 0; b <= n_instances; ++b) h_offsets[b] = 0; return TDV_OK; }
    int* d_off = nullptr;
    const int* d_frame_of = nullptr;
    TDV_TRY(frame_map_dev(ctx, n_instances, n_frames, h_frame_of_instance, &d_frame_of));
    TDV_TRY(depth_to_cloud_batch_count(ctx, d_raw, d_frame_of, d_masks, n_instances, layout, width, height, scale, mask_mode, zmax, &d_off, h_offsets));
    if ((long long)h_offsets[n_instances] > capacity || (h_offsets[n_instances] > 0 && !d_xyz)) return TDV_ERR_BAD_ARG;
    if (h_offsets[n_instances] > 0)
        TDV_TRY(depth_to_cloud_batch_emit(ctx, d_raw, d_frame_of, d_masks, d_bgr, n_instances, layout, width, height, scale, mask_mode,
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
