// Internal definitions of the MI355X registration backend (not part of the C ABI).
// One tdv_ctx = one HIP stream + a grow-only device workspace + pinned staging + optional
// per-kernel HIP-event timers.  Steady state performs no hipMalloc.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "tdv_hip.h"

namespace tdv {

struct TimerSlot {
    double total_ms = 0.0;
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace tdv

struct tdv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device workspace: a list of blocks; coalesced into one block at the next reset when it grew
    struct Block { char* p; size_t cap; };
    std::vector<Block> blocks;
    size_t cur_block = 0, cur_off = 0, high_water = 0, used_total = 0;
    // pinned host staging
    char* pin = nullptr;
    size_t pin_cap = 0;
    char err[512] = {0};
    bool timing = false;
    int icp_search = 0;      // TDV_ICP_SEARCH_AUTO / _BRUTE / _PRUNED / _GRID (tdv_ctx_set_icp_search)
    int icp_accumulate = 0;  // TDV_ICP_ACCUMULATE_TREE (f64 fixed tree) / _REFERENCE (f32, ascending source index: the CPU path's sums bit for bit)
    int ransac_score_mode = 0;    // TDV_RANSAC_SCORE_FAST (FMA pass + exact band) / _EXACT (the reference arithmetic only) / _MATRIX (tdv_ctx_set_ransac_score)
    double last_ransac_rescore = -1.0;   // fraction of (wave, 8-point chunk) pairs of the last RANSAC call that the fast pass scored again exactly (-1: exact mode)
    double last_ransac_scored = 1.0;     // share of the (hypothesis, point) tests the last RANSAC call evaluated (< 1: the exact bail-out left the rest out)
    int last_voxel_grouping = 0;   // 1: the table (k_vh_insert), 2: pixel windows (k_vs_group) - the last batched voxel call (tdv_ctx_last_voxel_grouping)
    int last_fm_path = 0;      // TDV_FM_PATH_* of the last descriptor match on this ctx (tdv_ctx_last_feature_match_path)
    int last_batch_lanes = 0;  // host lanes the last tdv_register_batch_dev call on this ctx spread its instances over (tdv_ctx_last_batch_lanes)
    int last_icp_search = 0; // the search the last ICP / correspondence call on this ctx actually ran (tdv_ctx_last_icp_search)
    unsigned* scan_ticket = nullptr;  // persistent device words: [0] exclusive_scan_dev's last-workgroup ticket, [1..] ICP's, [8] the tile ticket of depth.hip's chained scan
    unsigned long long* chain_status = nullptr;   // depth.hip k_depth_cloud_chain: one status word per tile, persistent (told apart by chain_epoch)
    size_t chain_cap = 0;
    unsigned chain_epoch = 0, chain_ticket_base = 0;   // calls so far; tickets handed out so far (the kernel's tile = ticket - base)
    uint16_t* depth_bits = nullptr;   // validity bitmap between the two passes of the batched depth -> cloud (workspace memory of the current call)
    tdv_ctx* helper = nullptr;   // second stream + workspace of the batched pipeline's other lane (owned; created on first use)
    tdv::TimerSlot timers[TDV_TIMER_COUNT];
    std::vector<hipEvent_t> event_pool;
};

namespace tdv {

inline int set_err(tdv_ctx* ctx, hipError_t e, const char* what, int line) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s failed at line %d: %s", what, line, hipGetErrorString(e));
    if (e == hipErrorOutOfMemory) return TDV_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return TDV_ERR_NO_DEVICE;
    return TDV_ERR_LAUNCH;
}

#define TDV_HIP(ctx, call)                                              \
    do {                                                                \
        hipError_t e__ = (call);                                        \
        if (e__ != hipSuccess) return tdv::set_err((ctx), e__, #call, __LINE__); \
    } while (0)

#define TDV_TRY(expr)                    \
    do {                                 \
        int s__ = (expr);                \
        if (s__ != TDV_OK) return s__;   \
    } while (0)

#define TDV_CHECK_LAUNCH(ctx) TDV_HIP((ctx), hipGetLastError())

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Study build (-DTDV_STUDY -> lib3dvision_hip_study.so, used by tools/studies/ and by the tests marked `study`): keeps the A/B
// variants that LOST their measurement (the matrix-core scoring pass, the merged scoring dispatch, round 1's key-ordered descriptor
// scan, one-point-per-wave SPFH / FPFH, the full bitonic voxel sort, ...) and the tuning knobs, all behind study_env().  In the
// product library study_env() is the constant nullptr: every branch behind it is dead code the compiler drops, and the kernels
// only those branches launch are not compiled at all (#ifdef TDV_STUDY).  What stays a real getenv() is what a deployer or a
// parity test of a LIVE path needs.
#ifdef TDV_STUDY
inline const char* study_env(const char* name) { return getenv(name); }
constexpr bool kStudyBuild = true;
#else
constexpr const char* study_env(const char*) { return nullptr; }
constexpr bool kStudyBuild = false;
#endif

#ifdef __HIPCC__
// Sum over the 64 lanes of a wave with DPP row operations (no LDS crossbar traffic, unlike __shfl_down): the result is
// valid in lane 63 and returned wave-uniform.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror: every lane of a 16-lane row holds the row sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
#endif

// Workspace: reset at the start of every public entry point.
int ws_reset(tdv_ctx* ctx);
int ws_alloc_bytes(tdv_ctx* ctx, size_t bytes, void** out);
template <class T>
inline int ws_alloc(tdv_ctx* ctx, size_t count, T** out) {
    void* p = nullptr;
    int s = ws_alloc_bytes(ctx, count * sizeof(T), &p);
    *out = static_cast<T*>(p);
    return s;
}
int pin_reserve(tdv_ctx* ctx, size_t bytes);
// mark / rewind: a batched caller frees everything one instance allocated while keeping what came before
struct WsMark { size_t block, off, used; };
inline WsMark ws_mark(tdv_ctx* ctx) { return WsMark{ctx->cur_block, ctx->cur_off, ctx->used_total}; }
inline void ws_rewind(tdv_ctx* ctx, const WsMark& m) { ctx->cur_block = m.block; ctx->cur_off = m.off; ctx->used_total = m.used; }

// Events for stream synchronisation points come from (and go back to) the ctx's pool: no hipEventCreate in steady state.
hipEvent_t event_acquire(tdv_ctx* ctx);
void event_release(tdv_ctx* ctx, hipEvent_t e);

// Timing helpers: bracket a launch with events when ctx->timing is on.
struct ScopedTimer {
    tdv_ctx* ctx; int slot; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(tdv_ctx* c, int s);
    ~ScopedTimer();
};

// ------------------------------------------------------------------ device pipelines
struct IcpOutputs {  // optional per-source outputs of one correspondence pass (device pointers)
    int* corr = nullptr; float* d2 = nullptr; uint8_t* accepted = nullptr;
};
struct SortedCloud;
// Hash grid over a target cloud for ICP's correspondence search at one acceptance threshold (icp.hip): cells of 2.2 x
// the threshold, open-addressing table of (32-bit cell tag, list head), the points of a cell as a linked list of (x, y, z, next) nodes indexed like the cloud.  Lives
// in the workspace of the ctx that built it; read-only afterwards.  usable = 0: coordinates too large for the cell
// arithmetic or too many points per cell (the threshold is large against the spacing) - the caller takes another search.
struct CellGrid { const void* table; const void* node; unsigned mask; int shift; float inv_cell; float thr; int n; int usable; };
int cell_grid_build(tdv_ctx* ctx, const float* d_tgt, int nt, float thr, CellGrid* g);
// tgt_sorted / tgt_grid (optional): the target in Morton order with its boxes (spatial_sort_cloud) resp. its hash grid
// (cell_grid_build, for this thr), built once and reused across calls
int icp_run_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, const float* d_tgt_normals, int nt,
                const float* T0, float thr, int max_iterations, int point_to_plane, int fixed_iterations,
                tdv_icp_result* out, const SortedCloud* tgt_sorted = nullptr, const CellGrid* tgt_grid = nullptr);
// many small problems against one target in one launch (icp.hip: k_icp_small); sizes up to icp_small_max_points() each
int icp_small_max_points();
long long icp_small_max_pairs_batch();
int icp_small_batch_dev(tdv_ctx* ctx, const float* d_src, const int* d_src_off, int n_prob, const float* d_tgt, const float* d_tgt_normals, int nt,
                        const float* T0s, float thr, int max_iterations, int point_to_plane, tdv_icp_result* out, int ns_max = 0 /* largest problem, if known */);
int icp_correspondences_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt,
                            const float* T, float thr, IcpOutputs outs, int* n_corr);
int ransac_run_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt,
                   const float* d_fs, const float* d_ft, const int* d_corr,
                   float voxel, int max_iterations, float confidence, uint32_t seed,
                   tdv_ransac_result* out, int* trace_inliers);
// ransac_run_dev for many small clouds against one target in a handful of launches (ransac.hip): cloud b = points
// [h_off[b], h_off[b+1]) of d_src with correspondences d_corr (same indexing); out[b].rmse is NOT evaluated (0).  *fell_back = 1:
// nothing was computed (a cloud too large, too many hypotheses, or the index sampler ran out of draws) - use ransac_run_dev.
int ransac_small_batch_dev(tdv_ctx* ctx, const float* d_src, const int* h_off, const int* d_off, int n_clouds, const float* d_tgt, int nt, const int* d_corr,
                           float voxel, int max_iterations, float confidence, uint32_t seed, tdv_ransac_result* out, int* fell_back);
int feature_match_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr);
// Packed index of a target descriptor set (fmatch.hip): rows in sort-tile-recursive order along the set's principal
// directions, padded per column, with 33-D boxes of the 64-row leaves and the 64-leaf groups.  Lives in the workspace
// of the ctx that built it (valid until that ctx's next ws_reset / rewind below the build); read-only afterwards, so a
// batch builds it once for the model and every instance (on either lane) queries it.
struct FmIndex {
    const float* ft = nullptr;                                      // the caller's rows, [nt][33] (must outlive the index: the fallback scan reads them)
    float* T = nullptr; int* torig = nullptr;                       // [leaf][33][64] (row r = column r % 64 of leaf r / 64), [rows] (INT_MAX = padding)
    float* lbox = nullptr;                                          // leaf boxes [group][min | max][33][64 leaves]
    float* gbox = nullptr;                                          // group boxes [chunk of 64 groups][min | max][33][64 groups]
    float *pbox = nullptr, *gpbox = nullptr;                        // the same for the 3 principal coordinates: [..][min | max][3][64]
    float *sleaf = nullptr, *sgroup = nullptr;                      // the same boxes once more, one box = 72 consecutive floats (min[33] | max[33] | pmin[3] | pmax[3]): staged in LDS by the leaf-major search (k_lm_boxes)
    unsigned* amax = nullptr; float pscale = 0.f;                   // largest |x_d - mean_d| of the targets (float bits); 0 disables the 3-D boxes
    float *basis = nullptr, *b0 = nullptr, *b1 = nullptr, *leaf_p2 = nullptr; const int* col_leaf0 = nullptr;   // locating a source's cell
    int nt = 0, rows = 0, nleaf = 0, ngroup = 0, S0 = 1, S1 = 1;
};
int fm_index_build(tdv_ctx* ctx, const float* d_ft, int nt, FmIndex* out);
int feature_match_indexed_dev(tdv_ctx* ctx, const float* d_fs, int ns, const FmIndex& ix, int* d_corr);
int depth_preprocess_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_mask, int w, int h, float scale,
                         int mask_mode, float* d_out);
int bilateral_filter_dev(tdv_ctx* ctx, const float* d_in, float* d_out, int w, int h, float sigma_spatial, float sigma_range);
int depth_to_cloud_dev(tdv_ctx* ctx, const uint16_t* d_raw, const float* d_depth_f32, const uint8_t* d_mask,
                       const uint8_t* d_bgr, int w, int h, float scale, int mask_mode,
                       float fx, float fy, float cx, float cy, float zmax,
                       float* d_xyz, float* d_rgb, int capacity, int* n_out);
int frame_map_dev(tdv_ctx* ctx, int n_inst, int n_frames, const int* h_frame_of, const int** d_frame_of);
int depth_to_cloud_batch_count(tdv_ctx* ctx, const uint16_t* d_raw, const int* d_frame_of, const uint8_t* d_masks, int n_inst, int stacked, int w, int h,
                               float scale, int mask_mode, float zmax, int** d_offsets_out, int* h_offsets);
int depth_to_cloud_batch_emit(tdv_ctx* ctx, const uint16_t* d_raw, const int* d_frame_of, const uint8_t* d_masks, const uint8_t* d_bgr, int n_inst, int stacked,
                              int w, int h, float scale, int mask_mode, float fx, float fy, float cx, float cy, float zmax,
                              const int* d_offsets, float* d_xyz, float* d_rgb);
// `stacked` above: 1 = one u8 mask per instance, 0 = one u8 label image (label = instance + 1), 2 = one u16 label image
int mask_resize_nearest_dev(tdv_ctx* ctx, const uint8_t* d_src, int n_masks, int sw, int sh, int dw, int dh, uint8_t* d_dst);
void resize_nn_tables(int sw, int sh, int dw, int dh, int* x_ofs, int* y_ofs);
int depth_batch_nonzero_any(tdv_ctx* ctx, const uint16_t* d_raw, const int* d_frame_of, const uint8_t* d_masks, int layout, int w, int h, float scale,
                            int mask_mode, const int* h_inst, int n_list, int* h_flags);
int estimate_normals_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float* d_normals, int* d_knn);
int compute_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, float radius,
                     float* d_desc, int* d_nbr, int* d_nbr_cnt);
// With TDV_VOXEL_ORDER_REFERENCE, `both` (optional) also receives the cloud in first-occurrence order and the permutation
// between the two orders: out_xyz[p] == first_xyz[ref2first[p]], first2ref[ref2first[p]] == p (each of capacity entries).
struct VoxelBothOrders { float* first_xyz; int* ref2first; int* first2ref; };
int voxel_downsample_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int n, float voxel, int order,
                         float* d_out_xyz, float* d_out_rgb, int capacity, int* n_out, const VoxelBothOrders* both = nullptr);

// batch: every cloud's voxels in first-occurrence order with one memset + two launches (voxel.hip); the reference order per cloud
// pinhole4 (optional: fx, fy, cx, cy) + h_seg_off (the clouds' offsets on the host): the clouds were unprojected from a depth image with these
// intrinsics, in row-major pixel order - they are then grouped through pixel windows in LDS instead of the table (voxel.hip: k_vs_group)
int voxel_downsample_batch_dev(tdv_ctx* ctx, const float* d_xyz, int total, const int* d_seg_off, int n_clouds, float voxel,
                               float* d_first_xyz, int* d_rank, int4* d_leaders, int* h_voff, int* overflowed, int* d_voff_keep = nullptr,
                               const float* pinhole4 = nullptr, const int* h_seg_off = nullptr);
// the reference's container order of every cloud of a batch, computed on the device (voxel.hip); h_failed[b] != 0: finish cloud b with voxel_reference_order
int voxel_reference_order_batch_dev(tdv_ctx* ctx, int n_clouds, const int* h_voff, const int* d_voff, const int4* d_leaders, const float* d_first_xyz,
                                    float* d_out_xyz, int* d_ref2first, int* d_first2ref, int* h_failed);
int voxel_reference_order(tdv_ctx* ctx, int v, int n, const int4* d_leaders, const float* tmp_xyz, const float* tmp_rgb, const int* d_rank, int rank_base,
                          float* d_out_xyz, float* d_out_rgb, const VoxelBothOrders* both);

int sort_records_dev(tdv_ctx* ctx, uint4* rec, size_t n_pow2);  // voxel.hip: ascending bitonic sort, n_pow2 >= 2048
// sort.hip: stable LSD radix sort (hand-written) of (key, value) pairs on the low end_bit bits of the 64-bit key; any n; scratch from the workspace
int radix_sort_pairs_dev(tdv_ctx* ctx, const unsigned long long* d_keys_in, unsigned long long* d_keys_out,
                         const unsigned* d_vals_in, unsigned* d_vals_out, size_t n, int end_bit);
// every segment [d_seg_start[c], d_seg_start[c + 1]) sorted on its own in one launch; segments of at most segment_sort_max_len() records
int segment_sort_records_dev(tdv_ctx* ctx, uint4* rec, const int* d_seg_start, int nseg);
int segment_sort_max_len();
size_t sort_pow2(size_t n);
int exclusive_scan_dev(tdv_ctx* ctx, const int* d_in, int n, int* d_out, int* d_total);  // voxel.hip
// Morton-ordered copy of a cloud with the bounding boxes of its 64-point leaves and 4096-point groups (workspace memory):
// sx/sy/sz are padded with +inf to `pad` (a multiple of 256), orig[i] = original index of sorted position i.
// Box arrays are [6][count]: min x,y,z then max x,y,z.
struct SortedCloud { float *sx, *sy, *sz; int* orig; float *lbox, *tbox; int n, pad, n_leaf, n_top; };
int spatial_sort_cloud(tdv_ctx* ctx, const float* d_xyz, int n, SortedCloud& out);

// normals + FPFH in one go, sharing one spatial sort and one radius scan (batch path; identical results)
// d_tie_ids / d_tie_ids_inv (optional, both or neither): neighbour lists are ordered by (d2, d_tie_ids[index]) instead of
// (d2, index) — the results are those of the cloud permuted so that point i sits at position d_tie_ids[i]
int normals_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float radius, float* d_normals, float* d_desc,
                     const int* d_tie_ids = nullptr, const int* d_tie_ids_inv = nullptr);                                      // padded record count for sort_records_dev

// the same for many small clouds stored back to back, in one set of launches (knn.hip); tie ids / their inverse are GLOBAL here
int normals_fpfh_batch_dev(tdv_ctx* ctx, const float* d_xyz, const int* h_voff, const int* d_voff, int n_clouds, int k, float radius,
                           float* d_normals, float* d_desc, const int* d_tie_ids = nullptr, const int* d_tie_ids_inv = nullptr);

int register_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_bgr, const uint8_t* d_masks, int n_instances,
                       const tdv_batch_params* prm, const float* d_model_xyz, const float* d_model_normals,
                       const float* d_model_fpfh, int n_model, tdv_instance_result* results);

// host helpers
void mt19937_lemire_triples(uint32_t seed, uint64_t n, int count, uint64_t* out);
void mt19937_raw(uint32_t seed, size_t count, uint32_t* out);
// the same index stream, drawn incrementally (one (i0, i1, i2) triple per call) so that it can be produced batch by
// batch while the GPU scores the previous batch
class TripleStream {
public:
    TripleStream(uint32_t seed, uint64_t n);
    ~TripleStream();
    void next(uint64_t* out3);
private:
    struct Impl; Impl* impl_; uint64_t n_;
    TripleStream(const TripleStream&) = delete;
};
// largest float f with sqrtf(f) <= thr  (accept  <=>  d2 <= f  <=>  !(sqrtf(d2) > thr))
float tau_le(float thr);
// smallest float f with sqrtf(f) >= thr (inlier <=>  d2 < f   <=>  sqrtf(d2) < thr)
float tau_lt(float thr);

}  // namespace tdv
