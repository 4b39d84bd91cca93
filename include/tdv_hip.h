/*
 * tdv_hip.h — C ABI of the MI355X (gfx950) point-cloud registration backend.
 *
 * This is the drop-in boundary: the thin extern "C" dispatch layer that replaces the
 * reference's CUDA dispatch TU src/gpu_impl.cpp and its launcher set in cuda/ (the .cuh headers)
 * (launchDepthPreprocess, launchDeproject, launchFindCorrespondences,
 * launchBuildLinearSystem).  The C++ adapter in 3dvision_amd/host/ defines the reference's
 * operator API (include/gpu_depth.hpp:9-22, include/gpu_registration.hpp:8-19,
 * include/registration.hpp:32-60) on top of these entry points; INTEGRATION.md shows the
 * binding a maintainer adds.
 *
 * Conventions
 *  - Every function returns a tdv_status (0 = ok, negative = error); nothing throws.
 *  - A tdv_ctx owns one HIP stream, a grow-only device workspace and pinned staging; it is
 *    NOT thread-safe: use one ctx per host thread (the reference calls its GPU ops from up to
 *    8 pool threads, include/thread_pool.hpp:17-33 / src/pipeline.cpp:321-327).
 *  - Host entry points take caller-owned host buffers; "_dev" entry points take device
 *    pointers valid on the ctx's device and enqueue on the ctx's stream.
 *  - Clouds are AoS float[n*3] (bit-identical to std::vector<Eigen::Vector3f>::data()),
 *    FPFH is float[n*33] (std::vector<std::array<float,33>>::data()),
 *    4x4 transforms are COLUMN-MAJOR float[16] (Eigen::Matrix4f::data()).
 *  - Results follow the reference's CPU path src/registration.cpp (the parity oracle), not its
 *    CUDA kernels, where the two differ.
 */
#ifndef TDV_HIP_H
#define TDV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tdv_ctx tdv_ctx;

typedef enum tdv_status {
    TDV_OK = 0,
    TDV_ERR_NO_DEVICE = -1,   /* no HIP device / HIP runtime failure at init                   */
    TDV_ERR_BAD_ARG = -2,     /* null pointer, negative size, capacity too small               */
    TDV_ERR_OOM = -3,         /* device or pinned allocation failed                            */
    TDV_ERR_LAUNCH = -4,      /* kernel launch / stream / copy failure (hipGetLastError)       */
    TDV_ERR_INTERNAL = -5
} tdv_status;

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* Replaces GPUDepth::isCudaAvailable / GPURegistration::isCudaAvailable (src/gpu_impl.cpp:18-26,
 * 131-139): *count = number of HIP devices (0 when none; still TDV_OK). */
int tdv_device_count(int* count);
int tdv_ctx_create(int device, tdv_ctx** out);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the ctx's own. */
int tdv_ctx_set_stream(tdv_ctx* ctx, void* hip_stream);
void* tdv_ctx_get_stream(tdv_ctx* ctx);
int tdv_ctx_synchronize(tdv_ctx* ctx);
void tdv_ctx_destroy(tdv_ctx* ctx);
/* ICP correspondence search.  The reference's kernel is a brute-force scan (cuda/icp.cu:14-55); the pruned search
 * (bounding-box walk over the Morton-ordered target) and the grid search (hash grid with cells of 2.2 x the acceptance
 * threshold: a neighbour within the threshold lies in 8 cells around the query) return the SAME correspondences bit
 * for bit (exact lower bounds resp. every candidate verified with the scan's distance expression, lowest index on ties)
 * and only differ in time.  AUTO (default; env TDV_ICP_SEARCH=brute|pruned|grid overrides at ctx creation) picks by
 * size and by how many target points a cell holds; GRID falls back to PRUNED when the grid is not usable (threshold
 * large against the point spacing, or coordinates beyond 2^17 cells). */
#define TDV_ICP_SEARCH_AUTO 0
#define TDV_ICP_SEARCH_BRUTE 1
#define TDV_ICP_SEARCH_PRUNED 2
#define TDV_ICP_SEARCH_GRID 3
int tdv_ctx_set_icp_search(tdv_ctx* ctx, int mode);
/* ICP accumulation of the per-iteration sums (n_corr, total_error, ATA, ATb resp. the means and the cross-covariance of the
 * point-to-point branch; src/registration.cpp:340-358,374-386).  The reference adds one correspondence after the other in
 * float, in ascending source index; a float sum depends on its order.
 *   TREE (default)  double accumulators in a fixed tree over the whole chip: ~25 us per iteration at 200k points, the same
 *                   bits run to run, and within ~1e-7 (relative) of the reference's sums - the refined transform agrees with
 *                   the CPU path to the tolerances of DESIGN.md 2, not to the bit.
 *   REFERENCE       the reference's own order and precision: every accepted correspondence's terms are stored as rows and one
 *                   workgroup adds them in index order, one lane per accumulator.  With it transformation, fitness, rmse and
 *                   the iteration count EQUAL the CPU path's (tests/test_gpu_icp_reference_order.py); the cost is a chain of
 *                   n dependent additions per iteration (~0.5 ms at 200k points).  Env TDV_ICP_ACCUMULATE=reference selects
 *                   it at ctx creation; tdv_register_batch_dev honours it for every instance. */
#define TDV_ICP_ACCUMULATE_TREE 0
#define TDV_ICP_ACCUMULATE_REFERENCE 1
int tdv_ctx_set_icp_accumulation(tdv_ctx* ctx, int mode);
/* RANSAC hypothesis scoring.  FAST (default; env TDV_RANSAC_SCORE=exact overrides) evaluates every (hypothesis, point) with
 * fused multiply-adds and re-scores, with the reference's unfused arithmetic, every chunk of points in which a distance
 * falls inside the rounding band where the two could disagree: the inlier counts are those of EXACT, which runs the
 * reference's arithmetic only (registration.cpp:270-279). */
#define TDV_RANSAC_SCORE_FAST 0
#define TDV_RANSAC_SCORE_EXACT 1
#define TDV_RANSAC_SCORE_MATRIX 2 /* A/B variant kept in the STUDY library only (lib3dvision_hip_study.so): the transform on the
                                   * matrix cores (f32 MFMA), same band scheme, same counts; measured slower than FAST.  The product
                                   * library returns TDV_ERR_BAD_ARG for it */
int tdv_ctx_set_ransac_score(tdv_ctx* ctx, int mode);
/* Statistics of the last tdv_ransac* call on this ctx: the fraction of the (hypothesis, point) tests the FAST pass scored a second time
 * with the reference arithmetic - whole waves of 64 hypotheses on a pair of points that one of them has inside its rounding band
 * (until round 4: on the whole 8-point chunk) - (-1 if the call ran in EXACT mode or none has run). */
double tdv_ctx_last_ransac_rescore(tdv_ctx* ctx);
/* The share of the (hypothesis, point) tests the last tdv_ransac* call on this ctx evaluated.  Below 1 when the call ran without
 * a per-iteration trace: a hypothesis whose count over a prefix of the points plus ALL remaining points cannot exceed the best
 * count of the earlier batches is not scored further — the loop of registration.cpp:284-290 only asks whether a count beats the
 * best so far, so transform, inlier count, fitness, rmse and best iteration are unchanged (env TDV_RANSAC_BAILOUT=0 turns it off). */
double tdv_ctx_last_ransac_scored(tdv_ctx* ctx);
/* The search the last tdv_icp* / tdv_icp_correspondences call on this ctx ran (BRUTE, PRUNED or GRID; 0 before any). */
int tdv_ctx_last_icp_search(tdv_ctx* ctx);
/* The search the last tdv_feature_match* call on this ctx ran (0 before any): the reference's scan (small sets), the leaf-major
 * search over the packed index, or the per-source walk of that index the leaf-major search hands over to when the descriptors
 * have no structure (its pair pool would run over).  Same correspondences every way. */
#define TDV_FM_PATH_SCAN 1
#define TDV_FM_PATH_LEAF_MAJOR 2
#define TDV_FM_PATH_WALK 3
int tdv_ctx_last_feature_match_path(tdv_ctx* ctx);
/* How the last tdv_register_batch_dev / tdv_voxel_downsample_batch_dev call on this ctx grouped its points into voxels (0 before any):
 * TABLE = a hash table over all clouds (any cloud), PIXELS = pixel windows in LDS, without the table - for clouds the call itself unprojected
 * from a depth image (row-major pixel order, known intrinsics); it hands over to TABLE when a tile is not covered (coarse voxels, rows longer
 * than its halo).  Same voxels, means and orders either way; TDV_VOXEL_PIXELS=0 forces TABLE (parity tests). */
#define TDV_VOXEL_GROUPING_TABLE 1
#define TDV_VOXEL_GROUPING_PIXELS 2
int tdv_ctx_last_voxel_grouping(tdv_ctx* ctx);
/* Host lanes (the caller's thread + helper threads) the last tdv_register_batch_dev call on this ctx used (0 before any). */
int tdv_ctx_last_batch_lanes(tdv_ctx* ctx);
/* Device memory this ctx holds in its grow-only workspace arenas, its batch lanes' included: the high-water mark of every
 * call made on it so far (the arena never shrinks; steady state allocates nothing). */
unsigned long long tdv_ctx_workspace_bytes(tdv_ctx* ctx);
const char* tdv_status_string(int status);
/* Text of the last HIP error seen by this ctx ("" if none). */
const char* tdv_last_error(tdv_ctx* ctx);
const char* tdv_version(void);

/* Kernel timing (HIP events on the ctx's stream around the dominant kernels).  Slots:
 * 0 = ICP nearest-neighbour scan, 1 = RANSAC scoring, 2 = feature match, 3 = kNN scan,
 * 4 = radius scan, 5 = depth+unproject, 6 = voxel, 7 = descriptor index.  Enabling adds one event pair per launch. */
#define TDV_TIMER_ICP_NN 0
#define TDV_TIMER_RANSAC_SCORE 1
#define TDV_TIMER_FEATURE_MATCH 2
#define TDV_TIMER_KNN 3
#define TDV_TIMER_RADIUS 4
#define TDV_TIMER_DEPTH 5
#define TDV_TIMER_VOXEL 6
#define TDV_TIMER_FM_INDEX 7   /* packing of the target descriptors (once per model in the batched chain) */
#define TDV_TIMER_COUNT 8
int tdv_timing_enable(tdv_ctx* ctx, int on);
/* Synchronizes the stream, then returns accumulated milliseconds and launch count; resets the slot. */
int tdv_timing_read(tdv_ctx* ctx, int slot, double* total_ms, int* launches);

/* ---- R1: depth scale + mask --------------------------------------------------------------- */
/* Replaces GPUDepth::preprocess (src/gpu_impl.cpp:28-66, kernel cuda/depth_processing.cu:10-30);
 * oracle = the CPU branch src/pipeline.cpp:46-54.
 * out[i] = float(raw[i]) * float(1.0 / scale); zeroed where the mask rejects the pixel.
 * mask may be NULL (no masking).  mask_mode: 0 = reference CPU semantics (keep mask > 10),
 * 1 = reference CUDA semantics (keep mask != 0). */
#define TDV_MASK_THRESHOLD10 0
#define TDV_MASK_NONZERO 1
/* label image: keep pixels whose mask value equals `label` (1..255): pass
 * TDV_MASK_LABEL_BASE + label.  One u8 image then serves every instance of a scene instead of one
 * full-frame mask per instance (src/pipeline.cpp:251-257, src/segmentation.cpp:12-42 produce those). */
#define TDV_MASK_LABEL_BASE 256
int tdv_depth_preprocess(tdv_ctx* ctx, const uint16_t* raw, const uint8_t* mask, int width, int height,
                         float scale, int mask_mode, float* out_depth);

/* Bilateral depth filter: the reference's kernel cuda/depth_processing.cu:62-122 /
 * launcher :124-155, which its dispatch never calls (config flag depth.bilateral_filter is parsed at
 * src/main.cpp:24 and never read).  in/out: float depth images; zero depths stay zero. */
int tdv_bilateral_filter(tdv_ctx* ctx, const float* depth, int width, int height, float sigma_spatial, float sigma_range,
                         float* out_depth);

/* ---- R2: unprojection --------------------------------------------------------------------- */
/* Replaces GPUPointCloud::generate (src/gpu_impl.cpp:69-128, kernel cuda/pointcloud.cu:11-51);
 * oracle = src/pipeline.cpp:61-84.  Keeps 0 < z <= zmax, x = (u-cx)*z/fx, y = (v-cy)*z/fy,
 * colour = BGR->RGB / 255.  Output order is the CPU's ROW-MAJOR scan order (deterministic; the
 * reference CUDA kernel's global-atomic order is not reproduced).  bgr / out_rgb may be NULL.
 * capacity = room in out_xyz/out_rgb in points; *n_out = points produced (TDV_ERR_BAD_ARG and
 * *n_out set to the needed count if capacity is too small). */
int tdv_deproject(tdv_ctx* ctx, const float* depth, const uint8_t* bgr, int width, int height,
                  float fx, float fy, float cx, float cy, float zmax,
                  float* out_xyz, float* out_rgb, int capacity, int* n_out);
/* Fused R1+R2 (one pass over the frame, no intermediate depth image on the host). */
int tdv_depth_to_cloud(tdv_ctx* ctx, const uint16_t* raw, const uint8_t* mask, const uint8_t* bgr,
                       int width, int height, float scale, int mask_mode,
                       float fx, float fy, float cx, float cy, float zmax,
                       float* out_xyz, float* out_rgb, int capacity, int* n_out);

/* All instances of a scene in two launches: n_instances masks — stacked u8 images
 * (mask_format 0, mask_mode as above), ONE u8 label image with label b+1 for instance b (mask_format 1, <= 255 instances) or ONE u16 label image, same rule (mask_format 2, <= 65535 instances; label images: one frame
 * only) — give n_instances clouds stored back to back, each in row-major pixel order.
 * Frames: d_raw (and d_bgr) hold n_frames images back to back (n_frames <= 1: one frame shared by every instance, the
 * reference's case, src/pipeline.cpp:321-327); instance b reads frame h_frame_of_instance[b] (host array), or, when that
 * is NULL, frame b * n_frames / n_instances (equal contiguous groups; n_frames == n_instances: one frame each).
 * All pointers are device pointers except h_frame_of_instance and h_offsets (host, n_instances + 1 entries): instance b
 * occupies points [h_offsets[b], h_offsets[b+1]).  capacity = room in d_xyz/d_rgb in points; if the total exceeds it
 * the call returns TDV_ERR_BAD_ARG with h_offsets filled (so the caller can size the buffers and call again). */
int tdv_depth_to_cloud_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_masks, const uint8_t* d_bgr,
                                 int n_instances, int mask_format, int n_frames, const int* h_frame_of_instance,
                                 int width, int height, float scale, int mask_mode,
                                 float fx, float fy, float cx, float cy, float zmax,
                                 float* d_xyz, float* d_rgb, long long capacity, int* h_offsets);

/* ---- R3: voxel downsample ----------------------------------------------------------------- */
/* Replaces Registration::voxelDownsample (src/registration.cpp:29-60).  Per-voxel mean of points
 * (and colours) summed in ascending input index, divided by the count; normals are dropped.
 * order: TDV_VOXEL_ORDER_FIRST = voxels ordered by their smallest input index (deterministic,
 * computed on the GPU); TDV_VOXEL_ORDER_REFERENCE = libstdc++ std::unordered_map iteration order
 * of the reference (the slot order is replayed on the host with the reference's hash,
 * registration.cpp:20-27; the means are still computed on the GPU). */
#define TDV_VOXEL_ORDER_FIRST 0
#define TDV_VOXEL_ORDER_REFERENCE 1
int tdv_voxel_downsample(tdv_ctx* ctx, const float* xyz, const float* rgb, int n, float voxel_size, int order,
                         float* out_xyz, float* out_rgb, int capacity, int* n_out);

/* ---- R4a: normals ------------------------------------------------------------------------- */
/* Replaces Registration::estimateNormals (src/registration.cpp:105-130): exact brute-force kNN
 * ((d2, idx) lexicographic order, self included), PCA normal, flipped towards the origin.
 * out_knn (optional, int[n*k], -1 padded) receives the neighbour lists. */
int tdv_estimate_normals(tdv_ctx* ctx, const float* xyz, int n, int k, float* out_normals, int* out_knn);

/* ---- R4b: FPFH ---------------------------------------------------------------------------- */
/* Replaces Registration::computeFPFH (src/registration.cpp:133-201): radius search d2 <= r^2,
 * (d2, idx) order, capped at 100 neighbours; SPFH + weighted FPFH, 33 bins. */
int tdv_compute_fpfh(tdv_ctx* ctx, const float* xyz, const float* normals, int n, float radius,
                     float* out_desc33, int* out_nbr /* optional int[n*100] */, int* out_nbr_cnt /* optional int[n] */);

/* ---- R5: RANSAC --------------------------------------------------------------------------- */
/* Feature correspondences of src/registration.cpp:216-232: argmin_j sum_d (fs[i][d]-ft[j][d])^2,
 * accumulated in d order, strict <, lowest j on ties. */
int tdv_feature_match(tdv_ctx* ctx, const float* fs, int ns, const float* ft, int nt, int* out_corr);

typedef struct tdv_ransac_result {
    float T[16];        /* column-major; identity if no hypothesis ever had an inlier            */
    float fitness;      /* inliers / ns of the winning hypothesis (0 if none)                    */
    float rmse;
    int inliers;        /* inlier count of the winning hypothesis                                 */
    int best_iteration; /* iteration index that produced it (-1 if none)                          */
    int iterations_run; /* iterations consumed, including skipped (degenerate-triple) ones        */
} tdv_ransac_result;

/* Replaces Registration::ransacRegistration (src/registration.cpp:204-295).  corr may be NULL
 * (then fs/ft are matched first) or a precomputed int[ns] (then fs/ft may be NULL).
 * Index triples come from mt19937(seed) + libstdc++-11 uniform_int_distribution<size_t>
 * (Lemire), iteration semantics (skip on repeated index, strict-> best, early exit on
 * fitness > confidence) as the reference.  seed = 42 reproduces registration.cpp:235.
 * trace_inliers (optional int[max_iterations]): per-iteration inlier count, -1 = skipped. */
int tdv_ransac(tdv_ctx* ctx, const float* src, int ns, const float* tgt, int nt,
               const float* fs, const float* ft, const int* corr,
               float voxel_size, int max_iterations, float confidence, uint32_t seed,
               tdv_ransac_result* out, int* trace_inliers);

/* ---- R6: ICP ------------------------------------------------------------------------------ */
typedef struct tdv_icp_result {
    float T[16];     /* column-major */
    float fitness;
    float rmse;
    int iterations;  /* iterations whose update was applied */
    int n_corr;      /* accepted correspondences of the last applied iteration */
} tdv_icp_result;

/* Replaces GPURegistration::icpRefine (src/gpu_impl.cpp:141-260, kernels cuda/icp.cu:14-55,90-142)
 * and Registration::icpRefine (src/registration.cpp:297-414, the oracle).  tgt_normals may be NULL;
 * point-to-plane is used iff point_to_plane != 0 and tgt_normals != NULL, else point-to-point
 * Kabsch (registration.cpp:343,365).  The whole loop runs on the device; the host polls a
 * convergence flag every few iterations. */
int tdv_icp(tdv_ctx* ctx, const float* src, int ns, const float* tgt, const float* tgt_normals, int nt,
            const float* T0, float distance_threshold, int max_iterations, int point_to_plane,
            tdv_icp_result* out);

/* One correspondence pass for a given T (registration.cpp:325-359): nearest target index per
 * source (always written), its squared distance, and the accepted flag; n_corr = accepted count.
 * Exposed for parity tests.  Full scan by default; with TDV_ICP_SEARCH_PRUNED set on the ctx the pruned
 * search is used and rows beyond the threshold report corr 0 / d2 FLT_MAX (accepted rows are identical). */
int tdv_icp_correspondences(tdv_ctx* ctx, const float* src, int ns, const float* tgt, int nt,
                            const float* T, float distance_threshold,
                            int* out_corr, float* out_d2, uint8_t* out_accepted, int* out_n_corr);

/* ---- device-resident entry points (inputs already in HBM; used by bench.py and by batched callers)
 * All pointers are device pointers on the ctx's device; work is enqueued on the ctx's stream and
 * the call returns after the stream has been synchronized (results are host structs). ------------ */
/* fixed_iterations != 0: run exactly max_iterations iterations (convergence and the n_corr<3 break
 * are still evaluated on the device and reported, but do not stop the loop) — benchmarking only. */
int tdv_icp_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, const float* d_tgt_normals, int nt,
                const float* T0, float distance_threshold, int max_iterations, int point_to_plane,
                int fixed_iterations, tdv_icp_result* out);
int tdv_ransac_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt,
                   const float* d_fs, const float* d_ft, const int* d_corr,
                   float voxel_size, int max_iterations, float confidence, uint32_t seed,
                   tdv_ransac_result* out, int* trace_inliers /* host, optional */);
int tdv_feature_match_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr);
int tdv_estimate_normals_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float* d_normals, int* d_knn);
int tdv_compute_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, float radius,
                         float* d_desc33, int* d_nbr, int* d_nbr_cnt);
/* estimateNormals(k) followed by computeFPFH(radius) on the same cloud (src/pipeline.cpp:93-95) with ONE neighbour walk: the radius
 * lists are sorted by (d2, idx) and capped at the 100 smallest, so wherever a point has >= k neighbours in radius its k nearest
 * neighbours ARE the head of its radius list (registration.cpp:68-74,95-99); only the deficient points (isolated points, silhouette
 * edges) go through a kNN search of their own, as a subset.  Normals and descriptors are those of the two separate calls bit for bit
 * (tests/test_gpu_features.py); what tdv_register_batch_dev and tdv_prepare_model_dev run.  k <= 100 shares the walk; larger k falls
 * back to the two calls. */
int tdv_normals_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float radius, float* d_normals, float* d_desc33);
/* The stable radix sort the descriptor index build uses (csrc/sort.hip, hand-written: a utility without a counterpart in the reference,
 * exported so that it is tested on its own): n (64-bit key, 32-bit value) pairs in device memory ordered by the low end_bit bits of the
 * key, pairs of equal keys in input order.  The in and out buffers must not overlap. */
int tdv_radix_sort_pairs_dev(tdv_ctx* ctx, const unsigned long long* d_keys_in, unsigned long long* d_keys_out, const unsigned* d_vals_in,
                             unsigned* d_vals_out, size_t n, int end_bit);
int tdv_depth_to_cloud_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_mask, const uint8_t* d_bgr,
                           int width, int height, float scale, int mask_mode,
                           float fx, float fy, float cx, float cy, float zmax,
                           float* d_xyz, float* d_rgb, int capacity, int* n_out /* host */);
/* order: TDV_VOXEL_ORDER_FIRST or TDV_VOXEL_ORDER_REFERENCE; for the latter the host receives 16 B per voxel (cell and
 * input index of its first point) to replay the reference's container — the cloud itself stays on the device. */
int tdv_voxel_downsample_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int n, float voxel_size, int order,
                             float* d_out_xyz, float* d_out_rgb, int capacity, int* n_out /* host */);

/* Registration::voxelDownsample (src/registration.cpp:29-60) for n_clouds clouds stored back to back at d_xyz (cloud b =
 * points [h_cloud_offsets[b], h_cloud_offsets[b+1]); host array of n_clouds + 1 entries starting at 0), in ONE set of launches:
 * cloud b's voxels, in first-occurrence order (TDV_VOXEL_ORDER_FIRST), come back at [h_voxel_offsets[b], h_voxel_offsets[b+1])
 * of d_out_xyz, which has room for as many points as d_xyz holds.  Means are the reference's (f32 sums in ascending input
 * index).  What tdv_register_batch_dev runs for its instances. */
int tdv_voxel_downsample_batch_dev(tdv_ctx* ctx, const float* d_xyz, const int* h_cloud_offsets, int n_clouds, float voxel_size,
                                   float* d_out_xyz, int* h_voxel_offsets);
/* The same for clouds that were unprojected from depth images with the given pinhole intrinsics and are still in the row-major pixel
 * order tdv_depth_to_cloud*_dev emits (src/pipeline.cpp:68-83): the members of a voxel then lie within a few pixels of each other, and the
 * points are grouped through pixel windows in LDS instead of a hash table (no device-scope atomics; csrc/voxel.hip: k_vs_group) - what
 * tdv_register_batch_dev does for its own clouds.  Same voxels, means and order, bit for bit; a cloud or voxel size the window argument
 * does not cover (coarse voxels, rows longer than the halo, points not in pixel order) is redone through the table inside the call
 * (tdv_ctx_last_voxel_grouping tells). */
int tdv_voxel_downsample_batch_pinhole_dev(tdv_ctx* ctx, const float* d_xyz, const int* h_cloud_offsets, int n_clouds, float voxel_size,
                                           float fx, float fy, float cx, float cy, float* d_out_xyz, int* h_voxel_offsets);

/* ---- batched, device-resident Pipeline::processInstance ------------------------
 * One call runs the whole per-instance chain of src/pipeline.cpp:25-150 for n_instances masks that
 * share one depth/colour frame and one prepared reference model, without returning to the host
 * between stages (the reference's operator API crosses PCIe at every op boundary):
 *   mask -> depth scale+mask -> unproject -> voxelDownsample -> estimateNormals(k) ->
 *   computeFPFH(voxel * fpfh_radius_factor) -> feature match + RANSAC -> ICP(threshold = voxel *
 *   icp_distance_factor) -> refined transform.
 * All pointers are device pointers; d_masks holds n_instances full-frame uint8 masks back to back;
 * the model (points, normals, FPFH) is what Pipeline::run prepares once (src/pipeline.cpp:291-294);
 * results is a HOST array of n_instances entries.  An instance whose masked depth image holds no non-zero value gets
 * status 1 (the reference returns nullopt at src/pipeline.cpp:57-60, "empty depth after masking"); one that has depth but
 * no pixel inside 0 < z <= zmax gets status 2 (:86-89, "empty point cloud").
 * voxel_order: TDV_VOXEL_ORDER_REFERENCE gives, per instance, exactly what the chain of host-buffer operators (and
 * the reference's processInstance) gives — RANSAC's mt19937 index stream picks points by position, so the pose depends
 * on the order of the downsampled cloud; TDV_VOXEL_ORDER_FIRST skips the computation of the reference's container order
 * (done on the device, ~17 small passes per batch) and yields a different, equally valid, coarse pose.
 * The RANSAC index stream is seeded per instance exactly as the reference does (mt19937(42) restarted for every
 * ransacRegistration call).  Frames: as tdv_depth_to_cloud_batch_dev (n_frames, frame_of_instance).
 * The call spreads the instances over several lanes (the caller's thread plus helper threads, each with its own stream
 * and workspace owned by the ctx): 6 for large instances, 12 for small ones (under 8,192 points on average), never more than the
 * host has hardware threads; TDV_BATCH_LANES=n overrides (at most 16).  Stages that do not depend on an instance run once for the
 * whole batch: the clouds (2 launches), the voxels and their reference order (on the device: no leader leaves it) and, for
 * small instances, the descriptor match of all instances' points against the model. */
typedef struct tdv_batch_params {
    int width, height;
    float scale_to_meters;      /* depth.scale_to_meters      (include/pipeline_config.hpp:18) */
    int mask_mode;              /* TDV_MASK_THRESHOLD10 / TDV_MASK_NONZERO                        */
    float fx, fy, cx, cy, zmax; /* intrinsics; zmax = depth.clipping_max                          */
    float voxel_size;           /* registration.voxel_size                                        */
    int normals_k;              /* 30                        (src/pipeline.cpp:93)                */
    float fpfh_radius_factor;   /* 5.0                       (src/pipeline.cpp:95)                */
    int ransac_max_iterations;  /* registration.ransac_max_iterations                             */
    float ransac_confidence;    /* 0.999                                                          */
    float icp_distance_factor;  /* 0.4                       (src/pipeline.cpp:104)               */
    int icp_max_iterations;     /* registration.icp_max_iterations                                */
    int point_to_plane;         /* registration.use_point_to_plane                                */
    uint32_t seed;              /* 42                        (src/registration.cpp:235)           */
    int voxel_order;            /* TDV_VOXEL_ORDER_FIRST / TDV_VOXEL_ORDER_REFERENCE                */
    int n_frames;               /* depth frames stored back to back at d_raw_depth (0 or 1: one)    */
    const int* frame_of_instance; /* HOST array [n_instances] or NULL (b * n_frames / n_instances)   */
    int mask_format;            /* 0: n_instances stacked u8 masks (mask_mode applies); 1: ONE u8 label image, instance b keeps
                                   the pixels equal to b + 1 (<= 255 instances); 2: ONE u16 label image, same rule (<= 65535
                                   instances).  Label images need n_frames <= 1.  SURVEY.md 8f N2                          */
    int mask_width, mask_height; /* size of the masks when it differs from the frame (0: the frame's): they are resized with
                                   nearest neighbour first, as cv::resize(..., INTER_NEAREST) in src/pipeline.cpp:38-41   */
} tdv_batch_params;

typedef struct tdv_instance_result {
    float T[16];            /* refined.transformation, column-major */
    float fitness, rmse;    /* of the ICP result */
    float coarse_fitness;   /* of the RANSAC result */
    int coarse_inliers;
    int icp_iterations;
    int n_points;           /* unprojected points */
    int n_voxels;           /* after voxelDownsample */
    int status;             /* 0 ok, 1 empty depth after masking, 2 empty cloud */
} tdv_instance_result;

/* cv::resize(mask, out, dsize, 0, 0, cv::INTER_NEAREST) (src/pipeline.cpp:38-41) for n_masks u8 images stored back to back:
 * out(y, x) = in(min(floor(y * ify), sh - 1), min(floor(x * ifx), sw - 1)), ifx = 1 / ((double)dw / sw) in double, as OpenCV's
 * resizeNN computes its index tables.  Host buffers; the _dev form takes device pointers. */
int tdv_mask_resize_nearest(tdv_ctx* ctx, const uint8_t* masks, int n_masks, int src_width, int src_height,
                            int dst_width, int dst_height, uint8_t* out);
int tdv_mask_resize_nearest_dev(tdv_ctx* ctx, const uint8_t* d_masks, int n_masks, int src_width, int src_height,
                                int dst_width, int dst_height, uint8_t* d_out);

int tdv_register_batch_dev(tdv_ctx* ctx, const uint16_t* d_raw_depth, const uint8_t* d_bgr /* may be NULL */,
                           const uint8_t* d_masks, int n_instances, const tdv_batch_params* params,
                           const float* d_model_xyz, const float* d_model_normals, const float* d_model_fpfh, int n_model,
                           tdv_instance_result* results);
/* Model preparation of src/pipeline.cpp:291-294 on device buffers: voxelDownsample(voxel_order) ->
 * estimateNormals(k) -> computeFPFH(voxel * radius_factor).  Outputs have capacity n. */
int tdv_prepare_model_dev(tdv_ctx* ctx, const float* d_xyz, int n, float voxel_size, int voxel_order, int normals_k,
                          float fpfh_radius_factor, float* d_out_xyz, float* d_out_normals, float* d_out_fpfh, int* n_out /* host */);

/* ---- multi-GPU: instances shard across one process per GPU; the reference's only parallel axis is the
 * same one, over host threads (src/pipeline.cpp:321-327).  rccl_comm is the caller's ncclComm_t (RCCL; one rank per
 * process, created by the host with ncclCommInitRank) — this library does not link RCCL, it resolves ncclBroadcast /
 * ncclAllGather among the process's loaded symbols, else from librccl.so.1, at the first call.  Both calls enqueue on the
 * ctx's stream and return after it has been synchronized; every rank must make the same calls in the same order.
 *
 * tdv_broadcast_model: the prepared model (what tdv_prepare_model_dev / Pipeline::run :291-294 produce) from rank `root`
 * to every rank, in place: on root *n_model is the input count, elsewhere it receives it; buffers hold `capacity` points
 * on every rank (TDV_ERR_BAD_ARG if the model does not fit).  d_normals may be NULL (no normals: ICP falls back to
 * point-to-point, registration.cpp:343); normals travel only if EVERY rank passed a buffer.
 * Rank-local arguments (buffers, capacity, counts) are validated through one all-gather of a 16-byte header before any
 * payload moves, so every rank takes the same branch and returns the SAME status — a bad argument on one rank makes all
 * ranks return TDV_ERR_BAD_ARG instead of leaving the others blocked in a collective.  `root` (and slots_per_rank below)
 * must agree across ranks like the arguments of any collective; a disagreement in slots_per_rank is detected and refused.
 * STATUS: verified on hardware at world size 1 only (tests/test_gpu_comm.py); world > 1 needs more GPUs than a test box has —
 * "parity unpinned" for world > 1 until the driver's multi-GPU run.
 * tdv_gather_results: every rank contributes n_local results in slots_per_rank slots (the same number on every rank,
 * >= n_local; unused slots come back with status -1) and receives all ranks' slots, rank-major, in `all`
 * (world_size * slots_per_rank entries).  Both are host arrays. */
int tdv_broadcast_model(tdv_ctx* ctx, void* rccl_comm, int root, float* d_xyz, float* d_normals, float* d_fpfh, int capacity, int* n_model);
int tdv_gather_results(tdv_ctx* ctx, void* rccl_comm, const tdv_instance_result* local, int n_local, int slots_per_rank,
                       tdv_instance_result* all);

/* ---- host-side helpers that are part of the path's semantics -------------------------------- */
/* The RANSAC index stream: count triples from mt19937(seed) + Lemire uniform over [0, n-1]
 * (src/registration.cpp:235-239 on libstdc++ 11).  Own implementation, no <random>. */
int tdv_sample_triples(uint32_t seed, uint64_t n, int count, uint64_t* out_triples);
/* Pose composition of src/pipeline.cpp:136-137: out = extrinsics * inverse(T). */
int tdv_pose_compose(const float* extrinsics, const float* T, float* out);
/* Pipeline::filterDuplicates (src/pipeline.cpp:153-180): greedy pass over n column-major 4x4 poses; a pose
 * within min_distance of a kept one is a duplicate and replaces it only if it is closer to the origin.
 * out_poses has room for n poses; *n_out = kept count. */
int tdv_filter_duplicates(const float* poses, int n, float min_distance, float* out_poses, int* n_out);
/* Registration::loadReferenceModel (src/registration.cpp:416-461): ASCII PLY, x y z [r g b] per vertex.
 * Keeps the reference's behaviour: colours are detected by "red" appearing in any header line and are
 * divided by 255 when r > 1; the header loop consumes the line AFTER end_header, so the first vertex is
 * skipped and the last read fails — the reference then pushes x = 0 and unspecified y, z (and colour);
 * here that last point is (0,0,0) with colour (0,0,0).  out_xyz / out_rgb (either may be NULL) have room
 * for `capacity` points; *n_out = points the reference would return (= the header's vertex count);
 * *has_color = 1 if colours are present.  Returns TDV_ERR_BAD_ARG if the file cannot be opened
 * (the reference returns an empty cloud there). */
int tdv_load_ply_ascii(const char* path, float* out_xyz, float* out_rgb, int capacity, int* n_out, int* has_color);

/* Segmentation::loadMasksFromDir (src/segmentation.cpp:12-42): every .png/.jpg/.jpeg of `dir` in sorted order, read
 * as 8-bit grey and thresholded (> 10 -> 255, else 0).  Decodable here: non-interlaced greyscale PNGs (colour type 0 or
 * 4, any bit depth); colour / palette PNGs and JPEGs are skipped and counted in *n_skipped (their grey conversion
 * depends on the image library).  tdv_load_mask_png: out may be NULL to query the size; capacity in pixels.
 * tdv_load_masks_from_dir: masks of exactly width x height are stacked into out[n][height][width] — the layout
 * tdv_register_batch_dev and tdv_depth_to_cloud_batch_dev take; *n_out = masks found (may exceed capacity_masks when
 * out is NULL: count query).  A missing directory yields 0 masks, as in the reference. */
int tdv_load_mask_png(const char* path, uint8_t* out, long long capacity, int* width, int* height);
int tdv_load_masks_from_dir(const char* dir, int width, int height, uint8_t* out, int capacity_masks, int* n_out, int* n_skipped);

#ifdef __cplusplus
}
#endif
#endif /* TDV_HIP_H */
