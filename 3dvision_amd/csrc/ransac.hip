// RANSAC coarse alignment on gfx950.
//
// Replaces Registration::ransacRegistration (/root/reference/src/registration.cpp:204-295), which
// has no GPU entry point in the reference (src/pipeline.cpp:97-102 calls the CPU static directly).
// Sub-steps and their kernels:
//  (i)   feature correspondences (registration.cpp:216-232): 33-D squared distance accumulated in d
//        order without FMA, strict <, lowest j wins; source descriptors live in VGPRs, target
//        descriptors are broadcast through the scalar data path (wave-uniform s_load of 33 floats
//        per target), 98 VALU ops per pair.  k_feature_match_scan is the plain scan (small problems);
//        k_feature_match_pruned visits key-ordered targets and skips 64-target boxes by an exact
//        33-D lower bound (same correspondences).
//  (ii)  index triples (registration.cpp:235-239): host, mt19937 + Lemire (ctx.hip), one batch at
//        a time; a batch is uploaded as int4 (i0,i1,i2,valid).
//  (iii) k_ransac_hypotheses: one lane per hypothesis — centroids, H = S_c T_c^T, Jacobi SVD,
//        R = V U^T with reflection fix, t = c_t - R c_s (registration.cpp:242-268).
//  (iv)  k_ransac_score: one hypothesis per lane (R,t in VGPRs); points (source p and its
//        pre-gathered match q, two points per component-interleaved record) are broadcast through
//        the scalar data path and processed two at a time with packed f32 ops;
//        28 VALU ops per (hypothesis, point); the inlier test sqrt(d2) < thr is evaluated as
//        d2 < tau with tau = min{f : sqrtf(f) >= thr} (exactly equivalent, no sqrt in the loop).
//        Inlier counts are integers: partial counts per point-split are added with integer
//        atomics, which are order-independent, so counts are bit-exact and reproducible.
//  (v)   selection (registration.cpp:281-290) on the host over the batch's counts in iteration
//        order: strict > on float(inliers)/ns, stop at the first fitness > confidence.
//  (vi)  k_ransac_rmse: error sum of the winning hypothesis only, fixed-order reduction.
#include "tdv_internal.hpp"
#include "device_linalg.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <algorithm>
#include <vector>

namespace tdv {

// ------------------------------------------------------------------ feature match
constexpr int FM_SPL = 2;
#ifndef FM_BLOCK_VALUE
#define FM_BLOCK_VALUE 256
#endif
constexpr int FM_BLOCK = FM_BLOCK_VALUE;
constexpr int FM_SRC_PER_BLOCK = FM_SPL * FM_BLOCK;
constexpr int FD = 33;
constexpr int FM_SEED = 256;   // targets of the seeding launch

// EARLY: partial-distance early exit.  dist accumulates non-negative terms in d order, and fl(a + b) >= a for b >= 0,
// so once the partial sum is >= the lane's best the final distance cannot pass the strict "<": a target is dropped
// as soon as that holds for every lane of the wave (checked after 11 and 22 of the 33 dimensions).  `seed` (the exact
// best over the first targets, computed by a first launch) lets every split start with a tight bound.
template <bool EARLY>
__global__ __launch_bounds__(FM_BLOCK)
void k_feature_match_scan(const float* __restrict__ fs, int ns, int ns_pad,
                          const float* __restrict__ ft, int j_begin, int j_end, int per_split,
                          const float* __restrict__ seed, const uint4* __restrict__ order,
                          float* __restrict__ pd, int* __restrict__ pj) {
    const int split = blockIdx.y;
    const int j0 = j_begin + split * per_split;
    const int j1 = min(j_end, j0 + per_split);
    const int base = blockIdx.x * FM_SRC_PER_BLOCK + threadIdx.x;
    float f[FM_SPL][FD];
    float best[FM_SPL]; int bj[FM_SPL]; int src[FM_SPL];
#pragma unroll
    for (int s = 0; s < FM_SPL; ++s) {
        // with `order` (records sorted by seed distance, .w = source index) a wave holds sources whose bounds are
        // alike, so it leaves a target as early as its typical lane does; results go back to the source's own row
        const int t = base + s * FM_BLOCK;
        src[s] = order ? (int)order[min(t, ns - 1)].w : t;
        const int i = min(src[s], ns - 1);
#pragma unroll
        for (int d = 0; d < FD; ++d) f[s][d] = fs[(size_t)i * FD + d];
        best[s] = seed ? seed[i] : FLT_MAX;   // a seed comes from lower target indices: strict < keeps the tie rule
        bj[s] = seed ? -1 : 0;
        if (order && t >= ns) src[s] = -1;    // padding lane: duplicate work, no output
    }
    for (int j = j0; j < j1; ++j) {
        const float* __restrict__ g = ft + (size_t)j * FD;  // wave-uniform -> scalar loads
        float q[FD];
#pragma unroll
        for (int d = 0; d < FD; ++d) q[d] = g[d];
        float dist[FM_SPL];
#pragma unroll
        for (int s = 0; s < FM_SPL; ++s) dist[s] = 0.f;
#pragma unroll
        for (int seg = 0; seg < 3; ++seg) {
#pragma unroll
            for (int s = 0; s < FM_SPL; ++s)
#pragma unroll
                for (int d = seg * 11; d < seg * 11 + 11; ++d) { float diff = f[s][d] - q[d]; dist[s] += diff * diff; }
            if (EARLY && seg < 2) {
                bool alive = false;
#pragma unroll
                for (int s = 0; s < FM_SPL; ++s) alive = alive || (dist[s] < best[s]);
                if (!__any(alive)) goto next_target;
            }
        }
#pragma unroll
        for (int s = 0; s < FM_SPL; ++s) {
            bool lt = dist[s] < best[s];
            best[s] = lt ? dist[s] : best[s];
            bj[s] = lt ? j : bj[s];
        }
    next_target:;
    }
#pragma unroll
    for (int s = 0; s < FM_SPL; ++s) {
        if (src[s] < 0) continue;
        size_t o = (size_t)split * ns_pad + src[s];
        pd[o] = best[s]; pj[o] = bj[s];
    }
}

// partial results are combined in launch/split order with strict <: the lowest target index wins ties
__global__ void k_feature_match_combine(int ns, int ns_pad, int nparts, const float* __restrict__ pd,
                                        const int* __restrict__ pj, int* __restrict__ corr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    float best = FLT_MAX; int bj = 0;
    for (int s = 0; s < nparts; ++s) {
        float d = pd[(size_t)s * ns_pad + i];
        if (d < best) { best = d; bj = pj[(size_t)s * ns_pad + i]; }
    }
    corr[i] = bj;
}

// ---- exact pruned descriptor match (large problems) -------------------------------------------------------------
// FPFH descriptors of a real part are strongly clustered (most of their variance lies along one direction), so both
// sides are ordered by a cheap scalar key (the three centre bins) with a counting sort, 33-D bounding boxes are built
// over runs of 64 ordered targets, and a wave of neighbouring sources skips every box whose lower bound exceeds all
// its lanes' current best.  The bound is the distance expression itself applied to the per-dimension gaps, summed in
// the same order: every term is <= the corresponding term of any target inside the box and float addition /
// multiplication are monotone, so lb <= fl(dist) holds exactly and no margin is needed.  Targets are visited
// inside-out from the wave's own key position; ties keep the lowest ORIGINAL target index, as the CPU scan does.
// The order only affects speed: any key (and the arbitrary order inside a bucket) gives the same correspondences.
constexpr int FMP_KEY_BITS = 7;                       // bits per key of the 2-D Morton bucket
constexpr int FMP_BUCKETS = 1 << (2 * FMP_KEY_BITS);   // 16384 (64 KB of LDS counters in the ordering kernels)
constexpr int FMP_BOX = 64;
constexpr int FMP_TWO_KEYS_MAX_TARGETS = 32768;

__device__ __forceinline__ int fm_bucket(const float* __restrict__ f, int two_keys) {
    // key 1: the three centre bins (descriptors sum to 1, so it lies in [0, 1]); key 2: the first moment of the phi
    // sub-histogram (in [0, 10]).  two_keys: FMP_KEY_BITS bits each, interleaved (a 128 x 128 Morton grid) — measured
    // better against a small model (C4: 128k x 9.4k, 0.71 -> 0.60 ms); else key 1 alone at full resolution — better
    // when the target side is large (100k x 100k: 8.3 vs 9.6 ms).  An offline study on real descriptors
    // (tools/studies/feature_match_box_pruning.py) put this pair ahead of every other cheap pair.
    const float c1 = f[5] + (f[16] + f[27]);
    if (!two_keys) {
        const float k = c1 * (float)FMP_BUCKETS;
        return (k == k) ? (int)fminf(fmaxf(k, 0.f), (float)(FMP_BUCKETS - 1)) : 0;
    }
    constexpr float LEVELS = (float)(1 << FMP_KEY_BITS);
    const float k1 = c1 * LEVELS;
    float k2 = 0.f;
#pragma unroll
    for (int b = 1; b < 11; ++b) k2 += (float)b * f[11 + b];
    k2 *= LEVELS * 0.1f;
    const unsigned a = (k1 == k1) ? (unsigned)fminf(fmaxf(k1, 0.f), LEVELS - 1.f) : 0u;
    const unsigned c = (k2 == k2) ? (unsigned)fminf(fmaxf(k2, 0.f), LEVELS - 1.f) : 0u;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < FMP_KEY_BITS; ++i) m |= (((a >> i) & 1u) << (2 * i + 1)) | (((c >> i) & 1u) << (2 * i));
    return (int)m;
}
// Real descriptors crowd a few buckets, so both passes count in an LDS histogram first (one global atomic per
// non-empty bucket and workgroup instead of one per row).
constexpr int FMP_SORT_BLOCK = 1024;
__global__ __launch_bounds__(FMP_SORT_BLOCK)
void k_fm_hist(const float* __restrict__ f, int n, int two_keys, int* __restrict__ bucket_of, int* __restrict__ hist) {
    __shared__ int h[FMP_BUCKETS];
    for (int b = threadIdx.x; b < FMP_BUCKETS; b += FMP_SORT_BLOCK) h[b] = 0;
    __syncthreads();
    const int i = blockIdx.x * FMP_SORT_BLOCK + threadIdx.x;
    if (i < n) {
        const int b = fm_bucket(f + (size_t)i * FD, two_keys);
        bucket_of[i] = b;
        atomicAdd(&h[b], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < FMP_BUCKETS; b += FMP_SORT_BLOCK) if (h[b]) atomicAdd(&hist[b], h[b]);
}
__global__ __launch_bounds__(FMP_SORT_BLOCK)
void k_fm_scatter(const int* __restrict__ bucket_of, int n, const int* __restrict__ start, int* __restrict__ cursor,
                  int* __restrict__ perm) {
    __shared__ int h[FMP_BUCKETS];      // rows of this workgroup per bucket, then the workgroup's base inside the bucket
    for (int b = threadIdx.x; b < FMP_BUCKETS; b += FMP_SORT_BLOCK) h[b] = 0;
    __syncthreads();
    const int i = blockIdx.x * FMP_SORT_BLOCK + threadIdx.x;
    int b = 0, local = 0;
    if (i < n) { b = bucket_of[i]; local = atomicAdd(&h[b], 1); }
    __syncthreads();
    for (int c = threadIdx.x; c < FMP_BUCKETS; c += FMP_SORT_BLOCK) if (h[c]) h[c] = atomicAdd(&cursor[c], h[c]);
    __syncthreads();
    if (i < n) perm[start[b] + h[b] + local] = i;   // order inside a bucket is irrelevant to the result
}
__global__ void k_fm_gather_targets(const float* __restrict__ ft, const int* __restrict__ perm, int nt, int nt_pad,
                                    float* __restrict__ T, int* __restrict__ torig) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nt_pad * FD) return;
    const int row = (int)(e / FD), d = (int)(e % FD);
    T[e] = row < nt ? ft[(size_t)perm[row] * FD + d] : INFINITY;   // padding rows: distance +inf, never chosen
    if (d == 0) torig[row] = row < nt ? perm[row] : INT_MAX;
}
__global__ void k_fm_boxes(const float* __restrict__ T, int nt, int nbox, float* __restrict__ bmin, float* __restrict__ bmax) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nbox * FD) return;
    const int b = e / FD, d = e % FD;
    float mn = INFINITY, mx = -INFINITY;
    for (int r = b * FMP_BOX; r < min(nt, (b + 1) * FMP_BOX); ++r) { float v = T[(size_t)r * FD + d]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    bmin[e] = mn; bmax[e] = mx;
}

// box visited at position v of the inside-out order centred at box c (bijection onto [0, nbox))
__device__ __forceinline__ int visit_inside_out(int v, int c, int nbox) {
    const int L = c, R = nbox - 1 - c;
    const int m = min(L, R);
    if (v <= 2 * m) { int k = (v + 1) >> 1; return (v & 1) ? c + k : c - k; }
    return R > L ? c + (v - m) : c - (v - m);
}

template <int SPL>
__global__ __launch_bounds__(FM_BLOCK)
void k_feature_match_pruned(const float* __restrict__ fs, const int* __restrict__ sperm, int ns, int ns_pad,
                            const float* __restrict__ T, const int* __restrict__ torig, int nbox,
                            const float* __restrict__ bmin, const float* __restrict__ bmax, const int* __restrict__ tstart,
                            int two_keys, int nsplit, float* __restrict__ pd, int* __restrict__ pj) {
    const int split = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wbase = (blockIdx.x * (FM_BLOCK / 64) + wave) * (64 * SPL);   // the wave's 64*SPL consecutive ordered sources
    float f[SPL][FD];
    float best[SPL]; int bj[SPL]; int src[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int t = wbase + s * 64 + lane;
        const int i = sperm[min(t, ns - 1)];
        src[s] = t < ns ? i : -1;   // padding lanes duplicate the last source and write nothing
#pragma unroll
        for (int d = 0; d < FD; ++d) f[s][d] = fs[(size_t)i * FD + d];
        best[s] = INFINITY; bj[s] = INT_MAX;
    }
    // start where the targets with the wave's own key begin
    const int c = min(nbox - 1, tstart[__builtin_amdgcn_readfirstlane(fm_bucket(f[0], two_keys))] / FMP_BOX);
    for (int v = split; v < nbox; v += nsplit) {
        const int b = visit_inside_out(v, c, nbox);
        const float* __restrict__ lo = bmin + (size_t)b * FD;   // wave-uniform -> scalar loads
        const float* __restrict__ hi = bmax + (size_t)b * FD;
        float lb[SPL];
#pragma unroll
        for (int s = 0; s < SPL; ++s) lb[s] = 0.f;
#pragma unroll
        for (int d = 0; d < FD; ++d) {
            const float l = lo[d], h = hi[d];
#pragma unroll
            for (int s = 0; s < SPL; ++s) { float g = fmaxf(fmaxf(l - f[s][d], f[s][d] - h), 0.f); lb[s] += g * g; }
        }
        bool alive = false;
#pragma unroll
        for (int s = 0; s < SPL; ++s) alive = alive || (lb[s] <= best[s]);   // <=: an equal distance with a lower index still wins
        if (!__any(alive)) continue;
#pragma unroll 1
        for (int t = 0; t < FMP_BOX; ++t) {
            const int j = b * FMP_BOX + t;
            const float* __restrict__ g = T + (size_t)j * FD;
            const int o = torig[j];
            float q[FD];
#pragma unroll
            for (int d = 0; d < FD; ++d) q[d] = g[d];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                float dist = 0.f;
#pragma unroll
                for (int d = 0; d < FD; ++d) { float diff = f[s][d] - q[d]; dist += diff * diff; }
                const bool take = dist < best[s] || (dist == best[s] && o < bj[s]);
                best[s] = take ? dist : best[s];
                bj[s] = take ? o : bj[s];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        if (src[s] < 0) continue;
        const size_t o = (size_t)split * ns_pad + src[s];
        pd[o] = best[s]; pj[o] = bj[s];
    }
}

// partials of the pruned match: lexicographic (distance, original index) minimum, order-independent
__global__ void k_feature_match_combine_lex(int ns, int ns_pad, int nparts, const float* __restrict__ pd,
                                            const int* __restrict__ pj, int* __restrict__ corr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    float best = INFINITY; int bj = INT_MAX;
    for (int s = 0; s < nparts; ++s) {
        const float d = pd[(size_t)s * ns_pad + i]; const int j = pj[(size_t)s * ns_pad + i];
        if (d < best || (d == best && j < bj)) { best = d; bj = j; }
    }
    corr[i] = bj == INT_MAX ? 0 : bj;   // nothing finite: the CPU loop keeps its initial index 0
}

namespace {
// counting sort of n descriptors by key bucket: perm (ordered position -> row) and, optionally, the bucket starts
int fm_order(tdv_ctx* ctx, const float* d_f, int n, int two_keys, int* perm, int* start /* FMP_BUCKETS + 1 */) {
    hipStream_t s = ctx->stream;
    int *hist, *cursor, *d_total, *bucket_of;
    TDV_TRY(ws_alloc(ctx, (size_t)FMP_BUCKETS, &hist));
    TDV_TRY(ws_alloc(ctx, (size_t)FMP_BUCKETS, &cursor));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &bucket_of));
    TDV_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)FMP_BUCKETS * 4, s));
    TDV_HIP(ctx, hipMemsetAsync(cursor, 0, (size_t)FMP_BUCKETS * 4, s));
    const int blocks = (n + FMP_SORT_BLOCK - 1) / FMP_SORT_BLOCK;
    k_fm_hist<<<blocks, FMP_SORT_BLOCK, 0, s>>>(d_f, n, two_keys, bucket_of, hist);
    TDV_TRY(exclusive_scan_dev(ctx, hist, FMP_BUCKETS, start, d_total));
    k_fm_scatter<<<blocks, FMP_SORT_BLOCK, 0, s>>>(bucket_of, n, start, cursor, perm);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

constexpr int FMP_SPL = 1;   // 1 measured better than 2 (C4: 0.84 vs 0.93 ms)

int feature_match_pruned_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr) {
    hipStream_t s = ctx->stream;
    const int nt_pad = (int)align_up((size_t)nt, FMP_BOX);
    const int nbox = nt_pad / FMP_BOX;
    constexpr int SRC_PER_BLOCK = FM_BLOCK * FMP_SPL;
    const int ns_pad = (int)align_up((size_t)ns, SRC_PER_BLOCK);
    const int blocks_x = ns_pad / SRC_PER_BLOCK;
    int want = (4096 + blocks_x - 1) / blocks_x;
    const int nsplit = std::max(1, std::min(std::min(want, std::max(1, nbox / 8)), 32));
    int *sperm, *tperm, *tstart, *sstart, *torig; float *T, *bmin, *bmax, *pd; int* pj;
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &sperm));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &tperm));
    TDV_TRY(ws_alloc(ctx, (size_t)FMP_BUCKETS + 1, &tstart));
    TDV_TRY(ws_alloc(ctx, (size_t)FMP_BUCKETS + 1, &sstart));
    TDV_TRY(ws_alloc(ctx, (size_t)nt_pad, &torig));
    TDV_TRY(ws_alloc(ctx, (size_t)nt_pad * FD, &T));
    TDV_TRY(ws_alloc(ctx, (size_t)nbox * FD, &bmin));
    TDV_TRY(ws_alloc(ctx, (size_t)nbox * FD, &bmax));
    TDV_TRY(ws_alloc(ctx, (size_t)nsplit * ns_pad, &pd));
    TDV_TRY(ws_alloc(ctx, (size_t)nsplit * ns_pad, &pj));
    ScopedTimer tm(ctx, TDV_TIMER_FEATURE_MATCH);
    const int two_keys = nt <= FMP_TWO_KEYS_MAX_TARGETS ? 1 : 0;
    TDV_TRY(fm_order(ctx, d_ft, nt, two_keys, tperm, tstart));
    TDV_TRY(fm_order(ctx, d_fs, ns, two_keys, sperm, sstart));
    k_fm_gather_targets<<<(unsigned)(((size_t)nt_pad * FD + 255) / 256), 256, 0, s>>>(d_ft, tperm, nt, nt_pad, T, torig);
    k_fm_boxes<<<(nbox * FD + 255) / 256, 256, 0, s>>>(T, nt, nbox, bmin, bmax);
    k_feature_match_pruned<FMP_SPL><<<dim3(blocks_x, nsplit), FM_BLOCK, 0, s>>>(d_fs, sperm, ns, ns_pad, T, torig, nbox, bmin, bmax, tstart,
                                                                               two_keys, nsplit, pd, pj);
    k_feature_match_combine_lex<<<(ns + 255) / 256, 256, 0, s>>>(ns, ns_pad, nsplit, pd, pj, d_corr);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}
}  // namespace

int feature_match_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr) {
    if (!ctx || !d_fs || !d_ft || !d_corr || ns < 0 || nt < 0) return TDV_ERR_BAD_ARG;
    if (ns == 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    if (nt == 0) { TDV_HIP(ctx, hipMemsetAsync(d_corr, 0, (size_t)ns * 4, s)); return TDV_OK; }
    static const bool pruned_ok = getenv("TDV_FM_BRUTE") == nullptr;         // A/B knob: same results either way
    if (pruned_ok && ns >= 4096 && nt >= 2048) return feature_match_pruned_dev(ctx, d_fs, ns, d_ft, nt, d_corr);
    static const bool early = getenv("TDV_FM_NO_EARLY_EXIT") == nullptr;   // A/B knob: same results either way
    const int ns_pad = (int)align_up((size_t)ns, FM_SRC_PER_BLOCK);
    const int blocks_x = ns_pad / FM_SRC_PER_BLOCK;
    // part 0: the first n_seed targets in one split (its exact best seeds the bound of every later split)
    const int n_seed = early ? std::min(nt, FM_SEED) : 0;
    const int rest = nt - n_seed;
    int want = (4096 + blocks_x - 1) / blocks_x;
    int nsplit = rest > 0 ? std::max(1, std::min(std::min(want, std::max(1, rest / 64)), 64)) : 0;
    int per_split = nsplit ? (rest + nsplit - 1) / nsplit : 0;
    nsplit = nsplit ? (rest + per_split - 1) / per_split : 0;
    const int nparts = nsplit + (n_seed ? 1 : 0);
    float* pd; int* pj;
    TDV_TRY(ws_alloc(ctx, (size_t)nparts * ns_pad, &pd));
    TDV_TRY(ws_alloc(ctx, (size_t)nparts * ns_pad, &pj));
    {
        ScopedTimer tm(ctx, TDV_TIMER_FEATURE_MATCH);
        if (early) {
            k_feature_match_scan<true><<<dim3(blocks_x, 1), FM_BLOCK, 0, s>>>(d_fs, ns, ns_pad, d_ft, 0, n_seed, n_seed, nullptr, nullptr, pd, pj);
            if (nsplit)   // (ordering the sources by seed distance was measured: no gain on FPFH descriptors, so rows stay in place)
                k_feature_match_scan<true><<<dim3(blocks_x, nsplit), FM_BLOCK, 0, s>>>(d_fs, ns, ns_pad, d_ft, n_seed, nt, per_split, pd, nullptr,
                                                                                      pd + ns_pad, pj + ns_pad);
        } else {
            k_feature_match_scan<false><<<dim3(blocks_x, nsplit), FM_BLOCK, 0, s>>>(d_fs, ns, ns_pad, d_ft, 0, nt, per_split, nullptr, nullptr, pd, pj);
        }
    }
    k_feature_match_combine<<<(ns + 255) / 256, 256, 0, s>>>(ns, ns_pad, nparts, pd, pj, d_corr);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

// ------------------------------------------------------------------ hypotheses
// pq layout: 8 floats per point: px py pz qx qy qz 0 0  (q = tgt[corr[i]]); padding points have
// p = 0 and q = +inf so that d2 = +inf and they are never inliers.
__global__ void k_gather_pq(const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ corr,
                            int ns, int ns_pad, float* __restrict__ pq) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns_pad) return;
    float4 a, b;
    if (i < ns) {
        int c = corr[i];
        a = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], tgt[3 * c]);
        b = make_float4(tgt[3 * c + 1], tgt[3 * c + 2], 0.f, 0.f);
    } else {
        a = make_float4(0.f, 0.f, 0.f, INFINITY);
        b = make_float4(INFINITY, INFINITY, 0.f, 0.f);
    }
    reinterpret_cast<float4*>(pq)[2 * (size_t)i] = a;
    reinterpret_cast<float4*>(pq)[2 * (size_t)i + 1] = b;
}

// hyp layout: SoA [12][h_pad]: r00 r10 r20 r01 r11 r21 r02 r12 r22 t0 t1 t2 (column-major R).
// Invalid (skipped) iterations get NaN so that no comparison is ever true -> 0 inliers.
__global__ void k_ransac_hypotheses(const float* __restrict__ pq, const int4* __restrict__ triples, int count, int h_pad,
                                    float* __restrict__ hyp) {
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= h_pad) return;
    float o[12];
    bool valid = false;
    int4 tr = make_int4(0, 0, 0, 0);
    if (h < count) { tr = triples[h]; valid = tr.w != 0; }
    if (valid) {
        const int id[3] = {tr.x, tr.y, tr.z};
        float sp[3][3], tp[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* r = pq + (size_t)id[k] * 8;
            sp[k][0] = r[0]; sp[k][1] = r[1]; sp[k][2] = r[2];
            tp[k][0] = r[3]; tp[k][1] = r[4]; tp[k][2] = r[5];
        }
        float sc[3], tc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            sc[r] = dl::s3(sp[0][r], sp[1][r], sp[2][r]) / 3.0f;
            tc[r] = dl::s3(tp[0][r], tp[1][r], tp[2][r]) / 3.0f;
        }
        dl::Mat3 S, Tt;  // S(r,c) = centred source column c ; Tt = (centred target)^T
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) { dl::el(S, r, c) = sp[c][r] - sc[r]; dl::el(Tt, c, r) = tp[c][r] - tc[r]; }
        dl::Mat3 H = dl::mul3(S, Tt);
        dl::Mat3 R = dl::kabsch_rotation(H);
        float rx, ry, rz;
        dl::mulv3(R, sc[0], sc[1], sc[2], rx, ry, rz);
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = R.a[k];
        o[9] = tc[0] - rx; o[10] = tc[1] - ry; o[11] = tc[2] - rz;
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = __builtin_nanf("");
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) hyp[(size_t)k * h_pad + h] = o[k];
}

// ------------------------------------------------------------------ scoring
#ifndef RS_BLOCK_VALUE
#define RS_BLOCK_VALUE 1024
#endif
#ifndef RS_WG_TARGET
#define RS_WG_TARGET 6144
#endif
constexpr int RS_BLOCK = RS_BLOCK_VALUE;   // 16 waves per workgroup walk the same points together (measured at 200k x 57k hyps: 128 threads 74 %,
                                           // 256 82 %, 512 85 %, 1024 87 % of the VALU peak; 3-12k workgroups make no difference)
constexpr int RS_HYP_PER_BLOCK = RS_BLOCK;   // one hypothesis per lane (measured best: R,t in 24 VGPRs, highest occupancy)
#ifndef RS_PCH_VALUE
#define RS_PCH_VALUE 8
#endif
constexpr int RS_PCH = RS_PCH_VALUE;   // points per scalar chunk: 4 records of 12 floats (8 measured 81 % of peak, 4: 80 %)

// The scoring loop reads a second pair array that holds TWO points per record, component-interleaved
// [px0 px1 | py0 py1 | pz0 pz1 | qx0 qx1 | qy0 qy1 | qz0 qz1] (48 B per 2 points), so that every arithmetic op is one
// v_pk_*_f32 on an aligned SGPR pair — the same per-element IEEE operations as a scalar loop, half the
// instructions, no SGPR shuffling (measured 81 % of the VALU peak against 79 % for the loop vectoriser's packing of
// the 8-float layout and 71 % for scalar code).
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_pack_pq2(const float* __restrict__ pq, int ns_pad, float* __restrict__ pq2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // pair index
    if (2 * i >= ns_pad) return;
    const float* a = pq + (size_t)(2 * i) * 8; const float* b = a + 8;
    float* o = pq2 + (size_t)i * 12;
#pragma unroll
    for (int c = 0; c < 6; ++c) { o[2 * c] = a[c]; o[2 * c + 1] = b[c]; }
}
__global__ __launch_bounds__(RS_BLOCK)
void k_ransac_score(const float* __restrict__ hyp, int h_pad, const float* __restrict__ pq2,
                       int n_pchunks, int pchunks_per_split, float tau, int* __restrict__ counts) {
    const int split = blockIdx.y;
    const int c0 = split * pchunks_per_split;
    const int c1 = min(n_pchunks, c0 + pchunks_per_split);
    const int base = blockIdx.x * RS_BLOCK + threadIdx.x;
    v2f r[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) { const float t = hyp[(size_t)e * h_pad + base]; r[e] = (v2f){t, t}; }
    int cnt = 0;
    for (int c = c0; c < c1; ++c) {
        const float* __restrict__ g = pq2 + (size_t)c * (6 * RS_PCH);  // RS_PCH points = RS_PCH/2 records of 12 floats, wave-uniform
        float v[6 * RS_PCH];
#pragma unroll
        for (int e = 0; e < 6 * RS_PCH; ++e) v[e] = g[e];
#pragma unroll
        for (int p = 0; p < RS_PCH / 2; ++p) {
            const v2f px = {v[12 * p + 0], v[12 * p + 1]}, py = {v[12 * p + 2], v[12 * p + 3]}, pz = {v[12 * p + 4], v[12 * p + 5]};
            const v2f qx = {v[12 * p + 6], v[12 * p + 7]}, qy = {v[12 * p + 8], v[12 * p + 9]}, qz = {v[12 * p + 10], v[12 * p + 11]};
            const v2f x = (r[0] * px + (r[3] * py + r[6] * pz)) + r[9];
            const v2f y = (r[1] * px + (r[4] * py + r[7] * pz)) + r[10];
            const v2f z = (r[2] * px + (r[5] * py + r[8] * pz)) + r[11];
            const v2f dx = x - qx, dy = y - qy, dz = z - qz;
            const v2f d2 = dx * dx + (dy * dy + dz * dz);
            cnt += (d2.x < tau) ? 1 : 0;
            cnt += (d2.y < tau) ? 1 : 0;
        }
    }
    atomicAdd(&counts[base], cnt);
}

// error sum of one hypothesis (column-major R in T[0..8], t in T[9..11]) over all points
__global__ __launch_bounds__(256)
void k_ransac_rmse_partial(const float* __restrict__ pq, int ns, const float* __restrict__ hyp12, float tau,
                           double* __restrict__ slabs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double e = 0.0, n = 0.0;
    if (i < ns) {
        const float* g = pq + (size_t)i * 8;
        float px = g[0], py = g[1], pz = g[2];
        float x = (hyp12[0] * px + (hyp12[3] * py + hyp12[6] * pz)) + hyp12[9];
        float y = (hyp12[1] * px + (hyp12[4] * py + hyp12[7] * pz)) + hyp12[10];
        float z = (hyp12[2] * px + (hyp12[5] * py + hyp12[8] * pz)) + hyp12[11];
        float dx = x - g[3], dy = y - g[4], dz = z - g[5];
        float d2 = dx * dx + (dy * dy + dz * dz);
        if (d2 < tau) { float err = sqrtf(d2); e = (double)(err * err); n = 1.0; }  // d2 < tau <=> sqrtf(d2) < thr
    }
    __shared__ double red[2][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off, 64); n += __shfl_down(n, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = e; red[1][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        slabs[2 * (size_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        slabs[2 * (size_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ void k_ransac_rmse_final(const double* __restrict__ slabs, int nblocks, double* __restrict__ out2) {
    __shared__ double pe[256], pn[256];
    double e = 0.0, n = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { e += slabs[2 * (size_t)b]; n += slabs[2 * (size_t)b + 1]; }
    pe[threadIdx.x] = e; pn[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { pe[threadIdx.x] += pe[threadIdx.x + off]; pn[threadIdx.x] += pn[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = pe[0]; out2[1] = pn[0]; }
}

int ransac_run_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt,
                   const float* d_fs, const float* d_ft, const int* d_corr_in,
                   float voxel, int max_iterations, float confidence, uint32_t seed,
                   tdv_ransac_result* out, int* trace_inliers) {
    if (!ctx || !out || ns < 0 || nt < 0 || max_iterations < 0) return TDV_ERR_BAD_ARG;
    if (ns > 0 && (!d_src || !d_tgt)) return TDV_ERR_BAD_ARG;
    if (!d_corr_in && ns > 0 && nt > 0 && (!d_fs || !d_ft)) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    // RegistrationResult defaults (include/registration.hpp:26-30)
    for (int i = 0; i < 16; ++i) out->T[i] = (i % 5 == 0) ? 1.f : 0.f;
    out->fitness = 0.f; out->rmse = 0.f; out->inliers = 0; out->best_iteration = -1; out->iterations_run = 0;
    if (ns == 0 || nt == 0 || max_iterations == 0) return TDV_OK;  // uniform_int over an empty range is UB in the reference
    hipStream_t s = ctx->stream;
    const float thr = voxel * 1.5f;  // registration.cpp:213
    const float tau = tau_lt(thr);

    const int* d_corr = d_corr_in;
    if (!d_corr) {
        int* c = nullptr;
        TDV_TRY(ws_alloc(ctx, (size_t)ns, &c));
        TDV_TRY(feature_match_dev(ctx, d_fs, ns, d_ft, nt, c));
        d_corr = c;
    }
    const int ns_pad = (int)align_up((size_t)ns, (size_t)RS_PCH * 64);
    float* pq = nullptr;
    TDV_TRY(ws_alloc(ctx, (size_t)ns_pad * 8, &pq));
    k_gather_pq<<<(ns_pad + 255) / 256, 256, 0, s>>>(d_src, d_tgt, d_corr, ns, ns_pad, pq);
    float* pq2 = nullptr;
    TDV_TRY(ws_alloc(ctx, (size_t)ns_pad * 6, &pq2));
    k_pack_pq2<<<(ns_pad / 2 + 255) / 256, 256, 0, s>>>(pq, ns_pad, pq2);
    TDV_CHECK_LAUNCH(ctx);

    // batch size: enough hypotheses to fill the chip, bounded for early exit granularity
    const int batch = std::min(std::max(max_iterations, 1), 65536);  // per-batch host sync is ~0.3 ms: amortise it
    const int h_pad = (int)align_up((size_t)batch, RS_HYP_PER_BLOCK);
    const int hblocks = h_pad / RS_HYP_PER_BLOCK;
    const int n_pchunks = ns_pad / RS_PCH;
    int want = (RS_WG_TARGET + hblocks - 1) / hblocks;
    int psplit = std::max(1, std::min(std::min(want, std::max(1, n_pchunks / 32)), 512));
    int pchunks_per_split = (n_pchunks + psplit - 1) / psplit;
    psplit = (n_pchunks + pchunks_per_split - 1) / pchunks_per_split;

    // two sets of batch buffers: batch k+1 is prepared on the host (index stream, triple packing) and enqueued while
    // the GPU scores batch k; results are consumed in iteration order, so the outcome is that of the sequential loop
    float* hyp[2] = {nullptr, nullptr}; int* counts[2] = {nullptr, nullptr}; int4* d_tri[2] = {nullptr, nullptr};
    double* slabs = nullptr; double* d_out2 = nullptr; float* d_best12 = nullptr;
    for (int q = 0; q < 2; ++q) {
        TDV_TRY(ws_alloc(ctx, (size_t)12 * h_pad, &hyp[q]));
        TDV_TRY(ws_alloc(ctx, (size_t)h_pad, &counts[q]));
        TDV_TRY(ws_alloc(ctx, (size_t)batch, &d_tri[q]));
    }
    const int rblocks = (ns + 255) / 256;
    TDV_TRY(ws_alloc(ctx, (size_t)2 * rblocks, &slabs));
    TDV_TRY(ws_alloc(ctx, 2, &d_out2));
    TDV_TRY(ws_alloc(ctx, 12, &d_best12));
    // pinned: 2 x triples (int4 * batch) | 2 x counts (int * batch) | best12 (12 floats) | out2 (2 doubles)
    const size_t sz_tri = align_up((size_t)batch * 16, 64), sz_cnt = align_up((size_t)batch * 4, 64);
    const size_t pin_b12 = 2 * sz_tri + 2 * sz_cnt, pin_o2 = pin_b12 + 64, pin_total = pin_o2 + 64;
    TDV_TRY(pin_reserve(ctx, pin_total));
    int4* h_tri[2] = {reinterpret_cast<int4*>(ctx->pin), reinterpret_cast<int4*>(ctx->pin + sz_tri)};
    int* h_cnt[2] = {reinterpret_cast<int*>(ctx->pin + 2 * sz_tri), reinterpret_cast<int*>(ctx->pin + 2 * sz_tri + sz_cnt)};
    float* h_b12 = reinterpret_cast<float*>(ctx->pin + pin_b12);
    double* h_o2 = reinterpret_cast<double*>(ctx->pin + pin_o2);
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int q = 0; q < 2; ++q) TDV_HIP(ctx, hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));

    TripleStream stream_idx(seed, (uint64_t)ns);   // sequential over the whole run (registration.cpp:235-239)
    auto prepare = [&](int q, int it0) -> int {    // host: draw + pack the triples of one batch
        const int cnt = std::min(batch, max_iterations - it0);
        uint64_t d[3];
        for (int k = 0; k < cnt; ++k) {
            stream_idx.next(d);
            int valid = !(d[0] == d[1] || d[1] == d[2] || d[0] == d[2]);  // registration.cpp:240
            h_tri[q][k] = make_int4((int)d[0], (int)d[1], (int)d[2], valid);
        }
        return cnt;
    };
    auto enqueue = [&](int q, int cnt) -> int {     // device: hypotheses + scoring + counts back to the host
        TDV_HIP(ctx, hipMemcpyAsync(d_tri[q], h_tri[q], (size_t)cnt * 16, hipMemcpyHostToDevice, s));
        TDV_HIP(ctx, hipMemsetAsync(counts[q], 0, (size_t)h_pad * 4, s));
        k_ransac_hypotheses<<<(h_pad + 255) / 256, 256, 0, s>>>(pq, d_tri[q], cnt, h_pad, hyp[q]);
        const int hb = (int)(align_up((size_t)cnt, RS_HYP_PER_BLOCK) / RS_HYP_PER_BLOCK);
        {
            ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
            k_ransac_score<<<dim3(hb, psplit), RS_BLOCK, 0, s>>>(hyp[q], h_pad, pq2, n_pchunks, pchunks_per_split, tau, counts[q]);
        }
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipMemcpyAsync(h_cnt[q], counts[q], (size_t)cnt * 4, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipEventRecord(ev[q], s));
        return TDV_OK;
    };

    float best_fitness = 0.f; int best_iter = -1, best_inliers = 0; bool stop = false;
    int done_iters = 0;
    int status = TDV_OK;
    int cur = 0, it0 = 0;
    int cnt_cur = prepare(cur, it0);
    status = enqueue(cur, cnt_cur);
    while (status == TDV_OK && cnt_cur > 0 && !stop) {
        const int nxt = cur ^ 1;
        const int it_next = it0 + cnt_cur;
        int cnt_next = 0;
        if (it_next < max_iterations) {             // overlap: prepare and enqueue the next batch behind the current one
            cnt_next = prepare(nxt, it_next);
            status = enqueue(nxt, cnt_next);
            if (status != TDV_OK) break;
        }
        if (hipEventSynchronize(ev[cur]) != hipSuccess) { status = TDV_ERR_LAUNCH; break; }
        int batch_best = -1;
        for (int k = 0; k < cnt_cur; ++k) {
            done_iters = it0 + k + 1;
            if (!h_tri[cur][k].w) { if (trace_inliers) trace_inliers[it0 + k] = -1; continue; }
            int inl = h_cnt[cur][k];
            if (trace_inliers) trace_inliers[it0 + k] = inl;
            float fitness = static_cast<float>(inl) / static_cast<float>((size_t)ns);  // registration.cpp:281
            if (fitness > best_fitness) { best_fitness = fitness; best_iter = it0 + k; best_inliers = inl; batch_best = k; }
            if (fitness > confidence) { stop = true; break; }
        }
        if (batch_best >= 0) {  // keep the winning (R,t) of this batch (hyp[cur] is not overwritten before batch cur+2 is enqueued)
            hipError_t e = hipMemcpy2DAsync(d_best12, 4, hyp[cur] + batch_best, (size_t)h_pad * 4, 4, 12, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) { status = set_err(ctx, e, "hipMemcpy2DAsync", __LINE__); break; }
        }
        cur = nxt; it0 = it_next; cnt_cur = cnt_next;
    }
    (void)hipStreamSynchronize(s);   // a speculative batch may still be in flight after an early exit
    for (int q = 0; q < 2; ++q) (void)hipEventDestroy(ev[q]);
    if (status != TDV_OK) return status;
    out->iterations_run = done_iters;
    if (best_iter >= 0) {
        k_ransac_rmse_partial<<<rblocks, 256, 0, s>>>(pq, ns, d_best12, tau, slabs);
        k_ransac_rmse_final<<<1, 256, 0, s>>>(slabs, rblocks, d_out2);
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipMemcpyAsync(h_b12, d_best12, 48, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipMemcpyAsync(h_o2, d_out2, 16, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipStreamSynchronize(s));
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) out->T[c * 4 + r] = h_b12[c * 3 + r];
        out->T[12] = h_b12[9]; out->T[13] = h_b12[10]; out->T[14] = h_b12[11];
        out->fitness = best_fitness;
        out->inliers = best_inliers;
        out->best_iteration = best_iter;
        // registration.cpp:282 (float total_error / int inliers)
        out->rmse = best_inliers > 0 ? std::sqrt((float)h_o2[0] / (float)best_inliers) : 999.0f;
        if ((int)(h_o2[1] + 0.5) != best_inliers) {
            snprintf(ctx->err, sizeof(ctx->err), "ransac: rmse pass counted %d inliers, scoring pass %d", (int)(h_o2[1] + 0.5), best_inliers);
            return TDV_ERR_INTERNAL;
        }
    }
    return TDV_OK;
}

}  // namespace tdv
