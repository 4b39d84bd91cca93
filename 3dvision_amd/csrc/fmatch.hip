// Descriptor correspondences of RANSAC on gfx950 (f32 VALU work; nothing here is HBM- or MFMA-bound).
//
// Replaces the matching loop of Registration::ransacRegistration (/root/reference/src/registration.cpp:216-232):
// for every source descriptor the target with the smallest 33-D squared distance, accumulated in d order without FMA,
// strict <, lowest target index on ties.  Three implementations with identical results:
//   * k_feature_match_scan      the reference's scan (small problems, TDV_FM_BRUTE=1): source descriptors in VGPRs,
//                               targets broadcast through the scalar data path, 98 VALU ops per pair;
//   * k_fm_query (default)      exact search over a packed index of the targets (below): STR packing along the
//                               targets' principal directions, 33-D boxes of 64-row leaves and 4096-row groups;
//   * k_feature_match_pruned    round 1's pruned scan over a scalar key order (TDV_FM_KEYORDER=1, kept for A/B).
// Every distance that is evaluated is the reference's expression; every target that is not evaluated is excluded by a
// box lower bound computed with the same expression on the per-dimension gaps (float sub/mul/add are monotone, so
// lb <= fl(dist) for every row of the box, no margin), or by the tie rule (equal bound, only higher indices inside).
// The order of rows, the principal directions and the host eigen-solver only decide WHICH rows are looked at first.
#include "tdv_internal.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace tdv {

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
// `list` / `n_list`: scan only these sources (the sources the packed-index search gave up on), results at the source's own
// row.  `seed`: a per-source starting bound.  SEED_FROM_LOWER: the seed is the exact best over LOWER target indices
// (part 0 seeding the later splits), so strict < keeps the lowest-index rule.  Otherwise the seed is a distance found
// somewhere in the table: the scan starts one ulp above it and walks every target in ascending order with strict <, so
// it finds that distance again and keeps the lowest index that reaches the minimum.
template <bool EARLY, bool SEED_FROM_LOWER>
__global__ __launch_bounds__(FM_BLOCK)
void k_feature_match_scan(const float* __restrict__ fs, int ns, int ns_pad,
                          const float* __restrict__ ft, int j_begin, int j_end, int per_split,
                          const float* __restrict__ seed, const int* __restrict__ list, int n_list,
                          float* __restrict__ pd, int* __restrict__ pj) {
    const int split = blockIdx.y;
    const int j0 = j_begin + split * per_split;
    const int j1 = min(j_end, j0 + per_split);
    const int base = blockIdx.x * FM_SRC_PER_BLOCK + threadIdx.x;
    const int n_here = list ? n_list : ns;
    float f[FM_SPL][FD];
    float best[FM_SPL]; int bj[FM_SPL]; int src[FM_SPL];
#pragma unroll
    for (int s = 0; s < FM_SPL; ++s) {
        const int t = base + s * FM_BLOCK;
        src[s] = list ? list[min(t, n_here - 1)] : t;
        const int i = min(max(src[s], 0), ns - 1);
#pragma unroll
        for (int d = 0; d < FD; ++d) f[s][d] = fs[(size_t)i * FD + d];
        if (seed) {
            const float sd = seed[i];
            best[s] = SEED_FROM_LOWER ? sd : ((sd < FLT_MAX && sd >= 0.f) ? __int_as_float(__float_as_int(sd) + 1) : FLT_MAX);
            bj[s] = -1;
        } else { best[s] = FLT_MAX; bj[s] = 0; }
        if (t >= n_here || src[s] < 0) src[s] = -1;    // padding lane: duplicate work, no output
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
        if (src[s] < 0 || src[s] >= ns) continue;
        size_t o = (size_t)split * ns_pad + src[s];
        pd[o] = best[s]; pj[o] = bj[s];
    }
}
// the listed sources' partial results, combined in split order with strict < (lowest target index wins ties)
__global__ void k_feature_match_combine_list(const int* __restrict__ list, int n_list, int ns, int ns_pad, int nparts, const float* __restrict__ pd,
                                             const int* __restrict__ pj, int* __restrict__ corr) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_list) return;
    const int i = list[t];
    if (i < 0 || i >= ns) return;
    float best = FLT_MAX; int bj = 0;
    for (int p0 = 0; p0 < nparts; p0 += 8) {   // 16 loads in flight
        float d[8]; int j[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = min(p0 + u, nparts - 1);
            d[u] = pd[(size_t)p * ns_pad + i]; j[u] = pj[(size_t)p * ns_pad + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (p0 + u < nparts && j[u] >= 0 && d[u] < best) { best = d[u]; bj = j[u]; }
    }
    corr[i] = bj;
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
#ifdef TDV_STUDY
constexpr int FMP_BOX = 64;
constexpr int FMP_TWO_KEYS_MAX_TARGETS = 32768;
#endif

#ifdef TDV_STUDY
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
#endif  // TDV_STUDY
// Real descriptors crowd a few buckets, so both passes count in an LDS histogram first (one global atomic per
// non-empty bucket and workgroup instead of one per row).
constexpr int FMP_SORT_BLOCK = 1024;
#ifdef TDV_STUDY
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
#endif  // TDV_STUDY
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
#ifdef TDV_STUDY
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
#endif  // TDV_STUDY

// The set bit of m nearest to position c, the higher one on a tie (the inside-out order c, c+1, c-1, c+2, ... restricted to
// the set bits), or -1: two shifts, a find-first and a count-leading on the wave's scalar unit instead of walking the
// positions one by one (that walk was 2,000 of the 4,800 instructions of a wave of the descriptor search).
__device__ __forceinline__ int nearest_set_bit(unsigned long long m, int c) {
    if (!m) return -1;
    const unsigned long long up = m >> c;                                  // bit 0 = position c
    const unsigned long long dn = c > 0 ? m << (64 - c) : 0ull;            // bit 63 = position c - 1
    const int du = up ? __ffsll((long long)up) - 1 : 128;
    const int dd = dn ? __clzll((long long)dn) + 1 : 128;
    return du <= dd ? c + du : c - dd;
}

// box visited at position v of the inside-out order centred at box c (bijection onto [0, nbox))
__device__ __forceinline__ int visit_inside_out(int v, int c, int nbox) {
    const int L = c, R = nbox - 1 - c;
    const int m = min(L, R);
    if (v <= 2 * m) { int k = (v + 1) >> 1; return (v & 1) ? c + k : c - k; }
    return R > L ? c + (v - m) : c - (v - m);
}

#ifdef TDV_STUDY
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
#endif  // TDV_STUDY

namespace {
#ifdef TDV_STUDY
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
#endif  // TDV_STUDY

#ifdef TDV_STUDY
constexpr int FMP_SPL = 1;   // 1 measured better than 2 (C4: 0.84 vs 0.93 ms)
int feature_match_keyorder_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr) {
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
#endif  // TDV_STUDY
}  // namespace

// ---- packed target index -----------------------------------------------------------------------------------------
// FPFH descriptors of a surface live close to a 3-D manifold of R^33 (96 % of their variance in three principal
// directions on the relief part).  The targets are therefore packed sort-tile-recursive along those directions:
// equal-count slabs along p0, equal-count columns along p1 inside every slab, rows sorted along p2 inside every column
// (two full sorts of 16-B records and one segmented sort inside the columns; slab / column counts proportional to the spread, chosen on the host from the
// eigenvalues).  Columns are padded to a multiple of 64 rows (+inf rows that never win), so a leaf = 64 consecutive
// rows never straddles two columns; group = 64 consecutive leaves.  Offline study on real descriptors
// (tools/studies/feature_match_pca_tree.py): a wave of 64 neighbouring sources has to open 2.3 % of the leaves with
// this packing against 17.8 % with round 1's scalar key.
constexpr int FX_LEAF = 64;
constexpr int FX_GROUP = 64;          // leaves per group
constexpr int FX_MAX_S = 64;          // slabs / columns per slab at most
constexpr int FX_NMOM = 561 + 33;     // upper triangle of sum f f^T, then sum f
#ifndef FMQ_WAVES_PER_SIMD
#define FMQ_WAVES_PER_SIMD 4
#endif

__device__ __forceinline__ unsigned sortable_bits(float v) {   // ascending float order as ascending unsigned order; NaN last
    if (v != v) return 0xffffffffu;
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// raw moments of the rows, per workgroup, in double; fixed order (deterministic basis -> deterministic packing)
constexpr int FX_MOM_BLOCK = 640;
constexpr int FX_MOM_TILE = 32;
__global__ __launch_bounds__(FX_MOM_BLOCK)
void k_fm_moments(const float* __restrict__ f, int n, int rows_per_block, double* __restrict__ partial) {
    __shared__ float tile[FX_MOM_TILE][FD + 1];
    const int t = threadIdx.x;
    int a = 0, b = 0;   // thread t < 561: pair (a <= b); 561 <= t < 594: column sum
    if (t < 561) { int r = t; a = 0; while (r >= FD - a) { r -= FD - a; ++a; } b = a + r; }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    double acc = 0.0;
    for (int base = r0; base < r1; base += FX_MOM_TILE) {
        const int m = min(FX_MOM_TILE, r1 - base);
        for (int e = t; e < m * FD; e += FX_MOM_BLOCK) tile[e / FD][e % FD] = f[(size_t)base * FD + e];
        __syncthreads();
        if (t < 561) { for (int r = 0; r < m; ++r) acc += (double)tile[r][a] * (double)tile[r][b]; }
        else if (t < FX_NMOM) { for (int r = 0; r < m; ++r) acc += (double)tile[r][t - 561]; }
        __syncthreads();
    }
    if (t < FX_NMOM) partial[(size_t)blockIdx.x * FX_NMOM + t] = acc;
}
// one wave per moment: the workgroups' partial sums in a fixed order (lane l takes blocks l, l + 64, ...; then a fixed tree)
__global__ __launch_bounds__(64)
void k_fm_moments_fold(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
    const int t = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[(size_t)b * FX_NMOM + t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) out[t] = s;
}

// Principal coordinates p_r(x) = sum_d (x_d - mean_d) * b_r[d], r = 0..2, as evaluated HERE (plain f32, d ascending):
// the one routine both sides use.  *amax receives (integer atomic max on the bits of a non-negative float) the largest
// |x_d - mean_d| seen, +inf for non-finite input: it scales the rounding margin of the principal-direction boxes.
// basis: [3][33] directions, then mean[33]
__device__ __forceinline__ void principal_coords(const float* __restrict__ x, const float* __restrict__ basis, float& a0, float& a1, float& a2, float& am) {
    a0 = 0.f; a1 = 0.f; a2 = 0.f; am = 0.f;
#pragma unroll
    for (int d = 0; d < FD; ++d) {
        const float v = x[d] - basis[3 * FD + d];
        a0 += v * basis[d]; a1 += v * basis[FD + d]; a2 += v * basis[2 * FD + d];
        const float av = fabsf(v);
        am = (av <= am) ? am : av;          // NaN: the comparison is false -> am = NaN, mapped to +inf below
    }
    if (!(am <= FLT_MAX)) am = INFINITY;
}
__global__ void k_fm_project(const float* __restrict__ f, int n, const float* __restrict__ basis, float* __restrict__ p0,
                             float* __restrict__ p1, float* __restrict__ p2, unsigned* __restrict__ amax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, am = 0.f;
    if (i < n) { principal_coords(f + (size_t)i * FD, basis, a0, a1, a2, am); p0[i] = a0; p1[i] = a1; p2[i] = a2; }
    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_max, __float_as_uint(am));
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, s_max);
}

// Bit-identical target rows (the descriptor of a flat patch: a quarter of the relief model's rows are two such values)
// give bit-identical distances, and the tie rule hands the match to the lowest index among them: only that row can ever
// win, so the packed index holds it alone.  Without this every source on such a plateau has to open every leaf holding
// a copy.  Two levels of open addressing keyed by a hash of the row's bits, rows always compared in full (a hash
// collision costs a probe, never a row): a workgroup first folds its own 512 rows in LDS - a popular value would
// otherwise queue tens of thousands of atomics on one L2 address - and only the lowest row of every value it holds
// goes to the global table.  After the kernel table[slot_of[i]] == i exactly for the lowest row of every distinct value.
constexpr int FX_DD_ROWS = 512;
constexpr int FX_DD_SLOTS = 1024;
__device__ __forceinline__ unsigned row_hash(const float* x) {
    unsigned h = 0x9e3779b9u;
#pragma unroll
    for (int d = 0; d < FD; ++d) { h ^= __float_as_uint(x[d]); h *= 0x85ebca6bu; h ^= h >> 13; }
    h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__global__ __launch_bounds__(FX_DD_ROWS)
void k_fm_dedupe_insert(const float* __restrict__ f, int n, int* table, unsigned mask, int* __restrict__ slot_of, int* __restrict__ kept) {
    __shared__ float tile[FX_DD_ROWS * FD];          // row-major, stride 33 dwords: lanes = consecutive rows hit distinct banks
    __shared__ int ltab[FX_DD_SLOTS], lres[FX_DD_SLOTS];
    __shared__ int claimed;
    const int t = threadIdx.x, base = blockIdx.x * FX_DD_ROWS, m = min(FX_DD_ROWS, n - base);
    for (int e = t; e < m * FD; e += FX_DD_ROWS) tile[e] = f[(size_t)base * FD + e];
    for (int e = t; e < FX_DD_SLOTS; e += FX_DD_ROWS) ltab[e] = -1;
    if (t == 0) claimed = 0;
    __syncthreads();
    const float* x = tile + t * FD;
    unsigned h = 0, ls = 0;
    if (t < m) {
        h = row_hash(x);
        ls = h & (FX_DD_SLOTS - 1);
        for (;;) {
            int cur = atomicCAS(&ltab[ls], -1, t);
            if (cur < 0) break;
            const float* y = tile + cur * FD;
            bool same = true;
#pragma unroll
            for (int d = 0; d < FD; ++d) same = same && (__float_as_uint(x[d]) == __float_as_uint(y[d]));
            if (same) { if (t < cur) atomicMin(&ltab[ls], t); break; }
            ls = (ls + 1) & (FX_DD_SLOTS - 1);       // half full at most: the probe ends
        }
    }
    __syncthreads();
    bool claim = false;
    if (t < m && ltab[ls] == t) {                    // lowest row of its value in this workgroup
        const int i = base + t;
        unsigned slot = h & mask;
        for (;;) {
            // plain load: a stale owner is still a row with the slot's value, a stale "empty" is corrected by the CAS
            int cur = table[slot];
            if (cur < 0) { cur = atomicCAS(&table[slot], -1, i); if (cur < 0) { claim = true; break; } }
            const float* y = f + (size_t)cur * FD;
            bool same = true;
#pragma unroll
            for (int d = 0; d < FD; ++d) same = same && (__float_as_uint(x[d]) == __float_as_uint(y[d]));
            if (same) { if (i < cur) atomicMin(&table[slot], i); break; }
            slot = (slot + 1) & mask;
        }
        lres[ls] = (int)slot;
    }
    // every distinct value claims exactly one empty slot of the global table: the claims count the rows that stay
    const unsigned long long cm = __ballot(claim);
    if ((t & 63) == 0 && cm) atomicAdd(&claimed, __popcll(cm));
    __syncthreads();
    if (t < m) slot_of[base + t] = lres[ls];
    if (t == 0 && claimed) atomicAdd(kept, claimed);
}

// first key: p0 of the rows that stay; the others sort behind every real row together with the padding
// Keys of the three sorts of the packing.  The first two are stable radix sorts of (key, row) pairs: rows enter in index order,
// so equal keys keep the lower row first.
__global__ void k_fm_key_p0(const float* __restrict__ p0, int n, const int* __restrict__ table, const int* __restrict__ slot_of,
                            unsigned long long* __restrict__ key, unsigned* __restrict__ row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool keep = table[slot_of[i]] == i;             // a bit-identical copy of an earlier row stays out of the index: it sorts last
    key[i] = keep ? (unsigned long long)sortable_bits(p0[i]) : (1ull << 32);
    row[i] = (unsigned)i;
}
// number of entries of the ascending array `starts` (m + 1 entries, starts[0] = 0) that are <= r, minus 1
__device__ __forceinline__ int segment_of(const int* __restrict__ starts, int m, int r) {
    int lo = 0, hi = m;   // invariant: starts[lo] <= r < starts[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= r) lo = mid; else hi = mid; }
    return lo;
}
// after the sort along p0: rank -> slab (equal counts); next key = (slab, p1); slab boundary values for locating
__global__ void k_fm_key_p1(const unsigned* __restrict__ row_in, int n, const int* __restrict__ slab_start, int S0, const float* __restrict__ p0,
                            const float* __restrict__ p1, float* __restrict__ b0, unsigned long long* __restrict__ key) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const unsigned idx = row_in[r];
    const int k = segment_of(slab_start, S0, r);
    if (r == slab_start[k]) b0[k] = p0[idx];
    key[r] = ((unsigned long long)(unsigned)k << 32) | sortable_bits(p1[idx]);
}
// after the sort along (slab, p1): rank -> column; last key = (column, p2, row), as 16-byte records for the per-column sort
__global__ void k_fm_rec_p2(const unsigned* __restrict__ row_in, int n, const int* __restrict__ col_start, int ncol, const float* __restrict__ p1,
                            const float* __restrict__ p2, float* __restrict__ b1, uint4* __restrict__ rec) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const unsigned idx = row_in[r];
    const int c = segment_of(col_start, ncol, r);
    if (r == col_start[c]) b1[c] = p1[idx];
    rec[r] = make_uint4((unsigned)c, sortable_bits(p2[idx]), idx, 0u);
}
__global__ void k_fm_fill_rows(float* __restrict__ T, int* __restrict__ torig, size_t n_rows) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_rows * FD) T[e] = INFINITY;
    if (e < n_rows) torig[e] = INT_MAX;
}
// Row r of the packed table lives in leaf r / 64 as column r % 64 of a [33][64] block: a wave reads one dimension of a
// whole leaf with one coalesced 256-B load (lane = row) and hands rows to its lanes' arithmetic with v_readlane.
__device__ __forceinline__ size_t row_elem(size_t row, int d) { return (row / FX_LEAF) * (size_t)(FD * FX_LEAF) + (size_t)d * FX_LEAF + row % FX_LEAF; }
// after the sort along (column, p2, row): rows to their padded positions
__global__ void k_fm_place_rows(const uint4* __restrict__ rec, int n, const int* __restrict__ col_start, const int* __restrict__ col_row0, int ncol,
                                const float* __restrict__ ft, const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ p2,
                                float* __restrict__ T, int* __restrict__ torig, float* __restrict__ leaf_p2, float* __restrict__ prow /* [3][rows] */, size_t rows) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n * FD) return;
    const int r = (int)(e / FD), d = (int)(e % FD);
    const unsigned idx = rec[r].z;
    const int c = segment_of(col_start, ncol, r);
    const size_t row = (size_t)col_row0[c] + (size_t)(r - col_start[c]);
    T[row_elem(row, d)] = ft[(size_t)idx * FD + d];
    if (d == 0) {
        torig[row] = (int)idx;
        if (row % FX_LEAF == 0) leaf_p2[row / FX_LEAF] = p2[idx];
        prow[row] = p0[idx]; prow[rows + row] = p1[idx]; prow[2 * rows + row] = p2[idx];
    }
}
// Leaf boxes over the real rows of 64 padded rows; a leaf of padding only gets the empty box (+inf, -inf): its bound is
// +inf.  Stored per group, transposed: lbox[group][min | max][33][64 leaves], so a wave reads one dimension of a group's
// 64 boxes with one coalesced load (lane = leaf); leaves past the end are empty.  Beside the 33-D box every leaf has a
// 3-D box of its rows' principal coordinates, pbox[group][min | max][3][64 leaves]: a leaf IS a cell of the packing in
// those coordinates, so this box is tight where the 33-D box (axis-aligned, the data are not) is loose; together they
// open 8 leaves per source where the 33-D box alone opens 19 (tools/studies/feature_match_tail.py).
constexpr int PD = 3;
__global__ void k_fm_leaf_boxes(const float* __restrict__ T, const int* __restrict__ torig, const float* __restrict__ prow, size_t rows,
                                int nleaf, int ngroup, float* __restrict__ lbox, float* __restrict__ pbox) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ngroup * FX_GROUP * (FD + PD)) return;
    const int b = e / (FD + PD), d = e % (FD + PD);
    float mn = INFINITY, mx = -INFINITY;
    if (b < nleaf)
        for (int r = b * FX_LEAF; r < (b + 1) * FX_LEAF; ++r) {
            if (torig[r] == INT_MAX) continue;
            const float v = d < FD ? T[row_elem((size_t)r, d)] : prow[(size_t)(d - FD) * rows + r];
            mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
    if (d < FD) {
        float* gb = lbox + (size_t)(b / FX_GROUP) * (2 * FD * FX_GROUP);
        gb[d * FX_GROUP + b % FX_GROUP] = mn; gb[(FD + d) * FX_GROUP + b % FX_GROUP] = mx;
    } else {
        float* gb = pbox + (size_t)(b / FX_GROUP) * (2 * PD * FX_GROUP);
        gb[(d - FD) * FX_GROUP + b % FX_GROUP] = mn; gb[(PD + d - FD) * FX_GROUP + b % FX_GROUP] = mx;
    }
}
// group boxes, same transposed layouts one level up: gbox[chunk of 64 groups][min | max][33][64], gpbox[chunk][min | max][3][64]
__global__ void k_fm_group_boxes(const float* __restrict__ lbox, const float* __restrict__ pbox, int ngroup, int nchunk,
                                 float* __restrict__ gbox, float* __restrict__ gpbox) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nchunk * 64 * (FD + PD)) return;
    const int g = e / (FD + PD), d = e % (FD + PD);
    float mn = INFINITY, mx = -INFINITY;
    if (g < ngroup) {
        const float* gb = d < FD ? lbox + (size_t)g * (2 * FD * FX_GROUP) : pbox + (size_t)g * (2 * PD * FX_GROUP);
        const int dd = d < FD ? d : d - FD, nd = d < FD ? FD : PD;
        for (int l = 0; l < FX_GROUP; ++l) { mn = fminf(mn, gb[dd * FX_GROUP + l]); mx = fmaxf(mx, gb[(nd + dd) * FX_GROUP + l]); }
    }
    if (d < FD) {
        float* cb = gbox + (size_t)(g / 64) * (2 * FD * 64);
        cb[d * 64 + g % 64] = mn; cb[(FD + d) * 64 + g % 64] = mx;
    } else {
        float* cb = gpbox + (size_t)(g / 64) * (2 * PD * 64);
        cb[(d - FD) * 64 + g % 64] = mn; cb[(PD + d - FD) * 64 + g % 64] = mx;
    }
}

// home leaf of every source: its cell of the target packing (slab by p0, column by p1, leaf by p2)
__global__ void k_fm_locate(const float* __restrict__ fs, int ns, const float* __restrict__ basis, int S0, int S1,
                            const float* __restrict__ b0, const float* __restrict__ b1, const int* __restrict__ col_leaf0,
                            const float* __restrict__ leaf_p2, int bucket_shift, int* __restrict__ home, int* __restrict__ bucket_of,
                            float* __restrict__ sp /* [ns][4]: p0 p1 p2 - */, unsigned* __restrict__ amax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, am = 0.f;
    if (i < ns) principal_coords(fs + (size_t)i * FD, basis, a0, a1, a2, am);
    {   // one atomic per workgroup (per wave they queue on one address for longer than the kernel's own work takes)
        __shared__ unsigned s_max;
        if (threadIdx.x == 0) s_max = 0u;
        __syncthreads();
        float wm = am;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&s_max, __float_as_uint(wm));
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(amax, s_max);
    }
    if (i >= ns) return;
    *reinterpret_cast<float4*>(sp + (size_t)i * 4) = make_float4(a0, a1, a2, 0.f);
    int k = 0;
    for (int s = 1; s < S0; ++s) k += (b0[s] <= a0) ? 1 : 0;           // boundaries ascend; NaN compares false -> cell 0
    int j = 0;
    for (int s = 1; s < S1; ++s) j += (b1[k * S1 + s] <= a1) ? 1 : 0;
    const int c = k * S1 + j;
    const int l0 = col_leaf0[c], l1 = col_leaf0[c + 1];
    int l = l0;
    for (int s = l0 + 1; s < l1; ++s) l += (leaf_p2[s] <= a2) ? 1 : 0;
    l = min(l, max(l1 - 1, l0));
    home[i] = l;
    bucket_of[i] = l >> bucket_shift;
}
__global__ __launch_bounds__(FMP_SORT_BLOCK)
void k_fm_bucket_hist(const int* __restrict__ bucket_of, int n, int* __restrict__ hist) {
    __shared__ int h[FMP_BUCKETS];
    for (int b = threadIdx.x; b < FMP_BUCKETS; b += FMP_SORT_BLOCK) h[b] = 0;
    __syncthreads();
    const int i = blockIdx.x * FMP_SORT_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&h[bucket_of[i]], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < FMP_BUCKETS; b += FMP_SORT_BLOCK) if (h[b]) atomicAdd(&hist[b], h[b]);
}

__device__ __forceinline__ float box_bound(const float (&f)[FD], const float* __restrict__ lo, const float* __restrict__ hi) {
    float lb = 0.f;
#pragma unroll
    for (int d = 0; d < FD; ++d) { const float g = fmaxf(fmaxf(lo[d] - f[d], f[d] - hi[d]), 0.f); lb += g * g; }
    return lb;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// the descriptors of the K sources of every wave, interleaved: fsk[wave][d][k] = fs[sperm[K wave + k]][d] (a wave past the
// end of an uneven count repeats the last source, as FmWave does): (q_k[d], q_k+1[d]) is then one aligned SGPR pair
__global__ void k_fm_interleave_rows(const float* __restrict__ fs, const int* __restrict__ sperm, int ns, int K, float* __restrict__ fsk) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nwave = ((size_t)ns + K - 1) / K;
    if (e >= nwave * K * FD) return;
    const size_t w = e / ((size_t)K * FD); const int r = (int)(e % ((size_t)K * FD)), d = r / K, k = r % K;
    fsk[e] = fs[(size_t)sperm[min((int)(K * w) + k, ns - 1)] * FD + d];
}

// ---- wave-level helpers (DPP: no LDS traffic) ----------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// minimum over the 64 lanes, returned wave-uniform (an SGPR after readlane)
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, dpp_f32<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
    v = fminf(v, dpp_f32<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
    v = fminf(v, dpp_f32<0x141, 0xF>(v));   // row_half_mirror
    v = fminf(v, dpp_f32<0x140, 0xF>(v));   // row_mirror: every lane of a 16-lane row holds the row minimum
    v = fminf(v, dpp_f32<0x142, 0xA>(v));   // row_bcast15 into rows 1 and 3
    v = fminf(v, dpp_f32<0x143, 0xC>(v));   // row_bcast31 into rows 2 and 3: lane 63 holds the wave minimum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// K sources per wave, LANE = TARGET-SIDE ITEM (a row of a leaf, a leaf box of a group, a group box): every lane does
// useful, distinct work, and the K source descriptors are wave-uniform (scalar loads of the wave's own 132-B rows, which
// stay in the scalar cache).  The sources of a wave share a home leaf (they were ordered by it), so they need nearly the
// same leaves: a leaf fetched once (33 coalesced 256-B loads, lane = row) serves all K of them.
//   round-2 history of this kernel at 143k x 151k real descriptors (profiles/r2/history/feature_match_designs.md):
//   lane = source, rows through the scalar path 7.5 ms (every wave opens its own leaves: the scalar cache misses on all
//   of them); lane = source, rows through v_readlane 7.6 ms (175 issue slots per row, and a wave of 64 sources opens the
//   union of their leaves: 78 against 19 for a single source); lane = row with K = 8 / 4 / 2 / 1 sources per wave:
//   12.8 (register spills) / 5.8 / 1.6 / 2.0 ms; four waves sharing one source pair through LDS bounds: 3.1 ms (the
//   table is then read 36 GB instead of 28 GB per call — the kernel is bound by L2 / Infinity-Cache bandwidth).
// Per source and lane a running (distance, original index) minimum over the rows that lane has seen; its wave minimum is
// the source's bound.  A box is opened when its bound <= the source's bound (<=: an equal distance with a lower index
// could still win; the exact lowest-index rule is applied by the final lexicographic reduction).
// Order: home leaf, home group, then every group whose box passes (tested once, lane = group, with the bounds the home
// group left), inside-out from the home group; inside a group the leaves whose boxes pass (lane = leaf), inside-out.
// Bounds only shrink, so a mask computed earlier can open a leaf too many, never skip one.
struct FmTables {   // device pointers of a packed index + the per-call source-side arrays (plain struct: passed by value)
    const float* fs; const int* sperm; const int* home_of; int ns;
    const float* fs2;   // even K: the descriptors of every wave's K sources interleaved, [wave][33][K] (k_fm_interleave_rows); else null
    const float* T; const int* torig; int nleaf, ngroup;
    const float *lbox, *gbox, *pbox, *gpbox;
    const float* sp; const unsigned *amax_t, *amax_s; float pscale;
};

// The search state of one wave: K sources, lane = target-side item.  SHARED: the bounds are also kept in LDS words
// (integer atomic min on the bits of a non-negative float) so that several waves working on the same sources tighten
// each other's bounds.
template <int K, bool SHARED>
struct FmWave {
    const FmTables& t;
    const int lane;
    int src[K];
    const float* q2 = nullptr;    // even K: this wave's interleaved sources, [33][K] (wave-uniform)
    unsigned long long lkey[K];   // this lane's best (distance bits : original index): unsigned order == (distance, index) order for distances >= 0
    float bound[K];
    float pmargin;
    int home, hg;
    int* s_bound;                 // SHARED only
    unsigned n_open = 0, n_leaf_tests = 0, n_group_tests = 0;

    __device__ __forceinline__ FmWave(const FmTables& tt, int s0, int* sb) : t(tt), lane(threadIdx.x & 63), s_bound(sb) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            src[k] = __builtin_amdgcn_readfirstlane(t.sperm[min(s0 + k, t.ns - 1)]);   // past the end: the last source again
            lkey[k] = (unsigned long long)__float_as_uint(FLT_MAX) << 32; bound[k] = FLT_MAX;   // registration.cpp:218-219: only dist < FLT_MAX is ever taken
        }
        if (K % 2 == 0) q2 = t.fs2 + (size_t)(s0 / K) * (K * FD);
        home = min(t.nleaf - 1, max(0, __builtin_amdgcn_readfirstlane(t.home_of[src[K / 2]])));
        hg = home / FX_GROUP;
        // rounding margin of a principal-coordinate gap (principal_bound_note): 3e-5 * largest |x_d - mean_d| on either side
        pmargin = 3e-5f * fmaxf(__uint_as_float(__builtin_amdgcn_readfirstlane(*t.amax_t)), __uint_as_float(__builtin_amdgcn_readfirstlane(*t.amax_s)));
    }
    __device__ __forceinline__ void refresh() {
        if (SHARED) {
#pragma unroll
            for (int k = 0; k < K; ++k) bound[k] = fminf(bound[k], __int_as_float(__builtin_amdgcn_readfirstlane(s_bound[k])));
        }
    }
    // a lane's row of a leaf: 33 floats kept as 17 aligned register pairs, so that the packed arithmetic below can name
    // either half of a pair (op_sel) instead of the compiler giving every element a pair of its own
    struct RowBuf { v2f p[(FD + 1) / 2]; __device__ __forceinline__ float at(int d) const { return (d & 1) ? p[d >> 1].y : p[d >> 1].x; } };
    __device__ __forceinline__ void load_leaf(int leaf, RowBuf& row, int& ro) const {
        const float* __restrict__ blk = t.T + (size_t)leaf * (FD * FX_LEAF);
#pragma unroll
        for (int d = 0; d < FD; ++d) { if (d & 1) row.p[d >> 1].y = blk[d * FX_LEAF + lane]; else row.p[d >> 1].x = blk[d * FX_LEAF + lane]; }
        row.p[FD >> 1].y = 0.f;
        ro = t.torig[(size_t)leaf * FX_LEAF + lane];
    }
    __device__ __forceinline__ void eval_leaf(const RowBuf& row, int ro) {
        if constexpr (K % 2 == 0) {
            // two sources at once: every op one v_pk_*_f32 on (source k, source k + 1) - the same IEEE operations per
            // element, half the instructions; the wave's descriptors sit interleaved in memory so that (q_k[d], q_k+1[d]) is
            // one aligned SGPR pair
            const float* __restrict__ qk = q2;   // (left to the compiler, the K x 33 values stay in SGPRs across leaves, a few of them
                                                 // parked in VGPR lanes; re-reading them per leaf through the scalar cache doubled the time)
            v2f dist[K / 2];
#pragma unroll
            for (int j = 0; j < K / 2; ++j) dist[j] = (v2f){0.f, 0.f};
#pragma unroll
            for (int d = 0; d < FD; ++d) {
                const v2f pr = row.p[d >> 1];
                const v2f rw = (d & 1) ? __builtin_shufflevector(pr, pr, 1, 1) : __builtin_shufflevector(pr, pr, 0, 0);
#pragma unroll
                for (int j = 0; j < K / 2; ++j) {
                    const v2f q = {qk[K * d + 2 * j], qk[K * d + 2 * j + 1]};
                    const v2f diff = q - rw;          // registration.cpp:222-224
                    dist[j] += diff * diff;
                }
            }
#pragma unroll
            for (int j = 0; j < K / 2; ++j) {
                const unsigned long long k0 = ((unsigned long long)__float_as_uint(dist[j].x) << 32) | (unsigned)ro;
                const unsigned long long k1 = ((unsigned long long)__float_as_uint(dist[j].y) << 32) | (unsigned)ro;
                lkey[2 * j] = k0 < lkey[2 * j] ? k0 : lkey[2 * j];
                lkey[(2 * j + 1) % K] = k1 < lkey[(2 * j + 1) % K] ? k1 : lkey[(2 * j + 1) % K];
            }
            ++n_open;
            return;
        }
#pragma unroll
        for (int k = 0; k < K && K % 2 != 0; ++k) {
            const float* __restrict__ q = t.fs + (size_t)src[k] * FD;   // wave-uniform -> scalar loads
            float dist = 0.f;
#pragma unroll
            for (int d = 0; d < FD; ++d) { const float diff = q[d] - row.at(d); dist += diff * diff; }   // registration.cpp:222-224
            // strict < on (distance, index): the lowest index among equal distances; NaN and +inf have larger bit patterns than
            // FLT_MAX and are never taken, like `dist < best_dist` in the reference
            const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)ro;
            lkey[k] = key < lkey[k] ? key : lkey[k];
        }
        ++n_open;
    }
    // the sources' bounds = wave minimum of the lanes' best distances; needed only where boxes are tested
    __device__ __forceinline__ void update_bounds() {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float m = wave_min_f32(__uint_as_float((unsigned)(lkey[k] >> 32)));
            if (m < bound[k]) {
                bound[k] = m;
                if (SHARED && lane == 0) atomicMin(&s_bound[k], __float_as_int(m));
            }
        }
    }
    __device__ __forceinline__ void open_leaf(int leaf) {
        RowBuf row; int ro;
        load_leaf(leaf, row, ro);
        eval_leaf(row, ro);
    }
    // lanes = the 64 boxes of one block; bit b of the result: some source may still find a better row in box b.
    // principal_bound_note — the 3-D box is tested first (6 loads), the 33-D box (66 loads) only if it leaves anything.
    // Why the 3-D bound is safe although a projection is not monotone in float arithmetic: for orthonormal directions
    // sum_r (p_r(q) - p_r(t))^2 <= |q - t|^2 in real numbers.  (i) The f32 directions are orthonormal to 1.2e-7 (checked on
    // the host, else pscale = 0 disables the test).  (ii) A computed coordinate (33 sequential mul/add, no FMA) is within
    // 34 u * sqrt(33) * M = 1.2e-5 M of its real value, M = largest |x_d - mean_d|: a computed gap to the box exceeds the
    // real gap to any row by at most 2.5e-5 M; pmargin = 3e-5 M is subtracted.  (iii) fl(dist) >= |q - t|^2 (1 - 36 u).
    // pscale = 1 - 1e-4 covers (i), (iii) and the rounding of the three squares with a factor 30 to spare.  Non-finite
    // input makes M = +inf: gaps clamp to 0 and only the 33-D test decides.
    __device__ __forceinline__ unsigned long long box_mask(const float* __restrict__ blk, const float* __restrict__ pblk, bool own_bounds = true) {
        if (own_bounds) update_bounds();      // (pass B tests the group boxes with bounds every wave shares: see there)
        float lbp[K]; bool pa[K];
#pragma unroll
        for (int k = 0; k < K; ++k) lbp[k] = 0.f;
#pragma unroll
        for (int r = 0; r < PD; ++r) {
            const float lo = pblk[r * 64 + lane], hi = pblk[(PD + r) * 64 + lane];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float pq = t.sp[(size_t)src[k] * 4 + r];   // wave-uniform -> scalar load
                const float g = fmaxf(fmaxf(lo - pq, pq - hi) - pmargin, 0.f);
                lbp[k] += g * g;
            }
        }
        unsigned long long any = 0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) { pa[k] = !(lbp[k] * t.pscale > bound[k]); any |= __ballot(pa[k]); }
        if (!any) return 0ull;
        float lb[K];
#pragma unroll
        for (int k = 0; k < K; ++k) lb[k] = 0.f;
        if constexpr (K % 2 == 0) {       // packed over pairs of sources, as in eval_leaf
            const float* __restrict__ qk = q2;
            v2f lb2[K / 2];
#pragma unroll
            for (int j = 0; j < K / 2; ++j) lb2[j] = (v2f){0.f, 0.f};
#pragma unroll 11
            for (int d = 0; d < FD; ++d) {
                const float lo = blk[d * 64 + lane], hi = blk[(FD + d) * 64 + lane];
#pragma unroll
                for (int j = 0; j < K / 2; ++j) {
                    const v2f q = {qk[K * d + 2 * j], qk[K * d + 2 * j + 1]};
                    const v2f a = (v2f){lo, lo} - q, b = q - (v2f){hi, hi};
                    const v2f g = {fmaxf(fmaxf(a.x, b.x), 0.f), fmaxf(fmaxf(a.y, b.y), 0.f)};
                    lb2[j] += g * g;
                }
            }
#pragma unroll
            for (int j = 0; j < K / 2; ++j) { lb[2 * j] = lb2[j].x; lb[(2 * j + 1) % K] = lb2[j].y; }
        } else
#pragma unroll 11   // 22 loads in flight; full unrolling hoists all 66 and spills
        for (int d = 0; d < FD && K % 2 != 0; ++d) {   // one dimension of the 64 boxes at a time: two coalesced loads, K bounds advance
            const float lo = blk[d * 64 + lane], hi = blk[(FD + d) * 64 + lane];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float qd = t.fs[(size_t)src[k] * FD + d];   // wave-uniform -> scalar load
                const float g = fmaxf(fmaxf(lo - qd, qd - hi), 0.f);
                lb[k] += g * g;
            }
        }
        unsigned long long m = 0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) m |= __ballot(pa[k] && lb[k] <= bound[k]);    // an empty box (+inf, -inf) has lb = +inf: never set
        return m;
    }
    __device__ __forceinline__ unsigned long long leaf_mask(int g) {
        ++n_leaf_tests;
        unsigned long long m = box_mask(t.lbox + (size_t)g * (2 * FD * FX_GROUP), t.pbox + (size_t)g * (2 * PD * FX_GROUP));
        if (g == hg) m &= ~(1ull << (home - g * FX_GROUP));
        return m;
    }
    __device__ __forceinline__ unsigned long long group_mask(int c, bool own_bounds = true) {
        ++n_group_tests;
        unsigned long long m = box_mask(t.gbox + (size_t)c * (2 * FD * 64), t.gpbox + (size_t)c * (2 * PD * 64), own_bounds);
        if (hg >= 0 && hg / 64 == c) m &= ~(1ull << (hg % 64));
        return m;
    }
    // The leaves of group g whose boxes pass, inside-out from the home side.  The mask is known before the first leaf is
    // opened, so the next leaf's 34 loads are issued before the current leaf is evaluated (two register buffers): a wave
    // that opens many leaves no longer pays a full memory round trip for each.
    // Returns false when the leaf budget ran out before the group was finished (pass A: the caller gives the sources up).
    __device__ __forceinline__ bool visit_group(int g, unsigned budget = 0xffffffffu) {
        const int l0 = g * FX_GROUP, cnt = min(FX_GROUP, t.nleaf - l0);
        if (cnt <= 0) return true;        // a group past the end: its empty box "passes" for a NaN query (every gap is NaN -> 0)
        unsigned long long m = leaf_mask(g);
        const int centre = (g == hg) ? home - l0 : (g < hg ? cnt - 1 : 0);   // enter a neighbouring group from the home side
        if (cnt < 64) m &= (1ull << cnt) - 1ull;
        auto next = [&]() -> int {
            const int l = nearest_set_bit(m, centre);
            if (l < 0) return -1;
            m &= ~(1ull << l);
            return l0 + l;
        };
        int cur = next();
        if (cur < 0) return true;
        RowBuf a, b; int roa, rob = 0;
        load_leaf(cur, a, roa);
        for (;;) {
            if (n_open >= budget) return false;
            const int n1 = next();
            if (n1 >= 0) load_leaf(n1, b, rob);
            eval_leaf(a, roa);
            if (n1 < 0) break;
            if (n_open >= budget) return false;
            const int n2 = next();
            if (n2 >= 0) load_leaf(n2, a, roa);
            eval_leaf(b, rob);
            if (n2 < 0) break;
        }
        return true;
    }
    // lowest (distance, original index) of source k over the lanes
    __device__ __forceinline__ void result(int k, float& bd, int& bo) const {
        unsigned hi = (unsigned)(lkey[k] >> 32), lo = (unsigned)lkey[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned oh = __shfl_xor(hi, off, 64), ol = __shfl_xor(lo, off, 64);
            const bool tk = oh < hi || (oh == hi && ol < lo);
            hi = tk ? oh : hi; lo = tk ? ol : lo;
        }
        bd = __uint_as_float(hi);
        bo = hi == __float_as_uint(FLT_MAX) ? INT_MAX : (int)lo;      // nothing below FLT_MAX was seen
    }
};

// Pass A: one wave per K sources.  Home leaf, home group, then every group whose box passes (tested once per chunk of 64
// groups, lane = group, with the bounds the home group left), inside-out from the home group.  A wave whose sources
// turn out to be outliers (far from every target: most boxes pass) stops after `leaf_limit` leaves, stores what it has
// (part_d / part_j) and puts its sources on the overflow list: pass B spreads them over 8 waves each.  Without that the
// call waits for a few waves that open hundreds of leaves one after the other (measured: 580 leaves, 1.3 of 1.5 ms).
template <int K, bool STATS>
__global__ __launch_bounds__(FM_BLOCK, FMQ_WAVES_PER_SIMD)
void k_fm_query(FmTables t, int blocks_per_xcd, int leaf_limit, int heavy_groups, int* __restrict__ overflow_count /* [2]: pass B, scan */, int* __restrict__ overflow_list,
                int* __restrict__ overflow_src, float* __restrict__ part_d, int* __restrict__ part_j, int* __restrict__ corr, unsigned long long* __restrict__ stats) {
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one).  Give every XCD a CONTIGUOUS stretch of
    // the home-ordered sources: neighbouring waves open the same leaves, so the stretch's leaves (1/8 of the table) stay
    // in that XCD's 4 MB L2 instead of every L2 seeing the whole table.
    const int block = (blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);
    const int wid = block * (FM_BLOCK / 64) + (threadIdx.x >> 6);
    const int s0 = __builtin_amdgcn_readfirstlane(wid * K);
    if (s0 >= t.ns) return;
    unsigned long long t_start = 0;
    if (STATS) t_start = wall_clock64();
    FmWave<K, false> w(t, s0, nullptr);
    w.open_leaf(w.home);
    bool overflow = !w.visit_group(w.hg, (unsigned)leaf_limit);
    const int nchunk = (t.ngroup + 63) / 64;
    int groups_left = overflow ? t.ngroup : 0;       // estimate of what is left when the wave gives up
    for (int c = 0; c < nchunk && !overflow; ++c) {     // (chunks in index order; within a chunk inside-out from the home group's side)
        unsigned long long m = w.group_mask(c);
        const int g0 = c * 64, cnt = min(64, t.ngroup - g0);
        const int centre = w.hg < g0 ? 0 : (w.hg >= g0 + cnt ? cnt - 1 : w.hg - g0);
        if (cnt < 64) m &= (1ull << cnt) - 1ull;
        while (m) {
            const int g = nearest_set_bit(m, centre);
            if ((int)w.n_open >= leaf_limit || !w.visit_group(g0 + g, (unsigned)leaf_limit)) {
                overflow = true;
                groups_left = __popcll(m) + 64 * (nchunk - 1 - c);
                break;
            }
            m &= ~(1ull << g);
        }
    }
    // Sources given up with few groups left go to pass B (8 waves each on the index); with many groups left nearly every
    // box passes (an outlier, or a plateau of near-identical rows) and the plain scan is the efficient way to finish them.
    const bool to_scan = overflow && groups_left >= heavy_groups;
    int slot = 0;
    if (overflow && w.lane == 0) slot = atomicAdd(overflow_count + (to_scan ? 1 : 0), 1);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float bd; int bo;
        w.result(k, bd, bo);
        if (w.lane == 0 && s0 + k < t.ns) {
            if (overflow) { part_d[w.src[k]] = bd; part_j[w.src[k]] = bo; }
            else corr[w.src[k]] = bo == INT_MAX ? 0 : bo;   // nothing finite -> the reference keeps index 0
        }
    }
    if (overflow && w.lane == 0) {
        if (!to_scan) overflow_list[slot] = s0;
        else {
            overflow_list[t.ns + 1 + slot] = s0;     // second half of the list: the scan class, as wave starts ...
#pragma unroll
            for (int k = 0; k < K; ++k) overflow_src[slot * K + k] = s0 + k < t.ns ? w.src[k] : -1;   // ... and as sources
        }
    }
    if (STATS && w.lane == 0) {   // [0] waves, [1] group-chunk tests, [2] groups visited, [3] leaves opened, [4] most leaves opened by one wave
        const unsigned long long dt = wall_clock64() - t_start;   // 100 MHz ticks
        atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)w.n_group_tests); atomicAdd(&stats[2], (unsigned long long)w.n_leaf_tests);
        atomicAdd(&stats[3], (unsigned long long)w.n_open); atomicMax(&stats[4], (unsigned long long)w.n_open);
        atomicAdd(&stats[5], dt); atomicMax(&stats[6], dt);
    }
}

// Pass B: the sources pass A gave up on, one workgroup of 8 waves per K of them.  Every wave tests the group boxes
// with the same bounds (pass A's best, which already saw the home neighbourhood) and takes every 8th passing group;
// bounds found by one wave reach the others through LDS.  The result is the lexicographic minimum of pass A's partial
// answer and the 8 waves' answers, so it does not matter that pass A's groups are visited again.
constexpr int FMB_WAVES = 8;
template <int K, bool STATS>
__global__ __launch_bounds__(FMB_WAVES * 64)
void k_fm_query_overflow(FmTables t, const int* __restrict__ overflow_count, const int* __restrict__ overflow_list,
                         const float* __restrict__ part_d, const int* __restrict__ part_j, int* __restrict__ corr,
                         unsigned long long* __restrict__ stats) {
    __shared__ int s_bound[K];
    __shared__ float s_fd[FMB_WAVES][K];
    __shared__ int s_fj[FMB_WAVES][K];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = *overflow_count;
    for (int e = blockIdx.x; e < count; e += gridDim.x) {
        const int s0 = overflow_list[e];
        FmWave<K, true> w(t, s0, s_bound);
        if (threadIdx.x < K) s_bound[threadIdx.x] = __float_as_int(fminf(part_d[t.sperm[min(s0 + (int)threadIdx.x, t.ns - 1)]], FLT_MAX));
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) w.bound[k] = __int_as_float(s_bound[k]);
        w.hg = -1;                                   // no group is special here: pass A's partial answer covers what it saw
        w.home = -1;
        const int nchunk = (t.ngroup + 63) / 64;
        int turn = 0;
        for (int c = 0; c < nchunk; ++c) {
            float keep[K];
#pragma unroll
            for (int k = 0; k < K; ++k) keep[k] = w.bound[k];
#pragma unroll
            for (int k = 0; k < K; ++k) w.bound[k] = __int_as_float(__float_as_int(part_d[w.src[k]]));   // the bound every wave shares
            unsigned long long m = w.group_mask(c, false);
#pragma unroll
            for (int k = 0; k < K; ++k) w.bound[k] = keep[k];
            const int g0 = c * 64;
            if (t.ngroup - g0 < 64) m &= (1ull << (t.ngroup - g0)) - 1ull;    // groups past the end (see visit_group)
            while (m) {
                const int g = __builtin_ctzll(m);
                m &= m - 1;
                if (turn++ % FMB_WAVES != wave) continue;
                w.refresh();
                w.visit_group(g0 + g);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float bd; int bo;
            w.result(k, bd, bo);
            if (w.lane == 0) { s_fd[wave][k] = bd; s_fj[wave][k] = bo; }
        }
        __syncthreads();
        if (threadIdx.x < K && s0 + (int)threadIdx.x < t.ns) {
            const int k = threadIdx.x;
            const int i = t.sperm[s0 + k];
            float bd = part_d[i]; int bo = part_j[i];
            for (int v = 0; v < FMB_WAVES; ++v) {
                const float od = s_fd[v][k]; const int oo = s_fj[v][k];
                if (od < bd || (od == bd && oo < bo)) { bd = od; bo = oo; }
            }
            corr[i] = bo == INT_MAX ? 0 : bo;
        }
        if (STATS && w.lane == 0) { atomicAdd(&stats[8], 1ull); atomicAdd(&stats[9], (unsigned long long)w.n_open); atomicMax(&stats[10], (unsigned long long)w.n_open); }
        __syncthreads();   // the LDS words are reused by the next entry
    }
}

// ---- leaf-major search (round 3; the default) --------------------------------------------------------------------------
// k_fm_query spends 40 % of its instructions on box tests (lane = box, one or two sources per wave) and walks a source's
// leaves one after the other: 13 % of the lane-op peak, and a few sources that need hundreds of leaves decide when the call
// ends.  Here the roles are swapped - LANE = SOURCE everywhere, boxes and target rows staged in LDS and read back as
// broadcasts - and the leaves are not walked by the sources that need them but COLLECTED per leaf and evaluated with full waves:
//   k_lm_plan<false> + k_lm_eval<true>   round 0: every source against its home leaf.  The search order is sorted by home
//                   leaf, so it already is the per-leaf list; the histogram the ordering made gives the work units.
//   k_lm_boxes<3>   a workgroup = 64 sources in home-leaf order, lane = source in each of its 4 waves.  Group boxes, then the
//                   leaf boxes of every group some lane cannot exclude - the 3-D principal box first, the 33-D box where any
//                   lane passes it - with the bound round 0 left.  A box a source cannot exclude is a (leaf, source) PAIR;
//                   a wave records them as entries (leaf, first source, lane mask) and counts them per pool and leaf.
//   k_lm_plan<true> one workgroup: sources per leaf (summed over the pools), exclusive scans -> where each leaf's sources go,
//                   where each pool's share of a leaf goes, and the list of work units (a leaf x up to 64 of its sources).
//   k_lm_scatter    one workgroup per pool, write positions in LDS: the entries' sources to their leaf's stretch.
//   k_lm_eval<false> one wave per unit: the leaf (8.4 KB) staged in the wave's own LDS, 64 rows x 33 dimensions against 64
//                   sources, two rows per packed instruction: 3 x 33 x 32 v_pk_*_f32 per unit, the reference's operations
//                   in the reference's order (registration.cpp:222-224); 64 % of the nominal lane-op peak.  A lane that found
//                   a smaller (distance, index) key lowers its source's key with a 64-bit atomic min.
//   k_lm_finish     keys -> correspondences; the overflow flag straight into pinned host memory.
// A source that needs hundreds of leaves simply owns hundreds of pairs spread over as many units: there is no tail and no
// overflow pass.  Exactness as before: every row that is not evaluated lies in a box whose bound (same expression, same
// order, monotone float operations) exceeds a distance the source had already reached; bounds only shrink, so a pair
// emitted early is at worst superfluous.  The result is the minimum over 64-bit (distance bits : original index) keys,
// which does not depend on the order of the atomics.  Descriptors without structure (every box passes) overflow the entry
// pools or the pair room (32 per source): the call then falls back to k_fm_query and its scan class.
// What was measured on the way (143k x 151k relief descriptors; profiles/r3/history/feature_match_leaf_major.md): rows and boxes
// through the scalar path 0.84 ms (a leaf is 8.4 KB, the scalar cache 16 KB per CU with a slow fill path); one global cursor for
// the pairs / one global counter per leaf for the positions: 9,000 and ~20 same-address returning atomics wait for each other
// (50 us / 45 us); two rounds of box tests (home groups first) 9.7 instead of 12.1 pairs per source but 0.525 instead of
// 0.487 ms; two sources per lane in k_lm_eval (half the LDS reads) and the next group's boxes prefetched into registers: no gain.
constexpr int LM_BOX = 72;        // floats per box as this search stores them: min[33] | max[33] | pmin[3] | pmax[3]
constexpr int LM_WAVES = 4;       // waves per workgroup of the evaluation
#ifndef LM_BOX_WAVES_VALUE
#define LM_BOX_WAVES_VALUE 4
#endif
constexpr int LM_BOX_WAVES = LM_BOX_WAVES_VALUE;   // waves that share the 64 sources of a box-test workgroup (query at 143k x 151k with 1 / 2 / 4 / 8: 0.67 / 0.55 / 0.49 / 0.55-0.63 ms)
constexpr int LM_ENTRIES = 96;    // (leaf, lane mask) entries a wave buffers in LDS before it reserves room for them in its pool
constexpr int LM_PAIRS_PER_SOURCE = 32;
constexpr int LM_POOLS = 64;
constexpr unsigned long long LM_KEY_NONE = (unsigned long long)0x7f7fffffu << 32;   // FLT_MAX : 0 - only dist < FLT_MAX is ever taken (registration.cpp:218-219)

struct LmLists {
    int *pool_count, *entry_cursor, *overflow;            // zeroed per round: [LM_POOLS][nleaf] (sources per pool and leaf) [LM_POOLS] [1]
    int *leaf_count, *pool_start;                         // written by k_lm_plan: [nleaf] sources per leaf, [LM_POOLS][nleaf] where a pool's sources of a leaf go
    int4* entries; int pool_cap;                          // (leaf, first source of the workgroup, lane mask): LM_POOLS pools of pool_cap entries, every pool
                                                          // with its own cursor (one cursor for everything: 9,000 returning atomics on one address, 50 us)
    int pair_cap;                                         // room in sorted_src
    int *leaf_start, *unit_start;                         // [nleaf + 1]
    int *unit_leaf;                                       // [pair_cap / 64 + nleaf + 2]
    int *sorted_src;                                      // [pair_cap]
    unsigned long long* keys;                             // [ns]
    int unit_cap;                                         // room in unit_leaf
};
struct LmSrc { v2f p[(FD + 1) / 2]; float pq[PD]; };     // a lane's source: descriptor as 17 register pairs, principal coordinates
// The original row indices of a leaf as k_lm_eval reads them (rarely: only where a lane can improve): the "constant" address space
// tells the compiler that nothing in the kernel writes them, which lets it use the scalar path although the kernel stores keys.
typedef const __attribute__((address_space(4))) int* lm_cint_p;

__global__ void k_lm_box_layout(const float* __restrict__ lbox, const float* __restrict__ pbox, const float* __restrict__ gbox, const float* __restrict__ gpbox,
                                int nleaf, int ngroup, float* __restrict__ sleaf, float* __restrict__ sgroup) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (nleaf + ngroup) * LM_BOX) return;
    const int b = e / LM_BOX, f = e % LM_BOX;
    if (b < nleaf) {
        const int g = b / FX_GROUP, l = b % FX_GROUP;
        sleaf[e] = f < 2 * FD ? lbox[(size_t)g * (2 * FD * FX_GROUP) + f * FX_GROUP + l] : pbox[(size_t)g * (2 * PD * FX_GROUP) + (f - 2 * FD) * FX_GROUP + l];
    } else {
        const int gi = b - nleaf, c = gi / 64, k = gi % 64;
        sgroup[(size_t)gi * LM_BOX + f] = f < 2 * FD ? gbox[(size_t)c * (2 * FD * 64) + f * 64 + k] : gpbox[(size_t)c * (2 * PD * 64) + (f - 2 * FD) * 64 + k];
    }
}

__device__ __forceinline__ void lm_load_source(const FmTables& t, int src, LmSrc& q) {
    const float* __restrict__ f = t.fs + (size_t)src * FD;
#pragma unroll
    for (int d = 0; d < FD; ++d) { if (d & 1) q.p[d >> 1].y = f[d]; else q.p[d >> 1].x = f[d]; }
    q.p[FD >> 1].y = 0.f;
#pragma unroll
    for (int r = 0; r < PD; ++r) q.pq[r] = t.sp[(size_t)src * 4 + r];
}
// The scalar path cannot feed this: a leaf is 8.4 KB, the scalar cache 16 KB per CU and its fill path slow (waves of one
// CU on different leaves evict each other: 193 us for the home leaves alone).  Rows and boxes are therefore STAGED IN LDS -
// coalesced vector loads in, broadcast reads (every lane the same address) out.
typedef float v4f_lm __attribute__((ext_vector_type(4)));
// the wave's own copy of a leaf: rows[d * 64 + r]
__device__ __forceinline__ void lm_stage_leaf(const float* __restrict__ T, int leaf, float* rows, int lane) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(T + (size_t)leaf * (FD * FX_LEAF));
    float4* dst = reinterpret_cast<float4*>(rows);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < (FD * FX_LEAF / 4 + 63) / 64; ++k) { const int e = k * 64 + lane; if (e < FD * FX_LEAF / 4) dst[e] = src[e]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// One staged leaf against every lane's source; key = the lane's best (distance bits : original index) so far.
__device__ __forceinline__ void lm_eval_leaf(const float* rows, const int* __restrict__ torig, int leaf, const LmSrc& q, unsigned long long& key) {
    const lm_cint_p ro = (lm_cint_p)(torig + (size_t)leaf * FX_LEAF);
#pragma unroll 1
    for (int r0 = 0; r0 < FX_LEAF; r0 += 16) {
        v2f acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (v2f){0.f, 0.f};
#pragma unroll
        for (int d = 0; d < FD; ++d) {
            const v2f pr = q.p[d >> 1];
            const v2f qq = (d & 1) ? __builtin_shufflevector(pr, pr, 1, 1) : __builtin_shufflevector(pr, pr, 0, 0);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const v4f_lm r4 = *reinterpret_cast<const v4f_lm*>(rows + d * FX_LEAF + r0 + 4 * j4);   // 4 rows of dimension d: one broadcast ds_read_b128
                const v2f d0 = qq - (v2f){r4[0], r4[1]}, d1 = qq - (v2f){r4[2], r4[3]};              // registration.cpp:222-224
                acc[2 * j4] += d0 * d0; acc[2 * j4 + 1] += d1 * d1;
            }
        }
        // the 16 distances against the lane's best: the keys are only built where some lane can improve (or tie)
        float mn = fminf(acc[0].x, acc[0].y);
#pragma unroll
        for (int j = 1; j < 8; ++j) mn = fminf(mn, fminf(acc[j].x, acc[j].y));
        if (__any(mn <= __uint_as_float((unsigned)(key >> 32)))) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long k0 = ((unsigned long long)__float_as_uint(acc[j].x) << 32) | (unsigned)ro[r0 + 2 * j];
                const unsigned long long k1 = ((unsigned long long)__float_as_uint(acc[j].y) << 32) | (unsigned)ro[r0 + 2 * j + 1];
                key = k0 < key ? k0 : key;      // strict < on (distance, index): NaN and +inf have larger bit patterns than FLT_MAX
                key = k1 < key ? k1 : key;
            }
        }
    }
}
// box: 72 floats in LDS (min[33] | max[33] | pmin[3] | pmax[3]), the same for every lane
__device__ __forceinline__ bool lm_pass3(const float* box, const LmSrc& q, float pmargin, float pscale, float bound) {   // principal_bound_note (FmWave::box_mask)
    float lbp = 0.f;
#pragma unroll
    for (int r = 0; r < PD; ++r) {
        const float g = fmaxf(fmaxf(box[2 * FD + r] - q.pq[r], q.pq[r] - box[2 * FD + PD + r]) - pmargin, 0.f);
        lbp += g * g;
    }
    return !(lbp * pscale > bound);
}
__device__ __forceinline__ float lm_bound33(const float* box, const LmSrc& q) {
    float lb = 0.f;
#pragma unroll
    for (int d = 0; d < FD; ++d) {
        const float qd = (d & 1) ? q.p[d >> 1].y : q.p[d >> 1].x;
        const float g = fmaxf(fmaxf(box[d] - qd, qd - box[FD + d]), 0.f);
        lb += g * g;
    }
    return lb;
}

// Rounds 1 and 2: box tests.  A WORKGROUP = 64 sources in home-leaf order (lane = source in each of its LM_BOX_WAVES waves); the
// boxes of a group are staged in LDS by the whole workgroup and its waves share the leaves between them (leaf i of the group
// goes to wave i % LM_BOX_WAVES).  (First version: one wave per 64 sources, boxes through the scalar path - a chain of dependent
// scalar loads with two waves per SIMD: 267 us for what is 20 us of instructions.)
// ROUND 1: the leaves of the workgroup's home group(s); ROUND 2: every other group, group boxes first.  The four waves hold the
// same sources, so every mask that steers the staging is the same in all of them: the barriers are reached together.
template <int ROUND>
__global__ __launch_bounds__(64 * LM_BOX_WAVES)
void k_lm_boxes(FmTables t, const float* __restrict__ sleaf, const float* __restrict__ sgroup, LmLists L, unsigned long long* __restrict__ stats) {
    const unsigned long long t_begin = stats ? wall_clock64() : 0ull;
    unsigned n3 = 0, n33 = 0, nemit = 0;        // (TDV_FM_STATS: 3-D tests, 33-D tests, boxes emitted by this wave)
    __shared__ __attribute__((aligned(16))) float s_box[FX_GROUP * LM_BOX];
    __shared__ int s_leaf[LM_BOX_WAVES][LM_ENTRIES];
    __shared__ unsigned long long s_mask[LM_BOX_WAVES][LM_ENTRIES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s0 = blockIdx.x * 64;
    if (s0 >= t.ns) return;
    const bool valid = s0 + lane < t.ns;
    const int src = t.sperm[min(s0 + lane, t.ns - 1)];          // past the end: the last source again (tested, never emitted)
    LmSrc q;
    {   // the workgroup's 64 descriptors: rows in (four threads per 132-B row), through LDS, a lane's own row out
        constexpr int PITCH = FD + 4;                            // 37 floats: a lane reading its row meets no bank twice
        const int part = threadIdx.x & 3;
        for (int r = threadIdx.x >> 2; r < 64; r += (64 * LM_BOX_WAVES) >> 2) {
            const float* __restrict__ row = t.fs + (size_t)t.sperm[min(s0 + r, t.ns - 1)] * FD;
            for (int d = part; d < FD; d += 4) s_box[r * PITCH + d] = row[d];
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < FD; ++d) { if (d & 1) q.p[d >> 1].y = s_box[lane * PITCH + d]; else q.p[d >> 1].x = s_box[lane * PITCH + d]; }
        q.p[FD >> 1].y = 0.f;
        const float4 pc = *reinterpret_cast<const float4*>(t.sp + (size_t)src * 4);
        q.pq[0] = pc.x; q.pq[1] = pc.y; q.pq[2] = pc.z;
    }
    const int home = min(t.nleaf - 1, max(0, t.home_of[src])), hg = home / FX_GROUP;
    const float pmargin = 3e-5f * fmaxf(__uint_as_float(__builtin_amdgcn_readfirstlane(*t.amax_t)), __uint_as_float(__builtin_amdgcn_readfirstlane(*t.amax_s)));
    const float bound = __uint_as_float((unsigned)(L.keys[src] >> 32));
    const unsigned long long t_loaded = stats ? (bound == -1.f ? 1ull : wall_clock64()) : 0ull;   // (reads `bound`: the clock is taken after the loads have landed)
    int n_ent = 0;
    auto stage = [&](const float* __restrict__ boxes, int count) {     // count boxes -> s_box, by the whole workgroup
        __syncthreads();
        const float4* __restrict__ from = reinterpret_cast<const float4*>(boxes);
        float4* to = reinterpret_cast<float4*>(s_box);
        for (int e = threadIdx.x; e < count * (LM_BOX / 4); e += 64 * LM_BOX_WAVES) to[e] = from[e];
        __syncthreads();
    };
    auto flush = [&]() {
        if (n_ent == 0) return;
        const int pool = (blockIdx.x * LM_BOX_WAVES + wave) % LM_POOLS;
        int base = 0;
        if (lane == 0) base = atomicAdd(L.entry_cursor + pool, n_ent);
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < n_ent; e += 64)      // one instruction for all the boxes' counters, nothing waits for them; per pool: a counter per leaf alone
            atomicAdd(&L.pool_count[(size_t)pool * t.nleaf + s_leaf[wave][e]], __popcll(s_mask[wave][e]));   // queued ~20 atomics from all over the chip on one address
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + n_ent > L.pool_cap) { if (lane == 0) *L.overflow = 1; }
        else
            for (int e = lane; e < n_ent; e += 64) {
                const unsigned long long m = s_mask[wave][e];
                L.entries[(size_t)pool * L.pool_cap + base + e] = make_int4(s_leaf[wave][e], s0, (int)(unsigned)m, (int)(unsigned)(m >> 32));
            }
        __builtin_amdgcn_wave_barrier();
        n_ent = 0;
    };
    auto leaves_of_group = [&](int g) {
        const int l0 = g * FX_GROUP, cnt = min(FX_GROUP, t.nleaf - l0);
        stage(sleaf + (size_t)l0 * LM_BOX, cnt);
        for (int i = wave; i < cnt; i += LM_BOX_WAVES) {
            const int l = l0 + i;
            const float* box = s_box + i * LM_BOX;
            const bool p3 = lm_pass3(box, q, pmargin, t.pscale, bound);
            ++n3;
            if (!__any(p3)) continue;
            ++n33;
            const float lb = lm_bound33(box, q);
            const unsigned long long m = __ballot(valid && p3 && lb <= bound && l != home);   // (its home leaf: round 0 evaluated it; an empty box (+inf, -inf) has lb = +inf: never set)
            if (!m) continue;
            if (lane == 0) { s_leaf[wave][n_ent] = l; s_mask[wave][n_ent] = m; }
            ++nemit;
            if (++n_ent == LM_ENTRIES) flush();
        }
    };
    if (ROUND == 1) {
        unsigned long long todo = __ballot(valid);
        while (todo) {                                           // the workgroup's distinct home groups
            const int g = __builtin_amdgcn_readlane(hg, __builtin_ctzll(todo));
            todo &= ~__ballot(hg == g);
            leaves_of_group(g);
        }
    } else {
        for (int g0 = 0; g0 < t.ngroup; g0 += 64) {
            const int gcnt = min(64, t.ngroup - g0);
            stage(sgroup + (size_t)g0 * LM_BOX, gcnt);
            unsigned long long gm = 0ull;                        // groups some lane cannot exclude (the same in all four waves)
            for (int i = 0; i < gcnt; ++i) {
                if (ROUND == 2 && __ballot(hg == g0 + i)) continue;   // one of the home groups: round 1 tested its leaves for every lane
                const float* box = s_box + i * LM_BOX;
                const bool p3 = lm_pass3(box, q, pmargin, t.pscale, bound);
                if (!__any(p3)) continue;
                const float lb = lm_bound33(box, q);
                if (__ballot(valid && p3 && lb <= bound)) gm |= 1ull << i;
            }
            while (gm) {
                const int g = g0 + __builtin_ctzll(gm);
                gm &= gm - 1ull;
                leaves_of_group(g);
            }
        }
    }
    const unsigned long long t_f0 = stats ? wall_clock64() : 0ull;
    flush();
    if (stats && lane == 0) {      // [0] waves [1] ticks total [2] ticks until the source is loaded [3] ticks of the last flush [4] 3-D tests [5] 33-D tests [6] boxes emitted [7] max ticks
        const unsigned long long t_end = wall_clock64();
        atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], t_end - t_begin); atomicAdd(&stats[2], t_loaded - t_begin); atomicAdd(&stats[3], t_end - t_f0);
        atomicAdd(&stats[4], (unsigned long long)n3); atomicAdd(&stats[5], (unsigned long long)n33); atomicAdd(&stats[6], (unsigned long long)nemit); atomicMax(&stats[7], t_end - t_begin);
    }
}

// leaf_start / unit_start: exclusive scans of the pair counts and of the unit counts (a unit = up to 64 sources of one leaf).
// POOLS: the counts come per pool; every pool also learns where its sources of every leaf go (pool order inside a leaf).
template <bool POOLS>
__global__ __launch_bounds__(1024)
void k_lm_plan(LmLists L, int nleaf) {
    __shared__ int s_w[2][16], s_carry[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    __syncthreads();
    const bool dead = *L.overflow != 0;                      // the entry pool ran over: nothing below may be trusted, the call falls back
    for (int l0 = 0; l0 < nleaf; l0 += 1024) {
        const int l = l0 + threadIdx.x;
        int c = 0;
        int pc_of[POOLS ? LM_POOLS : 1];                     // (registers: the loop is unrolled; loads first, stores later, so that they overlap)
        if (l < nleaf && !dead) {
            if (POOLS) {
#pragma unroll
                for (int p = 0; p < LM_POOLS; ++p) pc_of[p] = L.pool_count[(size_t)p * nleaf + l];
#pragma unroll
                for (int p = 0; p < LM_POOLS; ++p) c += pc_of[p];
                L.leaf_count[l] = c;
            } else c = L.leaf_count[l];
        }
        const int u = (c + 63) >> 6;
        int ic = c, iu = u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int a = __shfl_up(ic, off, 64), b = __shfl_up(iu, off, 64); if (lane >= off) { ic += a; iu += b; } }
        if (lane == 63) { s_w[0][wave] = ic; s_w[1][wave] = iu; }
        __syncthreads();
        int bc = s_carry[0], bu = s_carry[1];
        for (int w = 0; w < wave; ++w) { bc += s_w[0][w]; bu += s_w[1][w]; }
        const int pc = bc + ic - c, pu = bu + iu - u;
        if (l < nleaf) {
            L.leaf_start[l] = pc; L.unit_start[l] = pu;
            // (the entry pools admit up to 64 sources per entry, i.e. more units than unit_leaf holds when the descriptors have no
            //  structure: such a call falls back below - and must not have written past the array on its way there)
            for (int k = 0; k < u && pu + k < L.unit_cap; ++k) L.unit_leaf[pu + k] = l;
            if (POOLS && !dead) {
                int run = pc;
#pragma unroll
                for (int p = 0; p < LM_POOLS; ++p) { L.pool_start[(size_t)p * nleaf + l] = run; run += pc_of[p]; }
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) { s_carry[0] = bc + ic; s_carry[1] = bu + iu; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const bool fits = s_carry[0] <= L.pair_cap && s_carry[1] <= L.unit_cap;          // (round 0: the sources themselves)
        if (!fits) *L.overflow = 1;
        L.leaf_start[nleaf] = s_carry[0]; L.unit_start[nleaf] = fits ? s_carry[1] : 0;
    }
}

// every entry's sources to their leaf's stretch of sorted_src: ONE WORKGROUP PER POOL, with the pool's write positions (one per leaf) in
// LDS.  A wave takes 64 entries - a returning LDS atomic per lane - then goes through them entry by entry with lane = source.
// (Positions from returning atomics on one global counter per leaf: 45 us however the rest was arranged - some twenty atomics from all
// over the chip on one address wait for each other.)
__global__ __launch_bounds__(1024)
void k_lm_scatter(LmLists L, int nleaf) {      // grid LM_POOLS, dynamic LDS nleaf ints
    extern __shared__ int s_cur[];
    if (*L.overflow) return;
    const int pool = blockIdx.x;
    for (int l = threadIdx.x; l < nleaf; l += 1024) s_cur[l] = L.pool_start[(size_t)pool * nleaf + l];
    __syncthreads();
    const int n = min(L.entry_cursor[pool], L.pool_cap);
    const int4* __restrict__ ent = L.entries + (size_t)pool * L.pool_cap;
    int* __restrict__ out = L.sorted_src;
    const int lane = threadIdx.x & 63;
    for (int i0 = threadIdx.x - lane; i0 < n; i0 += 1024) {
        const bool valid = i0 + lane < n;
        const int4 e = valid ? ent[i0 + lane] : make_int4(0, 0, 0, 0);
        const unsigned long long m = ((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z;
        const int base = valid ? atomicAdd(&s_cur[e.x], __popcll(m)) : 0;
        const int cnt = min(64, n - i0);
        for (int k = 0; k < cnt; ++k) {
            const unsigned long long mk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(e.w, k) << 32) | (unsigned)__builtin_amdgcn_readlane(e.z, k);
            const int bk = __builtin_amdgcn_readlane(base, k), sk = __builtin_amdgcn_readlane(e.y, k);
            if ((mk >> lane) & 1ull) out[bk + __popcll(mk & ((1ull << lane) - 1ull))] = sk + lane;     // (the position in the search order: k_lm_eval looks the source up)
        }
    }
}

// INIT: round 0 - the units are the sources of every leaf that are at home there (L.sorted_src = the search order itself, L.leaf_count =
// the histogram the ordering made); every source occurs exactly once and gets its first key.
template <bool INIT>
__global__ __launch_bounds__(64 * LM_WAVES)
void k_lm_eval(FmTables t, LmLists L) {
    __shared__ __attribute__((aligned(16))) float s_rows[LM_WAVES][FD * FX_LEAF];
    if (*L.overflow) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = L.unit_start[t.nleaf];
    // workgroups are dealt round-robin over the 8 XCDs: every XCD takes a contiguous stretch of the units (= of the leaves)
    const int vblocks = (total + LM_WAVES - 1) / LM_WAVES, per_xcd = (vblocks + 7) / 8;
    const int xcd = blockIdx.x & 7, step = max(1, (int)gridDim.x >> 3);
    for (int j = blockIdx.x >> 3; j < per_xcd; j += step) {
        const int u = __builtin_amdgcn_readfirstlane((xcd * per_xcd + j) * LM_WAVES + wave);
        if (u >= total) continue;
        const int leaf = __builtin_amdgcn_readfirstlane(L.unit_leaf[u]);
        const int chunk = u - L.unit_start[leaf];
        const int p0 = L.leaf_start[leaf] + chunk * 64, cnt = L.leaf_count[leaf] - chunk * 64;
        const bool valid = lane < cnt;
        const int listed = L.sorted_src[p0 + (valid ? lane : 0)];
        const int src = INIT ? listed : t.sperm[listed];         // (round 0 walks the search order itself; later rounds list positions in it)
        const unsigned long long before = INIT ? LM_KEY_NONE : L.keys[src];
        LmSrc q;
        lm_load_source(t, src, q);
        unsigned long long key = before;
        lm_stage_leaf(t.T, leaf, s_rows[wave], lane);
        lm_eval_leaf(s_rows[wave], t.torig, leaf, q, key);
        if (INIT) { if (valid) L.keys[src] = key; }
        else if (valid && key < before) atomicMin(&L.keys[src], key);
    }
}

__global__ void k_lm_finish(const unsigned long long* __restrict__ keys, int ns, const int* __restrict__ ovf1, const int* __restrict__ ovf2,
                            int* __restrict__ corr, int* __restrict__ host_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { host_flag[0] = (*ovf1 | *ovf2) ? 1 : 0; __threadfence_system(); }
    if (i >= ns) return;
    const unsigned long long k = keys[i];
    corr[i] = (unsigned)(k >> 32) == 0x7f7fffffu ? 0 : (int)(unsigned)k;   // nothing finite -> the reference keeps index 0
}

namespace {

// cyclic Jacobi eigen-solver for a symmetric n x n matrix (host, double): eigenvalues descending, eigenvectors in rows
void jacobi_eigen_host(std::vector<double>& A, int n, std::vector<double>& evals, std::vector<double>& evecs) {
    std::vector<double> V((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
        if (!(off > 1e-30)) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq; A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk; A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq; V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * n + a] > A[(size_t)b * n + b]; });
    evals.resize(n); evecs.assign((size_t)n * n, 0.0);
    for (int r = 0; r < n; ++r) {
        evals[r] = A[(size_t)order[r] * n + order[r]];
        for (int k = 0; k < n; ++k) evecs[(size_t)r * n + k] = V[(size_t)k * n + order[r]];
    }
}

}  // namespace

int fm_index_build(tdv_ctx* ctx, const float* d_ft, int nt, FmIndex* ix) {
    if (!ctx || !d_ft || !ix || nt <= 0) return TDV_ERR_BAD_ARG;
    hipStream_t s = ctx->stream;
    ScopedTimer tm(ctx, TDV_TIMER_FM_INDEX);
    // 1. principal directions of the targets: raw moments on the device, 33 x 33 eigen-problem on the host
    const int mblocks = std::max(1, std::min(512, (nt + 255) / 256));
    const int rows_per_block = (nt + mblocks - 1) / mblocks;
    double *partial, *mom;
    TDV_TRY(ws_alloc(ctx, (size_t)mblocks * FX_NMOM, &partial));
    TDV_TRY(ws_alloc(ctx, (size_t)FX_NMOM, &mom));
    k_fm_moments<<<mblocks, FX_MOM_BLOCK, 0, s>>>(d_ft, nt, rows_per_block, partial);
    k_fm_moments_fold<<<FX_NMOM, 64, 0, s>>>(partial, mblocks, mom);
    // ... and, for the same round trip, which rows are copies of an earlier row (k_fm_dedupe_insert)
    const size_t table_size = sort_pow2((size_t)nt) * 2;
    int *table, *slot_of, *d_kept;
    TDV_TRY(ws_alloc(ctx, table_size, &table));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &slot_of));
    TDV_TRY(ws_alloc(ctx, 1, &d_kept));
    TDV_HIP(ctx, hipMemsetAsync(table, 0xff, table_size * 4, s));
    TDV_HIP(ctx, hipMemsetAsync(d_kept, 0, 4, s));
    k_fm_dedupe_insert<<<(unsigned)((nt + FX_DD_ROWS - 1) / FX_DD_ROWS), FX_DD_ROWS, 0, s>>>(d_ft, nt, table, (unsigned)(table_size - 1), slot_of, d_kept);
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, 64 * 1024));
    double* h_mom = reinterpret_cast<double*>(ctx->pin);
    TDV_HIP(ctx, hipMemcpyAsync(h_mom, mom, FX_NMOM * sizeof(double), hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipMemcpyAsync(ctx->pin + 6144, d_kept, 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    const int nk = *reinterpret_cast<const int*>(ctx->pin + 6144);   // distinct rows: what the index packs
    if (nk <= 0 || nk > nt) return TDV_ERR_INTERNAL;
    std::vector<double> C((size_t)FD * FD), mean(FD), evals, evecs;
    bool finite = true;
    for (int d = 0; d < FD; ++d) { mean[d] = h_mom[561 + d] / nt; finite = finite && std::isfinite(mean[d]); }
    for (int a = 0, e = 0; a < FD; ++a)
        for (int b = a; b < FD; ++b, ++e) {
            const double c = h_mom[e] / nt - mean[a] * mean[b];
            finite = finite && std::isfinite(c);
            C[(size_t)a * FD + b] = C[(size_t)b * FD + a] = c;
        }
    float h_basis[4 * FD];
    double e0 = 1, e1 = 1, e2 = 1;
    if (finite) {
        jacobi_eigen_host(C, FD, evals, evecs);
        for (int r = 0; r < 3; ++r) for (int d = 0; d < FD; ++d) h_basis[r * FD + d] = (float)evecs[(size_t)r * FD + d];
        for (int d = 0; d < FD; ++d) h_basis[3 * FD + d] = (float)mean[d];
        e0 = std::sqrt(std::max(evals[0], 0.0)); e1 = std::sqrt(std::max(evals[1], 0.0)); e2 = std::sqrt(std::max(evals[2], 0.0));
    } else {   // non-finite descriptors: any directions will do (the order only affects speed)
        for (int r = 0; r < 3; ++r) for (int d = 0; d < FD; ++d) h_basis[r * FD + d] = (d % 3 == r) ? 1.f : 0.f;
        for (int d = 0; d < FD; ++d) h_basis[3 * FD + d] = 0.f;
    }
    // how far the f32 directions are from orthonormal decides whether their boxes may be used (k_fm_query, principal_bound_note)
    {
        double dev = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double g = 0.0;
                for (int d = 0; d < FD; ++d) g += (double)h_basis[a * FD + d] * (double)h_basis[b * FD + d];
                dev += (g - (a == b ? 1.0 : 0.0)) * (g - (a == b ? 1.0 : 0.0));
            }
        ix->pscale = (std::sqrt(dev) <= 1e-6) ? (1.0f - 1e-4f) : 0.0f;
    }
    // 2. slab / column counts: S0 * S1 * S2 = number of leaves with S_d proportional to the spread along p_d
    const double nleaf_t = std::max(1.0, (double)nk / FX_LEAF);
    const double tiny = 1e-6 * std::max(e0, 1e-30);
    e0 = std::max(e0, tiny); e1 = std::max(e1, tiny); e2 = std::max(e2, tiny);
    double g = std::cbrt(nleaf_t / (e0 * e1 * e2));
    double s0 = e0 * g, s1 = e1 * g, s2 = e2 * g;
    if (s2 < 1.0) { const double k = std::sqrt(s2); s0 *= k; s1 *= k; s2 = 1.0; }
    if (s1 < 1.0) { s0 *= s1; s1 = 1.0; }
    const int S0 = std::max(1, std::min(FX_MAX_S, (int)std::lround(s0)));
    const int S1 = std::max(1, std::min(FX_MAX_S, (int)std::lround(s1)));
    const int ncol = S0 * S1;
    // equal-count cuts by rank are known without looking at the data
    std::vector<int> h_int((size_t)(S0 + 1) + 3 * ((size_t)ncol + 1));
    int* slab_start = h_int.data(); int* col_start = slab_start + S0 + 1; int* col_row0 = col_start + ncol + 1; int* col_leaf0 = col_row0 + ncol + 1;
    for (int k = 0; k <= S0; ++k) slab_start[k] = (int)((long long)nk * k / S0);
    for (int k = 0; k < S0; ++k) {
        const int c0 = slab_start[k], cnt = slab_start[k + 1] - c0;
        for (int j = 0; j < S1; ++j) col_start[k * S1 + j] = c0 + (int)((long long)cnt * j / S1);
    }
    col_start[ncol] = nk;
    size_t rows = 0;
    for (int c = 0; c < ncol; ++c) {
        col_row0[c] = (int)rows; col_leaf0[c] = (int)(rows / FX_LEAF);
        rows += align_up((size_t)(col_start[c + 1] - col_start[c]), FX_LEAF);
    }
    col_row0[ncol] = (int)rows; col_leaf0[ncol] = (int)(rows / FX_LEAF);
    if (rows == 0) rows = FX_LEAF;
    const int nleaf = (int)(rows / FX_LEAF), ngroup = (nleaf + FX_GROUP - 1) / FX_GROUP;
    // 3. device side
    float *basis, *p0, *p1, *p2; int* d_int; uint4* rec;
    size_t n_pow2 = sort_pow2((size_t)nt);
    TDV_TRY(ws_alloc(ctx, (size_t)4 * FD, &basis));
    TDV_TRY(ws_alloc(ctx, h_int.size(), &d_int));
    TDV_TRY(ws_alloc(ctx, (size_t)S0 + 1, &ix->b0));
    TDV_TRY(ws_alloc(ctx, (size_t)ncol + 1, &ix->b1));
    TDV_TRY(ws_alloc(ctx, (size_t)nleaf, &ix->leaf_p2));
    TDV_TRY(ws_alloc(ctx, rows * FD, &ix->T));
    TDV_TRY(ws_alloc(ctx, rows, &ix->torig));
    const int nchunk = (ngroup + 63) / 64;
    TDV_TRY(ws_alloc(ctx, (size_t)ngroup * 2 * FD * FX_GROUP, &ix->lbox));
    TDV_TRY(ws_alloc(ctx, (size_t)nchunk * 2 * FD * 64, &ix->gbox));
    TDV_TRY(ws_alloc(ctx, (size_t)ngroup * 2 * PD * FX_GROUP, &ix->pbox));
    TDV_TRY(ws_alloc(ctx, (size_t)nchunk * 2 * PD * 64, &ix->gpbox));
    TDV_TRY(ws_alloc(ctx, 1, &ix->amax));
    TDV_TRY(ws_alloc(ctx, (size_t)nleaf * LM_BOX, &ix->sleaf));
    TDV_TRY(ws_alloc(ctx, (size_t)ngroup * LM_BOX, &ix->sgroup));
    const WsMark scratch = ws_mark(ctx);   // everything below is build scratch
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &p0));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &p1));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &p2));
    TDV_TRY(ws_alloc(ctx, n_pow2, &rec));
    unsigned long long *key_a, *key_b; unsigned *row_a, *row_b;
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &key_a));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &key_b));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &row_a));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &row_b));
    float* prow;
    TDV_TRY(ws_alloc(ctx, rows * PD, &prow));
    char* stage = ctx->pin + 8192;   // the moments occupied the first bytes
    std::memcpy(stage, h_basis, sizeof(h_basis));
    std::memcpy(stage + sizeof(h_basis), h_int.data(), h_int.size() * 4);
    TDV_HIP(ctx, hipMemcpyAsync(basis, stage, sizeof(h_basis), hipMemcpyHostToDevice, s));
    TDV_HIP(ctx, hipMemcpyAsync(d_int, stage + sizeof(h_basis), h_int.size() * 4, hipMemcpyHostToDevice, s));
    const int* d_slab_start = d_int; const int* d_col_start = d_int + S0 + 1; const int* d_col_row0 = d_col_start + ncol + 1;
    ix->col_leaf0 = d_col_row0 + ncol + 1;
    ix->ft = d_ft; ix->basis = basis; ix->nt = nt; ix->rows = (int)rows; ix->nleaf = nleaf; ix->ngroup = ngroup; ix->S0 = S0; ix->S1 = S1;
    TDV_HIP(ctx, hipMemsetAsync(ix->leaf_p2, 0, (size_t)nleaf * 4, s));
    TDV_HIP(ctx, hipMemsetAsync(ix->b0, 0, ((size_t)S0 + 1) * 4, s));
    TDV_HIP(ctx, hipMemsetAsync(ix->b1, 0, ((size_t)ncol + 1) * 4, s));
    const unsigned gn = (unsigned)((nt + 255) / 256);
    TDV_HIP(ctx, hipMemsetAsync(ix->amax, 0, 4, s));
    k_fm_project<<<gn, 256, 0, s>>>(d_ft, nt, basis, p0, p1, p2, ix->amax);
    // slabs along p0, columns along p1: two stable radix sorts of (key, row) pairs (csrc/sort.hip; two bitonic sorts of 16-byte records,
    // ~30 launches and 0.19 ms each at 150k rows, until the end of round 2)
    k_fm_key_p0<<<gn, 256, 0, s>>>(p0, nt, table, slot_of, key_a, row_a);
    TDV_TRY(radix_sort_pairs_dev(ctx, key_a, key_b, row_a, row_b, (size_t)nt, 33));   // the distinct rows lead; the copies follow
    k_fm_key_p1<<<gn, 256, 0, s>>>(row_b, nk, d_slab_start, S0, p0, p1, ix->b0, key_a);
    int slab_bits = 1;
    while ((1 << slab_bits) < S0) ++slab_bits;
    TDV_TRY(radix_sort_pairs_dev(ctx, key_a, key_b, row_b, row_a, (size_t)nk, 32 + slab_bits));
    k_fm_rec_p2<<<gn, 256, 0, s>>>(row_a, nk, d_col_start, ncol, p1, p2, ix->b1, rec);
    // the third key only orders the rows INSIDE their column: columns of up to 2,048 rows are sorted by one workgroup each,
    // all in one launch, instead of a third full sort
    int max_col = 0;
    for (int c = 0; c < ncol; ++c) max_col = std::max(max_col, col_start[c + 1] - col_start[c]);
    if (max_col <= segment_sort_max_len()) TDV_TRY(segment_sort_records_dev(ctx, rec, d_col_start, ncol));
    else {
        const size_t nk_pow2 = sort_pow2((size_t)nk);
        if (nk_pow2 > (size_t)nk) TDV_HIP(ctx, hipMemsetAsync(rec + nk, 0xff, (nk_pow2 - (size_t)nk) * sizeof(uint4), s));   // padding sorts last
        TDV_TRY(sort_records_dev(ctx, rec, nk_pow2));
    }
    k_fm_fill_rows<<<(unsigned)((rows * FD + 255) / 256), 256, 0, s>>>(ix->T, ix->torig, rows);
    k_fm_place_rows<<<(unsigned)(((size_t)nk * FD + 255) / 256), 256, 0, s>>>(rec, nk, d_col_start, d_col_row0, ncol, d_ft, p0, p1, p2, ix->T, ix->torig, ix->leaf_p2, prow, rows);
    k_fm_leaf_boxes<<<(ngroup * FX_GROUP * (FD + PD) + 255) / 256, 256, 0, s>>>(ix->T, ix->torig, prow, rows, nleaf, ngroup, ix->lbox, ix->pbox);
    k_fm_group_boxes<<<(nchunk * 64 * (FD + PD) + 255) / 256, 256, 0, s>>>(ix->lbox, ix->pbox, ngroup, nchunk, ix->gbox, ix->gpbox);
    k_lm_box_layout<<<((nleaf + ngroup) * LM_BOX + 255) / 256, 256, 0, s>>>(ix->lbox, ix->pbox, ix->gbox, ix->gpbox, nleaf, ngroup, ix->sleaf, ix->sgroup);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipStreamSynchronize(s));   // the pinned staging is reused by later calls; the scratch is released here
    ws_rewind(ctx, scratch);
    return TDV_OK;
}

// Pass A over every source, then the sources it gave up on: few of them -> pass B (8 waves each on the packed index);
// many of them (descriptors without structure, or a plateau of near-identical rows: no box can exclude anything) -> the
// plain scan over just those sources, seeded with pass A's distances, which is the efficient way to do brute force.
template <int K>
static int launch_fm_query(tdv_ctx* ctx, const FmTables& t, const FmIndex& ix, int* overflow_count, int* overflow_list, int* overflow_src,
                           float* part_d, int* part_j, int* d_corr) {
    hipStream_t s = ctx->stream;
    const int waves = (t.ns + K - 1) / K;
    const int blocks_per_xcd = ((waves + FM_BLOCK / 64 - 1) / (FM_BLOCK / 64) + 7) / 8, blocks = blocks_per_xcd * 8;
    // tuning knobs.  The leaf budget of pass A was 32 while a leaf cost what it did in the middle of round 2; with the packed
    // arithmetic and the scalar-side savings a wave is better off opening up to 128 leaves itself than handing over early
    // (143k x 151k relief part 0.81 -> 0.73 ms, cuboid 200k x 200k 4.05 -> 2.04 ms, random rows 100k x 100k 16.8 -> 17.7 ms)
    static const int leaf_limit = study_env("TDV_FM_LIMIT") ? atoi(study_env("TDV_FM_LIMIT")) : 128;
    static const int heavy_groups = study_env("TDV_FM_HEAVY") ? atoi(study_env("TDV_FM_HEAVY")) : 8;
    const bool stats = study_env("TDV_FM_STATS") != nullptr;   // study knob: counts of box tests and leaf openings, printed to stderr
    unsigned long long* d_stats = nullptr;
    if (stats) {
        TDV_TRY(ws_alloc(ctx, 12, &d_stats));
        TDV_HIP(ctx, hipMemsetAsync(d_stats, 0, 96, s));
        k_fm_query<K, true><<<blocks, FM_BLOCK, 0, s>>>(t, blocks_per_xcd, leaf_limit, heavy_groups, overflow_count, overflow_list, overflow_src, part_d, part_j, d_corr, d_stats);
    } else {
        k_fm_query<K, false><<<blocks, FM_BLOCK, 0, s>>>(t, blocks_per_xcd, leaf_limit, heavy_groups, overflow_count, overflow_list, overflow_src, part_d, part_j, d_corr, nullptr);
    }
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, 256));
    int* h_over = reinterpret_cast<int*>(ctx->pin);
    TDV_HIP(ctx, hipMemcpyAsync(h_over, overflow_count, 8, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    const int n_b = h_over[0], n_scan = h_over[1];
    // the scan only pays when it has enough sources to fill the chip; a handful of outliers is cheaper on pass B
    const bool scan_them = (long long)n_scan * K >= 2048;
    for (int cls = 0; cls < 2; ++cls) {
        const int n = cls == 0 ? n_b : (scan_them ? 0 : n_scan);
        if (n <= 0) continue;
        const int* list = overflow_list + (cls ? t.ns + 1 : 0);
        const int bblocks = std::min(n, 1024);
        if (stats) k_fm_query_overflow<K, true><<<bblocks, FMB_WAVES * 64, 0, s>>>(t, overflow_count + cls, list, part_d, part_j, d_corr, d_stats);
        else k_fm_query_overflow<K, false><<<bblocks, FMB_WAVES * 64, 0, s>>>(t, overflow_count + cls, list, part_d, part_j, d_corr, nullptr);
    }
    if (n_scan > 0 && scan_them) {
        const int n_list = n_scan * K;
        const int ns_pad = (int)align_up((size_t)t.ns, FM_SRC_PER_BLOCK);
        const int blocks_x = (n_list + FM_SRC_PER_BLOCK - 1) / FM_SRC_PER_BLOCK;
        const int want = (6144 + blocks_x - 1) / blocks_x;
        int nsplit = std::max(1, std::min(std::min(want, std::max(1, ix.nt / 128)), 256));   // few sources: many target splits, or the chip stays empty
        const int per_split = (ix.nt + nsplit - 1) / nsplit;
        nsplit = (ix.nt + per_split - 1) / per_split;
        float* pd; int* pj;
        TDV_TRY(ws_alloc(ctx, (size_t)nsplit * ns_pad, &pd));
        TDV_TRY(ws_alloc(ctx, (size_t)nsplit * ns_pad, &pj));
        k_feature_match_scan<true, false><<<dim3(blocks_x, nsplit), FM_BLOCK, 0, s>>>(t.fs, t.ns, ns_pad, ix.ft, 0, ix.nt, per_split, part_d, overflow_src, n_list, pd, pj);
        k_feature_match_combine_list<<<(n_list + 255) / 256, 256, 0, s>>>(overflow_src, n_list, t.ns, ns_pad, nsplit, pd, pj, d_corr);
    }
    TDV_CHECK_LAUNCH(ctx);
    if (stats) {
        unsigned long long h[12];
        TDV_HIP(ctx, hipMemcpyAsync(h, d_stats, 96, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipStreamSynchronize(s));
        fprintf(stderr, "[tdv] fm query: %d sources x %d leaves, %d groups, %d sources per wave, %llu waves: per wave %.1f group-chunk tests, "
                "%.1f groups visited, %.1f leaves opened (max %llu); wave time mean %.1f us max %.1f us; gave up: %d waves to pass B "
                "(%llu helper waves, %.1f leaves each, max %llu), %d waves to the plain scan\n",
                t.ns, t.nleaf, t.ngroup, K, h[0], (double)h[1] / h[0], (double)h[2] / h[0], (double)h[3] / h[0], h[4],
                (double)h[5] / h[0] * 0.01, (double)h[6] * 0.01, n_b, h[8], h[8] ? (double)h[9] / h[8] : 0.0, h[10], n_scan);
    }
    return TDV_OK;
}

// The leaf-major search (kernels above).  *done = false: the pair pool ran over (descriptors without structure) - the caller
// runs k_fm_query and its fall-backs instead.  One synchronisation, like launch_fm_query.
static int launch_fm_leafmajor(tdv_ctx* ctx, const FmTables& t, const FmIndex& ix, int* home_hist /* sources per home leaf: what the ordering counted */, int* d_corr, bool* done) {
    hipStream_t s = ctx->stream;
    *done = false;
    const int nleaf = ix.nleaf;
    const size_t pair_cap = (size_t)LM_PAIRS_PER_SOURCE * (size_t)t.ns + 4096;
    const size_t pool_cap = ((size_t)t.ns + 4096) / LM_POOLS + 64;     // entries: one box x up to 64 sources each
    if (pair_cap > (size_t)INT_MAX / 2) return TDV_OK;
    const size_t unit_cap = pair_cap / 64 + (size_t)nleaf + 2, per_round = (size_t)LM_POOLS * nleaf + LM_POOLS + 1;
    const int waves = (t.ns + 63) / 64;
    int *zeroed, *sorted_src, *leaf_count, *pool_start, *leaf_start, *unit_start, *unit_leaf; unsigned long long* keys;
    TDV_TRY(ws_alloc(ctx, 2 * per_round + 1, &zeroed));
    int4* entries;
    TDV_TRY(ws_alloc(ctx, pool_cap * LM_POOLS, &entries));
    TDV_TRY(ws_alloc(ctx, pair_cap, &sorted_src));
    TDV_TRY(ws_alloc(ctx, (size_t)nleaf, &leaf_count));
    TDV_TRY(ws_alloc(ctx, (size_t)LM_POOLS * nleaf, &pool_start));
    TDV_TRY(ws_alloc(ctx, (size_t)nleaf + 1, &leaf_start));
    TDV_TRY(ws_alloc(ctx, (size_t)nleaf + 1, &unit_start));
    TDV_TRY(ws_alloc(ctx, unit_cap, &unit_leaf));
    TDV_TRY(ws_alloc(ctx, (size_t)t.ns, &keys));
    TDV_TRY(pin_reserve(ctx, 256));
    int* h_flag = reinterpret_cast<int*>(ctx->pin);
    h_flag[0] = 1;
    TDV_HIP(ctx, hipMemsetAsync(zeroed, 0, (2 * per_round + 1) * 4, s));
    LmLists L[3];
    for (int r = 0; r < 2; ++r) {
        int* z = zeroed + r * per_round;
        L[r + 1] = LmLists{z, z + (size_t)LM_POOLS * nleaf, z + (size_t)LM_POOLS * nleaf + LM_POOLS, leaf_count, pool_start, entries, (int)pool_cap, (int)pair_cap, leaf_start, unit_start, unit_leaf, sorted_src, keys, (int)unit_cap};
    }
    // round 0: every source against its home leaf.  The search order is sorted by home leaf, so it IS the sorted pair list.
    L[0] = LmLists{nullptr, nullptr, zeroed + 2 * per_round, home_hist, nullptr, nullptr, 0, INT_MAX, leaf_start, unit_start, unit_leaf, const_cast<int*>(t.sperm), keys, (int)unit_cap};
    unsigned long long* d_stats = nullptr;
    if (study_env("TDV_FM_STATS")) { TDV_TRY(ws_alloc(ctx, 16, &d_stats)); TDV_HIP(ctx, hipMemsetAsync(d_stats, 0, 128, s)); }
    static const int eval_blocks = study_env("TDV_LM_EVAL_BLOCKS") ? atoi(study_env("TDV_LM_EVAL_BLOCKS")) : 4096;   // tuning knob (a multiple of 8; 1024 / 2048 / 4096 / 8192: 0.51 / 0.49 / 0.477 / 0.478 ms at 143k x 151k)
    // One round of box tests after the home leaves (12.1 pairs per source at 143k x 151k).  TDV_LM_ROUNDS=2: the home groups first, the
    // other groups with the bounds those left (9.7 pairs per source, but a second set of launches: 0.525 against 0.487 ms).
    const char* rounds_env = study_env("TDV_LM_ROUNDS");
    const int rounds = (ix.ngroup > 1 && rounds_env && atoi(rounds_env) == 2) ? 2 : 1;
    k_lm_plan<false><<<1, 1024, 0, s>>>(L[0], nleaf);
    k_lm_eval<true><<<eval_blocks, 64 * LM_WAVES, 0, s>>>(t, L[0]);
    for (int r = 1; r <= rounds; ++r) {
        if (rounds == 1 && ix.ngroup > 1) k_lm_boxes<3><<<waves, 64 * LM_BOX_WAVES, 0, s>>>(t, ix.sleaf, ix.sgroup, L[r], d_stats);
        else if (r == 1) k_lm_boxes<1><<<waves, 64 * LM_BOX_WAVES, 0, s>>>(t, ix.sleaf, ix.sgroup, L[r], d_stats);
        else k_lm_boxes<2><<<waves, 64 * LM_BOX_WAVES, 0, s>>>(t, ix.sleaf, ix.sgroup, L[r], d_stats ? d_stats + 8 : nullptr);
        k_lm_plan<true><<<1, 1024, 0, s>>>(L[r], nleaf);
        k_lm_scatter<<<LM_POOLS, 1024, (size_t)nleaf * 4, s>>>(L[r], nleaf);
        k_lm_eval<false><<<eval_blocks, 64 * LM_WAVES, 0, s>>>(t, L[r]);
    }
    k_lm_finish<<<(t.ns + 255) / 256, 256, 0, s>>>(keys, t.ns, L[1].overflow, L[2].overflow, d_corr, h_flag);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipStreamSynchronize(s));
    *done = h_flag[0] == 0;
    if (d_stats) {                                           // study knob
        int h[2][LM_POOLS + 1];
        long long pairs[2] = {0, 0};
        for (int r = 0; r < 2; ++r) {
            TDV_HIP(ctx, hipMemcpy(h[r], L[r + 1].entry_cursor, (LM_POOLS + 1) * 4, hipMemcpyDeviceToHost));
            for (int k = 0; k < LM_POOLS; ++k) pairs[r] += h[r][k];
        }
        fprintf(stderr, "[tdv] fm leaf-major: %d sources x %d leaves in %d groups: round 1 %lld entries (%.2f per source)%s, round 2 %lld entries (%.2f per source)%s\n",
                t.ns, nleaf, ix.ngroup, pairs[0], (double)pairs[0] / t.ns, h[0][LM_POOLS] ? " OVERFLOW" : "", pairs[1], (double)pairs[1] / t.ns, h[1][LM_POOLS] ? " OVERFLOW" : "");
        unsigned long long st[16];
        TDV_HIP(ctx, hipMemcpy(st, d_stats, 128, hipMemcpyDeviceToHost));
        for (int r = 0; r < 2; ++r) {
            const unsigned long long* q = st + 8 * r; const double w = (double)std::max(1ull, q[0]);
            fprintf(stderr, "[tdv]   boxes round %d: %llu waves, per wave %.1f us (max %.1f), %.1f us until the source is loaded, %.1f us last flush; %.1f 3-D tests, %.1f 33-D tests, %.1f boxes emitted\n",
                    r + 1, q[0], q[1] / w * 0.01, q[7] * 0.01, q[2] / w * 0.01, q[3] / w * 0.01, q[4] / w, q[5] / w, q[6] / w);
        }
    }
    return TDV_OK;
}

int feature_match_indexed_dev(tdv_ctx* ctx, const float* d_fs, int ns, const FmIndex& ix, int* d_corr) {
    if (!ctx || !d_fs || !d_corr || ns < 0) return TDV_ERR_BAD_ARG;
    if (ns == 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    int bucket_shift = 0;
    while ((ix.nleaf >> bucket_shift) > FMP_BUCKETS) ++bucket_shift;
    int *home, *bucket_of, *sperm, *hist, *cursor, *start, *d_total; float* sp; unsigned* amax_s;
    int *overflow_count, *overflow_list, *overflow_src, *part_j; float* part_d;
    int* zeroed;                                                   // hist | cursor | overflow_count[2] | amax_s: one memset
    TDV_TRY(ws_alloc(ctx, (size_t)2 * FMP_BUCKETS + 4, &zeroed));
    hist = zeroed; cursor = zeroed + FMP_BUCKETS; overflow_count = zeroed + 2 * FMP_BUCKETS; amax_s = reinterpret_cast<unsigned*>(zeroed + 2 * FMP_BUCKETS + 2);
    TDV_TRY(ws_alloc(ctx, (size_t)2 * ns + 8, &overflow_list));   // two halves: pass-B class, scan class
    TDV_TRY(ws_alloc(ctx, (size_t)ns + 8, &overflow_src));
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &part_d));
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &part_j));
    TDV_TRY(ws_alloc(ctx, (size_t)ns * 4, &sp));
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &home));
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &bucket_of));
    TDV_TRY(ws_alloc(ctx, (size_t)ns, &sperm));
    TDV_TRY(ws_alloc(ctx, (size_t)FMP_BUCKETS + 1, &start));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    ScopedTimer tm(ctx, TDV_TIMER_FEATURE_MATCH);
    TDV_HIP(ctx, hipMemsetAsync(zeroed, 0, ((size_t)2 * FMP_BUCKETS + 4) * 4, s));
    k_fm_locate<<<(ns + 255) / 256, 256, 0, s>>>(d_fs, ns, ix.basis, ix.S0, ix.S1, ix.b0, ix.b1, ix.col_leaf0, ix.leaf_p2, bucket_shift, home, bucket_of, sp, amax_s);
    const int sblocks = (ns + FMP_SORT_BLOCK - 1) / FMP_SORT_BLOCK;
    k_fm_bucket_hist<<<sblocks, FMP_SORT_BLOCK, 0, s>>>(bucket_of, ns, hist);
    TDV_TRY(exclusive_scan_dev(ctx, hist, FMP_BUCKETS, start, d_total));
    k_fm_scatter<<<sblocks, FMP_SORT_BLOCK, 0, s>>>(bucket_of, ns, start, cursor, sperm);
    static const int force_k = study_env("TDV_FM_K") ? atoi(study_env("TDV_FM_K")) : 0;   // tuning knob (sources per wave)
    const int k = force_k ? force_k : 2;
    float* fs2 = nullptr;
    const char* lm = getenv("TDV_FM_LEAFMAJOR");                  // A/B knob, read per call: 0 = round 2's walk (k_fm_query) for everything
    if (!(lm && atoi(lm) == 0) && ix.sleaf && bucket_shift == 0) {     // (bucket_shift: more than 16,384 leaves - the ordering's histogram is then not per leaf)
        FmTables t{d_fs, sperm, home, ns, nullptr, ix.T, ix.torig, ix.nleaf, ix.ngroup, ix.lbox, ix.gbox, ix.pbox, ix.gpbox, sp, ix.amax, amax_s, ix.pscale};
        bool done = false;
        TDV_TRY(launch_fm_leafmajor(ctx, t, ix, hist, d_corr, &done));
        if (done) { ctx->last_fm_path = TDV_FM_PATH_LEAF_MAJOR; return TDV_OK; }
    }
    ctx->last_fm_path = TDV_FM_PATH_WALK;
    if (k >= 2) {
        const int kk = k >= 4 ? 4 : 2;
        const size_t n2 = ((size_t)ns + kk - 1) / kk * kk * FD;
        TDV_TRY(ws_alloc(ctx, n2, &fs2));
        k_fm_interleave_rows<<<(unsigned)((n2 + 255) / 256), 256, 0, s>>>(d_fs, sperm, ns, kk, fs2);
    }
    FmTables t{d_fs, sperm, home, ns, fs2, ix.T, ix.torig, ix.nleaf, ix.ngroup, ix.lbox, ix.gbox, ix.pbox, ix.gpbox, sp, ix.amax, amax_s, ix.pscale};
#ifdef TDV_STUDY
    if (k >= 4) return launch_fm_query<4>(ctx, t, ix, overflow_count, overflow_list, overflow_src, part_d, part_j, d_corr);
    if (k < 2) return launch_fm_query<1>(ctx, t, ix, overflow_count, overflow_list, overflow_src, part_d, part_j, d_corr);
#endif
    return launch_fm_query<2>(ctx, t, ix, overflow_count, overflow_list, overflow_src, part_d, part_j, d_corr);      // two sources per wave (1 and 4: measured slower, study build)
}

int feature_match_dev(tdv_ctx* ctx, const float* d_fs, int ns, const float* d_ft, int nt, int* d_corr) {
    if (!ctx || !d_fs || !d_ft || !d_corr || ns < 0 || nt < 0) return TDV_ERR_BAD_ARG;
    if (ns == 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    if (nt == 0) { TDV_HIP(ctx, hipMemsetAsync(d_corr, 0, (size_t)ns * 4, s)); return TDV_OK; }
    ctx->last_fm_path = TDV_FM_PATH_SCAN;
    const char* brute = getenv("TDV_FM_BRUTE");         // A/B knobs: same results every way
    const char* keyorder = study_env("TDV_FM_KEYORDER");
    if (!brute && ns >= 4096 && nt >= 2048) {
#ifdef TDV_STUDY
        if (keyorder) return feature_match_keyorder_dev(ctx, d_fs, ns, d_ft, nt, d_corr);      // round 1's key-ordered pruned scan
#endif
        (void)keyorder;
        FmIndex ix;
        TDV_TRY(fm_index_build(ctx, d_ft, nt, &ix));
        return feature_match_indexed_dev(ctx, d_fs, ns, ix, d_corr);
    }
    static const bool early = study_env("TDV_FM_NO_EARLY_EXIT") == nullptr;   // A/B knob: same results either way
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
            k_feature_match_scan<true, true><<<dim3(blocks_x, 1), FM_BLOCK, 0, s>>>(d_fs, ns, ns_pad, d_ft, 0, n_seed, n_seed, nullptr, nullptr, 0, pd, pj);
            if (nsplit)   // (ordering the sources by seed distance was measured: no gain on FPFH descriptors, so rows stay in place)
                k_feature_match_scan<true, true><<<dim3(blocks_x, nsplit), FM_BLOCK, 0, s>>>(d_fs, ns, ns_pad, d_ft, n_seed, nt, per_split, pd, nullptr, 0,
                                                                                      pd + ns_pad, pj + ns_pad);
        } else {
            k_feature_match_scan<false, true><<<dim3(blocks_x, nsplit), FM_BLOCK, 0, s>>>(d_fs, ns, ns_pad, d_ft, 0, nt, per_split, nullptr, nullptr, 0, pd, pj);
        }
    }
    k_feature_match_combine<<<(ns + 255) / 256, 256, 0, s>>>(ns, ns_pad, nparts, pd, pj, d_corr);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

}  // namespace tdv
