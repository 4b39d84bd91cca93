// Exact neighbour searches on gfx950 and the per-point estimators built on them.
//
// Replaces (no GPU entry point exists in the reference; src/pipeline.cpp:93-95 calls the CPU statics):
//   findKNN + Registration::estimateNormals   /root/reference/src/registration.cpp:63-81, :105-130
//   findRadiusNN + Registration::computeFPFH  /root/reference/src/registration.cpp:83-102, :133-201
//
// The reference scans all N points per query.  Here every (query, target) distance that is evaluated uses the
// reference's expression, d2 = dx*dx + (dy*dy + dz*dz) with no FMA, and every target that is NOT evaluated is
// excluded by an exact bounding-box lower bound (float subtraction, multiplication and addition are monotone under
// round-to-nearest, so lb <= fl(d2) for every point inside the box: no margin), so the neighbour lists are the
// reference's lists bit for bit, ties included:
//   0. the cloud is ordered along a Morton curve (counting sort on the top 21 bits of the 30-bit code) and the
//      bounding boxes of its 64-point leaves and 4096-point groups of leaves are built once per cloud;
//   1. k_query_wave: ONE WAVE PER QUERY.  Boxes are tested 64 per step (one per lane) against the query point and
//      its bound — first the 4096-point groups, then the 64 leaves of each group that passes; the 64 targets of a
//      leaf that passes are evaluated one per lane with coalesced loads; every target with d2 <= bound is appended to the query's row in LDS (ballot prefix: no atomics,
//      deterministic order).  When the row fills, it is sorted in-wave (bitonic network on 64-bit keys
//      d2 bits : original index, whose unsigned order is the reference's (d2, index) order), cut to the best k, and
//      the bound drops to the k-th: exact for any number of ties.  At the end the row is sorted and its first k
//      entries are the list.  kNN starts unbounded: the query's own leaf (its 64 curve neighbours) gives the first
//      bound, then the walk; the radius search (FPFH, cap 100) starts from r^2.
// In the batched chain normals_fpfh_dev shares ONE radius search between normals and FPFH: a radius list is sorted
// by (d2, idx), so its first k entries are the k nearest neighbours wherever it holds >= k; only the deficient
// points go through the kNN search as a subset.
// Per-point estimators run one lane per point with sequential sums in neighbour order, i.e. the same f32
// expression trees as the CPU loops; atan2 is glibc's atan2f restated (libm_f32.hpp): the same theta, the same histogram bin.
#include "tdv_internal.hpp"
#include "libm_f32.hpp"
#include "device_linalg.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <algorithm>
#include <cstdlib>

namespace tdv {

constexpr int KN_BLOCK = 256;

// XCD-aware block order (cdna_hip_programming.md T1, bijective form): workgroups are dealt round-robin over the 8
// XCDs, so block b of the launch takes the logical position that gives every XCD one CONTIGUOUS eighth of the curve-
// ordered work: the rows a workgroup gathers are then mostly in its own XCD's 4 MiB L2.  A speed choice only.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// ------------------------------------------------------------------ spatial sort
__global__ __launch_bounds__(256)
void k_bbox_partial(const float* __restrict__ xyz, int n, float* __restrict__ part /* [blocks][6] */) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) { float v = xyz[3 * (size_t)i + c]; if (v == v) { mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); } }
    __shared__ float red[4][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[c] = fminf(mn[c], __shfl_down(mn[c], off, 64)); mx[c] = fmaxf(mx[c], __shfl_down(mx[c], off, 64)); }
    }
    if ((threadIdx.x & 63) == 0) { for (int c = 0; c < 3; ++c) { red[threadIdx.x >> 6][c] = mn[c]; red[threadIdx.x >> 6][3 + c] = mx[c]; } }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(64)
void k_bbox_final(const float* __restrict__ part, int nblocks, float* __restrict__ bbox /* min xyz, max xyz */) {
    // one wave: lane l folds partials l, l+64, ... for all 6 components, then a shuffle reduction
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = c < 3 ? INFINITY : -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int c = 0; c < 6; ++c) v[c] = c < 3 ? fminf(v[c], part[b * 6 + c]) : fmaxf(v[c], part[b * 6 + c]);
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { float o = __shfl_down(v[c], off, 64); v[c] = c < 3 ? fminf(v[c], o) : fmaxf(v[c], o); }
    if (threadIdx.x == 0) for (int c = 0; c < 6; ++c) bbox[c] = v[c];
}
__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// Curve order by counting sort: bucket = the top 12..21 bits (by cloud size) of the 30-bit Morton code (up to a 128^3
// grid over the cloud's bounding box).  Only locality matters for the searches (their results do not depend on the order, nor on
// the arbitrary order of the few points that share a bucket), so 5 small launches replace a ~60-launch full sort.
constexpr int MORTON_MAX_BUCKET_BITS = 21;
__device__ __forceinline__ int morton_bucket(const float* __restrict__ xyz, int i, const float* __restrict__ bbox, int bits) {
    unsigned q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float lo = bbox[c], ext = bbox[3 + c] - lo;
        float t = ext > 0.f ? (xyz[3 * (size_t)i + c] - lo) / ext : 0.f;   // only the visiting order depends on this
        t = fminf(fmaxf(t, 0.f), 1.f);
        if (!(t == t)) t = 0.f;
        q[c] = (unsigned)(t * 1023.f);
    }
    unsigned code = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    return (int)(code >> (30 - bits));
}
__global__ void k_morton_hist(const float* __restrict__ xyz, int n, const float* __restrict__ bbox, int bits, int* __restrict__ bucket_of,
                              int* __restrict__ hist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = morton_bucket(xyz, i, bbox, bits);
    bucket_of[i] = b;
    atomicAdd(&hist[b], 1);
}
__global__ void k_morton_scatter(const float* __restrict__ xyz, int n, int n_pad, const int* __restrict__ bucket_of,
                                 const int* __restrict__ start, int* __restrict__ cursor,
                                 float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz, int* __restrict__ orig) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    if (i < n) {
        const int b = bucket_of[i];
        const int pos = start[b] + atomicAdd(&cursor[b], 1);
        sx[pos] = xyz[3 * (size_t)i]; sy[pos] = xyz[3 * (size_t)i + 1]; sz[pos] = xyz[3 * (size_t)i + 2]; orig[pos] = i;
    } else { sx[i] = INFINITY; sy[i] = INFINITY; sz[i] = INFINITY; orig[i] = INT_MAX; }   // padding positions n..n_pad
}

// Bounding boxes of every 64-point run ("leaf") of the ordered cloud and of every group of 64 leaves ("top",
// 4096 points): a wave tests 64 boxes per step, one per lane, and evaluates a leaf's 64 points in one step.
// box layout: 6 arrays [minx|miny|minz|maxx|maxy|maxz][count]
__global__ void k_leaf_boxes(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                             int n_leaf, float* __restrict__ lb) {
    // one wave per leaf
    const int leaf = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (leaf >= n_leaf) return;
    const int lane = threadIdx.x & 63;
    const float v[3] = {sx[leaf * 64 + lane], sy[leaf * 64 + lane], sz[leaf * 64 + lane]};   // +inf padding widens the last box: still valid
    float mn[3] = {v[0], v[1], v[2]}, mx[3] = {v[0], v[1], v[2]};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
    if (lane < 3) lb[(size_t)lane * n_leaf + leaf] = lane == 0 ? mn[0] : (lane == 1 ? mn[1] : mn[2]);
    else if (lane < 6) lb[(size_t)lane * n_leaf + leaf] = lane == 3 ? mx[0] : (lane == 4 ? mx[1] : mx[2]);
}
// one wave per group: lane = leaf (a lane per group walking its 64 leaves one after the other took 14 us for 37 groups)
__global__ __launch_bounds__(64)
void k_top_boxes(const float* __restrict__ lb, int n_leaf, int n_top, float* __restrict__ tb) {
    const int u = blockIdx.x, lane = threadIdx.x;
    if (u >= n_top) return;
    const int c = u * 64 + lane;
    float mn[3], mx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = c < n_leaf ? lb[(size_t)a * n_leaf + c] : INFINITY; mx[a] = c < n_leaf ? lb[(size_t)(3 + a) * n_leaf + c] : -INFINITY; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
    if (lane < 3) tb[(size_t)lane * n_top + u] = lane == 0 ? mn[0] : (lane == 1 ? mn[1] : mn[2]);
    else if (lane < 6) tb[(size_t)lane * n_top + u] = lane == 3 ? mx[0] : (lane == 4 ? mx[1] : mx[2]);
}

// Lower bound on the reference's float d2 = dx*dx + (dy*dy + dz*dz) between ANY query inside [qmin,qmax] and ANY target
// inside [bmin,bmax]: per-axis gaps by one float subtraction each, then the same expression tree.  Float subtraction,
// multiplication and addition are monotone under round-to-nearest, so lb <= fl(d2) for every such pair — no margin
// is needed, and "lb > bound" proves that nothing in the box can pass "d2 <= bound".
__device__ __forceinline__ float box_lower_bound(const float* __restrict__ box, int count, int idx,
                                                 const float (&qmin)[3], const float (&qmax)[3]) {
    float g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float bmin = box[(size_t)a * count + idx], bmax = box[(size_t)(3 + a) * count + idx];
        g[a] = fmaxf(0.f, fmaxf(bmin - qmax[a], qmin[a] - bmax));
    }
    return g[0] * g[0] + (g[1] * g[1] + g[2] * g[2]);
}

// ------------------------------------------------------------------ one wave per query: walk, collect, select
#ifndef QW_WAVES_VALUE
#define QW_WAVES_VALUE 1
#endif
constexpr int QW_WAVES = QW_WAVES_VALUE;   // queries (waves) per workgroup of k_query_wave (measured kNN 200k: 1: 0.383 ms, 4: 0.401, 16: 0.487)
// kNN start, measured at 50k/100k/200k (k = 30): own leaf only 0.55/0.55/1.00 ms; own leaf +-1 0.23/0.41/0.61; with nearest-leaf-first
// inside a group 0.19/0.37/0.67 (kept: it also protects clouds of uneven density); +-2 leaves no better
constexpr int QW_SEED_SPAN = 1;
constexpr bool QW_BEST_FIRST = true;

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int j) {
    const unsigned lo = __shfl_xor((unsigned)v, j, 64), hi = __shfl_xor((unsigned)(v >> 32), j, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// Ascending bitonic sort of the first NSORT (64, 128, ... 64*R) keys of a wave; element i = r*64 + lane.  The
// network is fully unrolled (compile-time strides: the exchanges become DPP / swizzle / permute with constant
// patterns and the direction masks fold to one bit test each).  Keys are distinct (distinct indices) except the ~0
// padding, which sorts last.
template <int R, int NSORT>
__device__ __forceinline__ void wave_sort_keys_fixed(unsigned long long (&key)[R], int lane) {
#pragma unroll
    for (int kk = 2; kk <= NSORT; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 1; j >>= 1) {
            if (j >= 64) {   // partner is another key of the same lane
                const int dr = j >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((r & dr) != 0 || (r + dr) * 64 >= NSORT) continue;
                    const bool asc = ((r * 64) & kk) == 0;   // kk >= 128 here: the bit lies in r, not in the lane
                    const bool swap = asc == (key[r + dr] < key[r]);
                    const unsigned long long a = key[r], c = key[r + dr];
                    key[r] = swap ? c : a;
                    key[r + dr] = swap ? a : c;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r * 64 >= NSORT) continue;
                    const unsigned long long other = shfl_xor_u64(key[r], j);
                    const bool keep_min = ((((r * 64) | lane) & kk) == 0) == ((lane & j) == 0);
                    key[r] = ((other < key[r]) == keep_min) ? other : key[r];
                }
            }
        }
    }
}
template <int R>
__device__ __forceinline__ void wave_sort_keys(unsigned long long (&key)[R], int m, int lane) {
    // the span is wave-uniform: the smallest power-of-two multiple of 64 that holds m keys
    if (m <= 64) wave_sort_keys_fixed<R, 64>(key, lane);
    else if (R >= 2 && m <= 128) wave_sort_keys_fixed<R, (R >= 2 ? 128 : 64)>(key, lane);
    else if (R >= 4 && m <= 256) wave_sort_keys_fixed<R, (R >= 4 ? 256 : 64)>(key, lane);
    else wave_sort_keys_fixed<R, 64 * R>(key, lane);
}

// row[0..m) -> registers, sorted ascending
template <int R>
__device__ __forceinline__ void load_sort_row(const unsigned long long* row, int m, int lane, unsigned long long (&key)[R]) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the row was written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        key[r] = e < m ? row[e] : ~0ull;
    }
    wave_sort_keys<R>(key, m, lane);
}

// lists[original index * stride + r] = the first min(k, found) targets in (d2, idx) order with d2 <= bound
// (bound = bound[slot] if given, else bound0), cnt_out[original index] = their number.  Requires k <= 64*R - 64.
template <int R, int SEED_SPAN, bool BEST_FIRST>
__global__ __launch_bounds__(64 * QW_WAVES)
void k_query_wave(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                  const int* __restrict__ orig, int n, int n_leaf, const float* __restrict__ lbox, int n_top, const float* __restrict__ tbox,
                  const int* __restrict__ qsel, int nqq, const float* __restrict__ bound, float bound0, int seed_own,
                  int k, int stride, int* __restrict__ lists, int* __restrict__ cnt_out,
                  const int* __restrict__ okey /* sorted position -> tie-break id (null: orig) */,
                  const int* __restrict__ key2idx /* tie-break id -> array index written to the lists (null: the id itself) */,
                  const int* __restrict__ inst_leaf = nullptr /* several clouds in one array: cloud b owns leaves [inst_leaf[b], inst_leaf[b + 1]) */,
                  int n_inst = 0) {
    constexpr int ROW = 64 * R;
    __shared__ unsigned long long rows[QW_WAVES][ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = xcd_contiguous_block(blockIdx.x, gridDim.x) * QW_WAVES + wave;   // curve order: an XCD works on one stretch
    if (slot >= nqq) return;   // wave-uniform; no block-level barrier below
    unsigned long long* row = rows[wave];
    const int sp = qsel ? qsel[slot] : slot;
    const float qx = sx[sp], qy = sy[sp], qz = sz[sp];
    const float qp[3] = {qx, qy, qz};
    float B = bound ? bound[slot] : bound0;   // wave-uniform; only ever decreases
    int wcnt = 0;                             // wave-uniform fill of the row

    // keep the best k of the row (needs wcnt >= k) and drop the bound to the k-th.  No sort: the k-th smallest key is
    // found by bisection on its bits with ballot counts (distance bits first, then — only when several candidates tie
    // at that distance — the index bits), and the survivors are packed by ballot prefix.
    auto compact = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned hi[R], lo[R]; bool has[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 64 + lane;
            has[r] = e < wcnt;
            const unsigned long long key = has[r] ? row[e] : ~0ull;
            hi[r] = (unsigned)(key >> 32); lo[r] = (unsigned)key;
        }
        auto count_if = [&](auto pred) {
            int c = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) c += __popcll(__ballot(has[r] && pred(r)));
            return c;
        };
        unsigned kd = 0;   // k-th smallest distance bits: count(hi < kd) < k <= count(hi <= kd)
        for (int b = 31; b >= 0; --b) {
            const unsigned cand = kd | (1u << b);
            if (count_if([&](int r) { return hi[r] < cand; }) < k) kd = cand;
        }
        const int below = count_if([&](int r) { return hi[r] < kd; });
        unsigned ki = 0xffffffffu;   // among the candidates AT that distance keep the (k - below) lowest indices
        if (count_if([&](int r) { return hi[r] <= kd; }) > k) {
            ki = 0;
            const int need = k - below;
            for (int b = 31; b >= 0; --b) {
                const unsigned cand = ki | (1u << b);
                if (count_if([&](int r) { return hi[r] == kd && lo[r] < cand; }) < need) ki = cand;
            }
        }
        int base = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool keep = has[r] && (hi[r] < kd || (hi[r] == kd && lo[r] <= ki));
            const unsigned long long km = __ballot(keep);
            if (keep) row[base + __popcll(km & ((1ull << lane) - 1ull))] = ((unsigned long long)hi[r] << 32) | lo[r];
            base += __popcll(km);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wcnt = base;   // == k
        B = __uint_as_float(kd);
    };
    // the 64 targets of one leaf, one per lane
    auto eval_leaf = [&](int leaf) {
        const int pidx = leaf * 64 + lane;   // arrays are padded with +inf to a multiple of 256
        float dx = sx[pidx] - qx, dy = sy[pidx] - qy, dz = sz[pidx] - qz;   // (points[i] - query)
        float d2 = dx * dx + (dy * dy + dz * dz);
        bool acc = pidx < n && d2 <= B;
        unsigned long long am = __ballot(acc);
        if (!am) return;
        if (wcnt + __popcll(am) > ROW) {   // the row cannot take them all (wcnt > ROW - 64 >= k)
            compact();
            acc = acc && d2 <= B;
            am = __ballot(acc);
            if (!am) return;
        }
        const int at = wcnt + __popcll(am & ((1ull << lane) - 1ull));
        if (acc) row[at] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(okey ? okey[pidx] : orig[pidx]);
        wcnt += __popcll(am);
    };

    const int own = sp >> 6;
    // Several clouds in one array (the batch's small instances, each padded to whole leaves with NaN coordinates): the walk
    // stays inside the query's own cloud - boxes of other clouds are never looked at, whatever their coordinates.
    int l_lo = 0, l_hi = n_leaf;
    if (inst_leaf) {
        int a = 0, z = n_inst;
        while (z - a > 1) { const int m = (a + z) >> 1; if (inst_leaf[m] <= own) a = m; else z = m; }
        l_lo = inst_leaf[a]; l_hi = inst_leaf[a + 1];
    }
    const int seed_lo = seed_own ? max(l_lo, own - SEED_SPAN) : 1, seed_hi = seed_own ? min(l_hi - 1, own + SEED_SPAN) : 0;
    if (seed_own) {
        // unbounded start: the query's own leaf (its 64 curve neighbours) gives the first bound, its curve-adjacent
        // leaves follow; the walk skips them
        eval_leaf(own);
        if (wcnt >= k) compact();
        for (int l = seed_lo; l <= seed_hi; ++l) if (l != own) eval_leaf(l);
    }
    // Validity is tracked explicitly (never through "+inf <= bound"): the bound itself is +inf while an unbounded
    // search has seen fewer than k points, or for an unbounded radius.
    const int t_hi = inst_leaf ? ((l_hi - 1) >> 6) + 1 : n_top;
    for (int tb = inst_leaf ? (l_lo >> 6) : 0; tb < t_hi; tb += 64) {
        const int t = tb + lane;
        const bool t_valid = t < t_hi;
        const float lbt = t_valid ? box_lower_bound(tbox, n_top, t, qp, qp) : INFINITY;
        unsigned long long tmask = __ballot(t_valid && lbt <= B);
        while (tmask) {
            const int bt = __ffsll((long long)tmask) - 1;
            tmask &= tmask - 1;
            if (__shfl(lbt, bt, 64) > B) continue;   // the bound may have dropped since the test
            const int u = (tb + bt) * 64 + lane;
            bool pending = u >= l_lo && u < l_hi && !(u >= seed_lo && u <= seed_hi);   // a leaf of this group (and cloud) not evaluated yet
            const float lbl = pending ? box_lower_bound(lbox, n_leaf, u, qp, qp) : INFINITY;
            if (BEST_FIRST && seed_own) {
                while (true) {   // nearest leaf first: the bound tightens before the far leaves are looked at
                    const bool cand = pending && lbl <= B;
                    if (!__any(cand)) break;
                    float m = cand ? lbl : INFINITY;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
                    const int bl = __ffsll((long long)__ballot(cand && lbl == m)) - 1;
                    if (lane == bl) pending = false;
                    eval_leaf((tb + bt) * 64 + bl);
                }
            } else {
                unsigned long long lmask = __ballot(pending && lbl <= B);
                while (lmask) {
                    const int bl = __ffsll((long long)lmask) - 1;
                    lmask &= lmask - 1;
                    if (__shfl(lbl, bl, 64) > B) continue;
                    eval_leaf((tb + bt) * 64 + bl);
                }
            }
        }
    }
    unsigned long long key[R];
    load_sort_row<R>(row, wcnt, lane, key);
    const int i0 = orig[sp];
    const int c = min(k, wcnt);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        if (e < c) lists[(size_t)i0 * stride + e] = key2idx ? key2idx[(int)(unsigned)key[r]] : (int)(unsigned)key[r];   // a row per query: coalesced
    }
    if (lane == 0) cnt_out[i0] = c;
}

// ------------------------------------------------------------------ normals (registration.cpp:105-130)
// One lane per point, points taken in curve order (order[t], so that the lanes of a wave gather from the same
// neighbourhood).  The k nearest neighbours come from listsA when it holds at least k entries (FPFH's radius list:
// sorted by (d2, idx), so its first k ARE the k nearest), else from listsB (kNN search).  Lists are rows per point.
__global__ __launch_bounds__(KN_BLOCK)
void k_normals_from_lists(const float* __restrict__ xyz, int n, const int* __restrict__ order, int k,
                          const int* __restrict__ listsA, int strideA, const int* __restrict__ cntA,
                          const int* __restrict__ listsB, int strideB, const int* __restrict__ cntB,
                          float* __restrict__ normals, int* __restrict__ knn_out, int knn_stride) {
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * KN_BLOCK + threadIdx.x;
    const bool live = t < n;
    const int i = order[live ? t : n - 1];
    const bool useA = listsA && cntA[i] >= k;
    const int* __restrict__ Lg = useA ? listsA + (size_t)i * strideA : listsB + (size_t)i * strideB;
    const int cnt = useA ? k : cntB[i];
    // Rows are contiguous per point: a wave reads its 64 rows coalesced (lane r of row q) into LDS, then every lane
    // walks its own row there (k <= 32; longer lists are read in place).
    __shared__ int srow[KN_BLOCK / 64][64][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool staged = k <= 32;
    if (staged) {
        const unsigned long long pbits = (unsigned long long)Lg;
        for (int q = 0; q < 64; ++q) {
            const unsigned lo32 = __shfl((unsigned)pbits, q, 64), hi32 = __shfl((unsigned)(pbits >> 32), q, 64);
            const int* row = (const int*)(((unsigned long long)hi32 << 32) | lo32);
            const int c = __shfl(cnt, q, 64);
            if (lane < c) srow[wave][q][lane] = row[lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!live) return;
    const int* L = staged ? &srow[wave][lane][0] : Lg;
    if (knn_out) for (int r = 0; r < knn_stride; ++r) knn_out[(size_t)i * knn_stride + r] = r < cnt ? L[r] : -1;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = L[r];
        cx += xyz[3 * (size_t)j]; cy += xyz[3 * (size_t)j + 1]; cz += xyz[3 * (size_t)j + 2];
    }
    const float fc = (float)cnt;
    cx /= fc; cy /= fc; cz /= fc;
    float c00 = 0.f, c10 = 0.f, c20 = 0.f, c11 = 0.f, c21 = 0.f, c22 = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = L[r];
        float dx = xyz[3 * (size_t)j] - cx, dy = xyz[3 * (size_t)j + 1] - cy, dz = xyz[3 * (size_t)j + 2] - cz;
        c00 += dx * dx; c10 += dy * dx; c20 += dz * dx; c11 += dy * dy; c21 += dz * dy; c22 += dz * dz;
    }
    c00 /= fc; c10 /= fc; c20 /= fc; c11 /= fc; c21 /= fc; c22 /= fc;
    float nx, ny, nz;
    dl::smallest_eigvec3(c00, c10, c20, c11, c21, c22, nx, ny, nz);
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    float dot = nx * (-px) + (ny * (-py) + nz * (-pz));  // normals[i].dot(-points[i])
    if (dot < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    normals[3 * (size_t)i] = nx; normals[3 * (size_t)i + 1] = ny; normals[3 * (size_t)i + 2] = nz;
}

// deficient[sp] = 1 if the radius list of the point at sorted position sp has fewer than k entries
__global__ void k_flag_deficient(const int* __restrict__ orig, const int* __restrict__ cntA, int n, int k, int* __restrict__ flag) {
    int sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp < n) flag[sp] = cntA[orig[sp]] < k ? 1 : 0;
}
__global__ void k_compact_flagged(const int* __restrict__ flag, const int* __restrict__ pos, int n, int* __restrict__ qsel) {
    int sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp < n && flag[sp]) qsel[pos[sp]] = sp;   // ascending sorted positions: the subset stays spatially coherent
}

// ------------------------------------------------------------------ FPFH (registration.cpp:133-201)
constexpr int FP_MAXNN = 100;
#ifndef FP_WAVES_VALUE
#define FP_WAVES_VALUE 4
#endif
constexpr int FP_WAVES = FP_WAVES_VALUE;   // points (waves) per workgroup of k_spfh / k_fpfh

#ifdef TDV_STUDY
// SPFH (registration.cpp:137-170), ONE WAVE PER POINT in curve order, one neighbour per lane: the pair features
// (incl. atan2f) are computed in parallel, and the histogram is counted with ballots — the CPU loop adds 1.0f
// per pair, and sums of ones are exact in any order, so counting is the same arithmetic.
__global__ __launch_bounds__(64 * FP_WAVES)
void k_spfh(const float* __restrict__ xyz, const float* __restrict__ nrm, int n, const int* __restrict__ order,
            const int* __restrict__ nbr, const int* __restrict__ nbr_cnt, float* __restrict__ spfh) {
    const int lane = threadIdx.x & 63;
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * FP_WAVES + (threadIdx.x >> 6);
    if (t >= n) return;   // wave-uniform
    const int i = __builtin_amdgcn_readfirstlane(order[t]);
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    const float ux = nrm[3 * (size_t)i], uy = nrm[3 * (size_t)i + 1], uz = nrm[3 * (size_t)i + 2];
    const int cnt = __builtin_amdgcn_readfirstlane(nbr_cnt[i]);
    // the histogram: 33 LDS counters per wave, three integer atomic adds per pair (33 ballots + lane selects per 64 pairs
    // were a quarter of the kernel's instructions; integer counts do not depend on the order)
    __shared__ int hist[FP_WAVES][64];
    int* myhist = hist[threadIdx.x >> 6];
    myhist[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int r0 = 0; r0 < cnt; r0 += 64) {
        const int r = r0 + lane;
        const int j = r < cnt ? nbr[(size_t)i * FP_MAXNN + r] : i;
        bool valid = r < cnt && j != i;
        int bin_a = 0, bin_p = 0, bin_t = 0;
        if (valid) {
            float dx = xyz[3 * (size_t)j] - px, dy = xyz[3 * (size_t)j + 1] - py, dz = xyz[3 * (size_t)j + 2] - pz;
            float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
            valid = !(dist < 1e-8f);
            if (valid) {
                float ex = dx / dist, ey = dy / dist, ez = dz / dist;
                float vx = uy * ez - uz * ey, vy = uz * ex - ux * ez, vz = ux * ey - uy * ex;   // v = u x d
                float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;   // w = u x v
                float njx = nrm[3 * (size_t)j], njy = nrm[3 * (size_t)j + 1], njz = nrm[3 * (size_t)j + 2];
                float alpha = vx * njx + (vy * njy + vz * njz);
                float phi = ux * ex + (uy * ey + uz * ez);
                float wn = wx * njx + (wy * njy + wz * njz);
                float un = ux * njx + (uy * njy + uz * njz);
                float theta = lm::atan2f_glibc(wn, un);      // glibc's atan2f, the function registration.cpp:154 calls (libm_f32.hpp)
                bin_a = min(max((int)((alpha + 1.0f) * 5.5f), 0), 10);
                bin_p = min(max((int)((phi + 1.0f) * 5.5f), 0), 10);
                bin_t = min(max((int)(((double)theta / 3.14159265358979323846 + (double)1.0f) * (double)5.5f), 0), 10);
            }
        }
        if (valid) { atomicAdd(&myhist[bin_a], 1); atomicAdd(&myhist[11 + bin_p], 1); atomicAdd(&myhist[22 + bin_t], 1); }
    }
    __builtin_amdgcn_wave_barrier();
    const int mine = myhist[lane];   // lane b < 33: the count of bin b
    const float hv = (float)mine;   // = the CPU's sum of 1.0f increments (counts stay far below 2^24)
    float sum = 0.f;
    for (int b = 0; b < 33; ++b) sum += __shfl(hv, b, 64);
    float v = hv;
    if (sum > 0.f) v /= sum;
    if (lane < 33) spfh[(size_t)i * 33 + lane] = v;
}
#endif  // TDV_STUDY

#ifdef TDV_STUDY
// FPFH (registration.cpp:176-197), ONE WAVE PER POINT in curve order: lane d < 33 owns bin d and adds
// w_r * spfh[j_r][d] for r = 0, 1, ... in list order (the CPU loop's order per bin); every step reads one 132-byte
// row coalesced, and the four points of a workgroup are spatial neighbours, so most rows come from L1/L2.
__global__ __launch_bounds__(64 * FP_WAVES)
void k_fpfh(const float* __restrict__ xyz, int n, const int* __restrict__ order, const int* __restrict__ nbr,
            const int* __restrict__ nbr_cnt, const float* __restrict__ spfh, float* __restrict__ desc,
            int* __restrict__ nbr_out /* [n][100] or null */) {
    const int lane = threadIdx.x & 63;
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * FP_WAVES + (threadIdx.x >> 6);
    if (t >= n) return;   // wave-uniform
    const int i = __builtin_amdgcn_readfirstlane(order[t]);          // wave-uniform, and known to the compiler as such:
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    const int cnt = __builtin_amdgcn_readfirstlane(nbr_cnt[i]);      // the loop conditions below stay on the scalar unit
    // neighbours r = lane and r = 64 + lane: index, weight 1/dist, or skipped (self, coincident point)
    int j[2]; float w[2]; bool use[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = h * 64 + lane;
        j[h] = r < cnt ? nbr[(size_t)i * FP_MAXNN + r] : -1;
        if (nbr_out && r < FP_MAXNN) nbr_out[(size_t)i * FP_MAXNN + r] = j[h];
        use[h] = false; w[h] = 0.f;
        if (j[h] >= 0 && j[h] != i) {
            float dx = xyz[3 * (size_t)j[h]] - px, dy = xyz[3 * (size_t)j[h] + 1] - py, dz = xyz[3 * (size_t)j[h] + 2] - pz;
            float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
            if (!(dist < 1e-8f)) { use[h] = true; w[h] = 1.0f / dist; }
        }
    }
    const int d = min(lane, 32);
    float f = spfh[(size_t)i * 33 + d];
    // Eight rows in flight per step (the gathers are latency-bound otherwise); the adds stay in list order.  The loop is
    // bound by its instruction count, not by the rows' bytes (SQ counters: 817 vector + 731 scalar instructions per wave
    // before, most of the scalar ones per-neighbour decoding): a neighbour's row offset and weight come out of the lanes
    // with one v_readlane each, the address is one vector add on a fixed base, the list's two halves are walked separately,
    // and an entry that is not to be added (self, a coincident point, the lanes past the end of the list) needs no test: its
    // weight is +0 and its row offset 0, SPFH rows are finite and non-negative, so it adds w * val = +0 to a sum that is
    // never -0 — the same bits as skipping it.
    const unsigned row_off[2] = {use[0] ? (unsigned)j[0] * 33u : 0u, use[1] ? (unsigned)j[1] * 33u : 0u};   // in floats; < 2^32 for n < 1.3e8
    constexpr int FU = 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int nh = min(64, cnt - 64 * h);            // wave-uniform
        for (int l0 = 0; l0 < nh; l0 += FU) {            // (nh <= 0: no pass)
            float val[FU], wr[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int l = l0 + u;                    // <= 63: nh <= 64 and FU divides 64
                const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)row_off[h], l) + (unsigned)d;
                wr[u] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(w[h]), l));
                val[u] = spfh[o];
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) f = f + wr[u] * val[u];
        }
    }
    float sum = 0.f;
    for (int b = 0; b < 33; ++b) sum += __shfl(f, b, 64);   // d = 0..32 in order, as the CPU loop
    float v = f;
    if (sum > 0.f) v /= sum;
    if (lane < 33) desc[(size_t)i * 33 + lane] = v;
}
#endif  // TDV_STUDY

// SPFH with TWO POINTS PER WAVE (round 3; the default): k_spfh gives a point's ~81 neighbours two passes of 64 lanes, the second
// a quarter full.  Here a wave owns two points and lays their neighbour lists end to end over its lanes (three passes for ~162
// pairs); a lane picks its point's position and normal by its place in the combined list.  Every pair's features are the same
// expressions as in k_spfh and the histograms are integer counts, so the rows are k_spfh's bit for bit.
__global__ __launch_bounds__(64 * FP_WAVES)
void k_spfh_pairs(const float* __restrict__ xyz, const float* __restrict__ nrm, int n, const int* __restrict__ order,
                  const int* __restrict__ nbr, const int* __restrict__ nbr_cnt, float* __restrict__ spfh) {
    const int lane = threadIdx.x & 63;
    const int t0 = (xcd_contiguous_block(blockIdx.x, gridDim.x) * FP_WAVES + (threadIdx.x >> 6)) * 2;
    if (t0 >= n) return;   // wave-uniform
    const bool two = t0 + 1 < n;
    const int iA = __builtin_amdgcn_readfirstlane(order[t0]), iB = __builtin_amdgcn_readfirstlane(order[two ? t0 + 1 : t0]);
    const int cA = __builtin_amdgcn_readfirstlane(nbr_cnt[iA]), cB = two ? __builtin_amdgcn_readfirstlane(nbr_cnt[iB]) : 0;
    __shared__ int hist[FP_WAVES][2][64];
    int (*myhist)[64] = hist[threadIdx.x >> 6];
    myhist[0][lane] = 0; myhist[1][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const float pAx = xyz[3 * (size_t)iA], pAy = xyz[3 * (size_t)iA + 1], pAz = xyz[3 * (size_t)iA + 2];
    const float uAx = nrm[3 * (size_t)iA], uAy = nrm[3 * (size_t)iA + 1], uAz = nrm[3 * (size_t)iA + 2];
    const float pBx = xyz[3 * (size_t)iB], pBy = xyz[3 * (size_t)iB + 1], pBz = xyz[3 * (size_t)iB + 2];
    const float uBx = nrm[3 * (size_t)iB], uBy = nrm[3 * (size_t)iB + 1], uBz = nrm[3 * (size_t)iB + 2];
    const int total = cA + cB;
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + lane;
        const bool sel = c >= cA;                                  // this lane's pair belongs to the second point
        const int i = sel ? iB : iA, r = sel ? c - cA : c;
        const float px = sel ? pBx : pAx, py = sel ? pBy : pAy, pz = sel ? pBz : pAz;
        const float ux = sel ? uBx : uAx, uy = sel ? uBy : uAy, uz = sel ? uBz : uAz;
        const int j = c < total ? nbr[(size_t)i * FP_MAXNN + r] : i;
        bool valid = c < total && j != i;
        int bin_a = 0, bin_p = 0, bin_t = 0;
        if (valid) {
            float dx = xyz[3 * (size_t)j] - px, dy = xyz[3 * (size_t)j + 1] - py, dz = xyz[3 * (size_t)j + 2] - pz;
            float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
            valid = !(dist < 1e-8f);
            if (valid) {
                float ex = dx / dist, ey = dy / dist, ez = dz / dist;
                float vx = uy * ez - uz * ey, vy = uz * ex - ux * ez, vz = ux * ey - uy * ex;   // v = u x d
                float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;   // w = u x v
                float njx = nrm[3 * (size_t)j], njy = nrm[3 * (size_t)j + 1], njz = nrm[3 * (size_t)j + 2];
                float alpha = vx * njx + (vy * njy + vz * njz);
                float phi = ux * ex + (uy * ey + uz * ez);
                float wn = wx * njx + (wy * njy + wz * njz);
                float un = ux * njx + (uy * njy + uz * njz);
                float theta = lm::atan2f_glibc(wn, un);      // glibc's atan2f, the function registration.cpp:154 calls (libm_f32.hpp)
                bin_a = min(max((int)((alpha + 1.0f) * 5.5f), 0), 10);
                bin_p = min(max((int)((phi + 1.0f) * 5.5f), 0), 10);
                bin_t = min(max((int)(((double)theta / 3.14159265358979323846 + (double)1.0f) * (double)5.5f), 0), 10);
            }
        }
        if (valid) { int* h = myhist[sel ? 1 : 0]; atomicAdd(&h[bin_a], 1); atomicAdd(&h[11 + bin_p], 1); atomicAdd(&h[22 + bin_t], 1); }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && !two) break;
        const float hv = (float)myhist[which][lane];   // lane b < 33: the count of bin b = the CPU's sum of 1.0f increments
        float sum = 0.f;
        for (int b = 0; b < 33; ++b) sum += __shfl(hv, b, 64);
        float v = hv;
        if (sum > 0.f) v /= sum;
        if (lane < 33) spfh[(size_t)(which ? iB : iA) * 33 + lane] = v;
    }
}

// FPFH with TWO POINTS PER WAVE (round 3; the default): k_fpfh keeps 33 of 64 lanes busy and is bound by issuing its ~7 instructions
// per neighbour.  Here each half of a wave owns a point - lane l of the half holds bin l, lane 0 also bin 32 - and a step adds one
// neighbour to BOTH points: the per-neighbour (row offset, weight) pairs of either point are staged in LDS (a lane reads its own
// half's entry: two addresses per wave), the two 128-byte row segments are one load.  Per bin the sum still runs over the point's
// neighbours in list order with the same expression (registration.cpp:176-197): the outputs are k_fpfh's bit for bit.
constexpr int FP2_WAVES = 4;
__global__ __launch_bounds__(64 * FP2_WAVES)
void k_fpfh_pairs(const float* __restrict__ xyz, int n, const int* __restrict__ order, const int* __restrict__ nbr,
                  const int* __restrict__ nbr_cnt, const float* __restrict__ spfh, float* __restrict__ desc,
                  int* __restrict__ nbr_out /* [n][100] or null */) {
    __shared__ uint2 s_ent[FP2_WAVES][2][128];                  // (row offset in floats, weight bits) of neighbour r of either point; r >= cnt: (0, +0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l = lane & 31;
    const int t0 = (xcd_contiguous_block(blockIdx.x, gridDim.x) * FP2_WAVES + wave) * 2;
    if (t0 >= n) return;                                          // wave-uniform
    const int t = min(t0 + half, n - 1);                          // an odd n: the last wave's second half repeats its first (computed, not written)
    const bool writes = t0 + half < n;
    const int i = order[t];
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    const int cnt = nbr_cnt[i];
    uint2* ent = s_ent[wave][half];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * 32 + l;
        const int j = r < cnt ? nbr[(size_t)i * FP_MAXNN + r] : -1;
        if (nbr_out && writes && r < FP_MAXNN) nbr_out[(size_t)i * FP_MAXNN + r] = j;
        unsigned off = 0u; float w = 0.f;                         // not to be added (self, a coincident point, past the end): weight +0 on row 0 adds +0
        if (j >= 0 && j != i) {
            const float dx = xyz[3 * (size_t)j] - px, dy = xyz[3 * (size_t)j + 1] - py, dz = xyz[3 * (size_t)j + 2] - pz;
            const float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
            if (!(dist < 1e-8f)) { off = (unsigned)j * 33u; w = 1.0f / dist; }
        }
        ent[r] = make_uint2(off, __float_as_uint(w));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float f = spfh[(size_t)i * 33 + l];
    float f32 = spfh[(size_t)i * 33 + 32];                        // (every lane of the half carries bin 32; lane 0's copy is the one written)
    const int steps = max(cnt, __shfl_xor(cnt, 32, 64));          // wave-uniform: the longer of the two lists
    constexpr int FU = 8;
    for (int r0 = 0; r0 < steps; r0 += FU) {
        float val[FU], v32[FU], wr[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const uint2 e = ent[min(r0 + u, 127)];
            wr[u] = __uint_as_float(e.y);
            val[u] = spfh[e.x + (unsigned)l];
            v32[u] = spfh[e.x + 32u];
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) { f = f + wr[u] * val[u]; f32 = f32 + wr[u] * v32[u]; }
    }
    float sum = 0.f;
    for (int b = 0; b < 32; ++b) sum += __shfl(f, half * 32 + b, 64);   // d = 0..32 in order, as the CPU loop
    sum += f32;
    const bool norm = sum > 0.f;
    if (writes) {
        desc[(size_t)i * 33 + l] = norm ? f / sum : f;
        if (l == 0) desc[(size_t)i * 33 + 32] = norm ? f32 / sum : f32;
    }
}

namespace {

struct ScanPlan { int n_pad, nt_pad, n_chunks, blocks_x; };

ScanPlan make_scan_plan(int n) {
    ScanPlan p;
    p.n_pad = (int)align_up((size_t)n, KN_BLOCK);
    p.nt_pad = (int)align_up((size_t)n, 16);
    p.n_chunks = p.nt_pad / 16;          // 16-target chunks
    p.blocks_x = p.n_pad / KN_BLOCK;
    return p;
}

// okey / key2idx (optional): ties in (d2, id) order are broken by okey[position] instead of the array index, and the lists
// receive key2idx[id] — the batch runs these stages on the voxel cloud in first-occurrence order (spatially coherent: the
// gathers of the estimators stay local) while ordering every list as the reference's container order would.
struct Sorted { float *sx, *sy, *sz; int* orig; float *lbox, *tbox; int n_leaf, n_top; const int* okey = nullptr; const int* key2idx = nullptr; };

// Morton sort of the cloud: sorted SoA coordinates (padded with +inf) and the original index of each position
int spatial_sort(tdv_ctx* ctx, const float* d_xyz, int n, const ScanPlan& p, Sorted& so) {
    hipStream_t s = ctx->stream;
    const int pad = std::max(p.n_pad, (int)align_up((size_t)p.nt_pad, 16));   // multiple of 256 >= n: covers 8- and 16-target chunks
    float* soa; float *part, *bbox; int *bucket_of, *hist, *cursor, *start, *d_total;
    TDV_TRY(ws_alloc(ctx, (size_t)3 * pad, &soa));
    TDV_TRY(ws_alloc(ctx, (size_t)pad, &so.orig));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &bucket_of));
    // a few buckets per point for a surface sample (most cells of the grid are empty)
    static const int per_point = study_env("TDV_MORTON_BUCKETS_PER_POINT") ? atoi(study_env("TDV_MORTON_BUCKETS_PER_POINT")) : 8;
    int bits = 12;
    while (bits < MORTON_MAX_BUCKET_BITS && (1ll << bits) < (long long)per_point * n) bits += 1;
    const int nbuckets = 1 << bits;
    TDV_TRY(ws_alloc(ctx, (size_t)2 * nbuckets, &hist));          // hist | cursor: one memset
    cursor = hist + nbuckets;
    TDV_TRY(ws_alloc(ctx, (size_t)nbuckets, &start));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    const int bblocks = std::min(1024, (n + 255) / 256);
    TDV_TRY(ws_alloc(ctx, (size_t)bblocks * 6, &part));
    TDV_TRY(ws_alloc(ctx, 6, &bbox));
    so.sx = soa; so.sy = soa + pad; so.sz = soa + 2 * (size_t)pad;
    TDV_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)2 * nbuckets * 4, s));
    k_bbox_partial<<<bblocks, 256, 0, s>>>(d_xyz, n, part);
    k_bbox_final<<<1, 64, 0, s>>>(part, bblocks, bbox);
    k_morton_hist<<<(n + 255) / 256, 256, 0, s>>>(d_xyz, n, bbox, bits, bucket_of, hist);
    TDV_TRY(exclusive_scan_dev(ctx, hist, nbuckets, start, d_total));
    k_morton_scatter<<<(pad + 255) / 256, 256, 0, s>>>(d_xyz, n, pad, bucket_of, start, cursor, so.sx, so.sy, so.sz, so.orig);
    // bounding boxes of the 64-point leaves and of the 4096-point groups of 64 leaves (exact pruning of the searches)
    so.n_leaf = (int)(align_up((size_t)n, 64) / 64);
    so.n_top = (so.n_leaf + 63) / 64;
    TDV_TRY(ws_alloc(ctx, (size_t)6 * so.n_leaf, &so.lbox));
    TDV_TRY(ws_alloc(ctx, (size_t)6 * so.n_top, &so.tbox));
    k_leaf_boxes<<<(so.n_leaf + 3) / 4, 256, 0, s>>>(so.sx, so.sy, so.sz, so.n_leaf, so.lbox);
    k_top_boxes<<<so.n_top, 64, 0, s>>>(so.lbox, so.n_leaf, so.n_top, so.tbox);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

}  // namespace

int spatial_sort_cloud(tdv_ctx* ctx, const float* d_xyz, int n, SortedCloud& out) {
    const ScanPlan p = make_scan_plan(n);
    Sorted so;
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    out.sx = so.sx; out.sy = so.sy; out.sz = so.sz; out.orig = so.orig; out.lbox = so.lbox; out.tbox = so.tbox;
    out.n = n; out.pad = std::max(p.n_pad, (int)align_up((size_t)p.nt_pad, 16));
    out.n_leaf = so.n_leaf; out.n_top = so.n_top;
    return TDV_OK;
}

namespace {

// One k_query_wave launch: lists[original index * k + r] / cnt[original index] for all n queries
// (qsel == nullptr) or for the nsel sorted positions in qsel (device).
int query_to_lists(tdv_ctx* ctx, const Sorted& so, int n, const ScanPlan& p, int k, const float* bound, float bound0, int seed_own,
                   int timer, const int* qsel, int nsel, int* lists, int* cnt) {
    const int nqq = qsel ? nsel : n;
    if (nqq <= 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    const unsigned grid = (unsigned)((nqq + QW_WAVES - 1) / QW_WAVES);
    ScopedTimer tm(ctx, timer);
#define TDV_QW(RR) k_query_wave<RR, QW_SEED_SPAN, QW_BEST_FIRST><<<grid, 64 * QW_WAVES, 0, s>>>(so.sx, so.sy, so.sz, so.orig, n, so.n_leaf, so.lbox, so.n_top, so.tbox, \
                                                            qsel, nqq, bound, bound0, seed_own, k, k, lists, cnt, so.okey, so.key2idx)
    if (k <= 64) TDV_QW(2);
    else if (k <= 192) TDV_QW(4);
    else if (k <= 448) TDV_QW(8);
    else return TDV_ERR_BAD_ARG;
#undef TDV_QW
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

// radius search: every target with d2 <= r2, the first k in (d2, idx) order
int radius_to_lists(tdv_ctx* ctx, const Sorted& so, int n, const ScanPlan& p, int k, float r2, int* lists, int* cnt) {
    return query_to_lists(ctx, so, n, p, k, nullptr, r2, 0, TDV_TIMER_RADIUS, nullptr, 0, lists, cnt);
}

// exact kNN lists of all queries (qsel == nullptr) or of the subset qsel: unbounded start from the query's own run
// (an initial bound from a 768-point window in a separate kernel was measured slower at every size)
int knn_to_lists(tdv_ctx* ctx, const Sorted& so, int n, const ScanPlan& p, int k, const int* qsel, int nsel, int* lists, int* cnt) {
    return query_to_lists(ctx, so, n, p, k, nullptr, INFINITY, 1, TDV_TIMER_KNN, qsel, nsel, lists, cnt);
}

}  // namespace

int estimate_normals_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float* d_normals, int* d_knn) {
    if (!ctx || n < 0 || k <= 0 || k > 255 || (n > 0 && (!d_xyz || !d_normals))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const int kk = std::min(k, n);  // std::min(k, dists.size()), registration.cpp:74
    const ScanPlan p = make_scan_plan(n);
    Sorted so; int *lists, *cnt;
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    TDV_TRY(ws_alloc(ctx, (size_t)kk * p.n_pad, &lists));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cnt));
    TDV_TRY(knn_to_lists(ctx, so, n, p, kk, nullptr, 0, lists, cnt));
    k_normals_from_lists<<<p.blocks_x, KN_BLOCK, 0, ctx->stream>>>(d_xyz, n, so.orig, kk, nullptr, 0, nullptr, lists, kk, cnt, d_normals, d_knn, k);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

namespace {
int fpfh_from_lists(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, const ScanPlan& p, const int* order, const int* nbr, const int* cnt,
                    float* d_desc, int* d_nbr, int* d_nbr_cnt) {
    if ((size_t)n * 33 > 0xffffffffull) return TDV_ERR_BAD_ARG;   // k_fpfh addresses SPFH rows with 32-bit float offsets (130 M points)
    float* spfh;
    TDV_TRY(ws_alloc(ctx, (size_t)n * 33, &spfh));
    hipStream_t s = ctx->stream;
#ifdef TDV_STUDY
    const char* pairs_env = study_env("TDV_FPFH_PAIRS");       // study build, read per call: 0 = one point per wave (k_spfh, k_fpfh: measured slower, 0.94-0.99 vs 0.87-0.92 ms at 200k)
    if (pairs_env && atoi(pairs_env) == 0) {
        k_spfh<<<(n + FP_WAVES - 1) / FP_WAVES, 64 * FP_WAVES, 0, s>>>(d_xyz, d_normals, n, order, nbr, cnt, spfh);
        k_fpfh<<<(n + FP_WAVES - 1) / FP_WAVES, 64 * FP_WAVES, 0, s>>>(d_xyz, n, order, nbr, cnt, spfh, d_desc, d_nbr);
    } else
#endif
    {
        k_spfh_pairs<<<(n + 2 * FP_WAVES - 1) / (2 * FP_WAVES), 64 * FP_WAVES, 0, s>>>(d_xyz, d_normals, n, order, nbr, cnt, spfh);
        k_fpfh_pairs<<<(n + 2 * FP2_WAVES - 1) / (2 * FP2_WAVES), 64 * FP2_WAVES, 0, s>>>(d_xyz, n, order, nbr, cnt, spfh, d_desc, d_nbr);
    }
    TDV_CHECK_LAUNCH(ctx);
    if (d_nbr_cnt) TDV_HIP(ctx, hipMemcpyAsync(d_nbr_cnt, cnt, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return TDV_OK;
}
}  // namespace

int compute_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, float radius,
                     float* d_desc, int* d_nbr, int* d_nbr_cnt) {
    if (!ctx || n < 0 || (n > 0 && (!d_xyz || !d_normals || !d_desc))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const float r2 = radius * radius;  // registration.cpp:89
    const ScanPlan p = make_scan_plan(n);
    Sorted so; int *nbr, *cnt;
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    TDV_TRY(ws_alloc(ctx, (size_t)FP_MAXNN * p.n_pad, &nbr));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cnt));
    TDV_TRY(radius_to_lists(ctx, so, n, p, FP_MAXNN, r2, nbr, cnt));
    return fpfh_from_lists(ctx, d_xyz, d_normals, n, p, so.orig, nbr, cnt, d_desc, d_nbr, d_nbr_cnt);
}

// estimateNormals(k) followed by computeFPFH(radius) on the same cloud (src/pipeline.cpp:93-95), sharing one
// spatial sort and ONE full scan: the radius lists are sorted by (d2, idx), so wherever a point has >= k
// neighbours in radius its k nearest neighbours are the first k entries; only the deficient points (isolated
// points, silhouette edges) go through a kNN scan, as a subset.  Results are identical to the two separate calls.
__global__ void k_gather_ids(const int* __restrict__ orig, const int* __restrict__ ids, int n, int n_pad, int* __restrict__ okey) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pad) okey[p] = p < n ? ids[orig[p]] : INT_MAX;
}

int normals_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float radius, float* d_normals, float* d_desc,
                     const int* d_tie_ids, const int* d_tie_ids_inv) {
    if (!ctx || n < 0 || k <= 0 || k > 255 || (n > 0 && (!d_xyz || !d_normals || !d_desc))) return TDV_ERR_BAD_ARG;
    if ((d_tie_ids == nullptr) != (d_tie_ids_inv == nullptr)) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const int kk = std::min(k, n);
    if (kk > FP_MAXNN) {  // a radius list (cap 100) cannot serve as the kNN list: the two plain calls
        if (d_tie_ids) return TDV_ERR_BAD_ARG;   // (the batch never gets here: its k is the reference's 30)
        TDV_TRY(estimate_normals_dev(ctx, d_xyz, n, k, d_normals, nullptr));
        return compute_fpfh_dev(ctx, d_xyz, d_normals, n, radius, d_desc, nullptr, nullptr);
    }
    const float r2 = radius * radius;
    const ScanPlan p = make_scan_plan(n);
    hipStream_t s = ctx->stream;
    Sorted so; int *nbr, *cnt, *flag, *pos, *qsel, *d_total, *listsK, *cntK;
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    if (d_tie_ids) {
        const int pad = std::max(p.n_pad, (int)align_up((size_t)p.nt_pad, 16));
        int* okey;
        TDV_TRY(ws_alloc(ctx, (size_t)pad, &okey));
        k_gather_ids<<<(pad + 255) / 256, 256, 0, s>>>(so.orig, d_tie_ids, n, pad, okey);
        so.okey = okey; so.key2idx = d_tie_ids_inv;
    }
    TDV_TRY(ws_alloc(ctx, (size_t)FP_MAXNN * p.n_pad, &nbr));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cnt));
    TDV_TRY(radius_to_lists(ctx, so, n, p, FP_MAXNN, r2, nbr, cnt));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &flag));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &pos));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &qsel));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(ws_alloc(ctx, (size_t)kk * p.n_pad, &listsK));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cntK));
    k_flag_deficient<<<(n + 255) / 256, 256, 0, s>>>(so.orig, cnt, n, kk, flag);
    TDV_TRY(exclusive_scan_dev(ctx, flag, n, pos, d_total));
    k_compact_flagged<<<(n + 255) / 256, 256, 0, s>>>(flag, pos, n, qsel);
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, 64));
    int* h_total = reinterpret_cast<int*>(ctx->pin);
    TDV_HIP(ctx, hipMemcpyAsync(h_total, d_total, 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    const int nsel = *h_total;
    if (getenv("TDV_DEBUG")) fprintf(stderr, "[tdv] normals_fpfh: n=%d deficient=%d (k=%d)\n", n, nsel, kk);
    TDV_TRY(knn_to_lists(ctx, so, n, p, kk, qsel, nsel, listsK, cntK));
    k_normals_from_lists<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, so.orig, kk, nbr, FP_MAXNN, cnt, listsK, kk, cntK, d_normals, nullptr, 0);
    TDV_CHECK_LAUNCH(ctx);
    return fpfh_from_lists(ctx, d_xyz, d_normals, n, p, so.orig, nbr, cnt, d_desc, nullptr, nullptr);
}


// ---- estimateNormals + computeFPFH for MANY SMALL CLOUDS in one set of launches (round 3) ---------------------------------
// The batch's small instances (config C5: 1,024 clouds of ~400 voxels) are bound by the rate at which the host can issue their
// ~16 launches each, not by the kernels.  Here all clouds sit in ONE array - cloud b padded to whole 64-point leaves with NaN
// coordinates (a NaN never passes `d2 <= bound` and is ignored by the boxes' min / max) - and k_query_wave keeps every query
// inside its own cloud's leaves (inst_leaf).  No Morton order: for a few hundred points in image order the walk visits every
// leaf of the cloud anyway.  Lists, normals and descriptors are those of normals_fpfh_dev called per cloud, bit for bit: the
// searches are exact, the lists ordered by (d2, tie id), the estimators the same kernels on global point indices.
__global__ void k_batch_layout(const float* __restrict__ xyz, const int* __restrict__ voff, const int* __restrict__ ppos, int n_inst, int total_pos,
                               const int* __restrict__ tie_ids, float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
                               int* __restrict__ orig, int* __restrict__ okey, int* __restrict__ pos_of_point, int* __restrict__ iota) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= total_pos) return;
    int a = 0, z = n_inst;
    while (z - a > 1) { const int m = (a + z) >> 1; if (ppos[m] <= pos) a = m; else z = m; }
    const int e = pos - ppos[a], v0 = voff[a], v = voff[a + 1] - v0;
    if (pos < ppos[n_inst] && e < v) {
        const size_t P = (size_t)v0 + e;
        sx[pos] = xyz[3 * P]; sy[pos] = xyz[3 * P + 1]; sz[pos] = xyz[3 * P + 2];
        orig[pos] = (int)P; pos_of_point[P] = pos; iota[P] = (int)P;
        if (okey) okey[pos] = tie_ids[P];
    } else {
        const float q = __builtin_nanf("");
        sx[pos] = q; sy[pos] = q; sz[pos] = q; orig[pos] = INT_MAX;
        if (okey) okey[pos] = INT_MAX;
    }
}
__global__ void k_batch_flag_deficient(const int* __restrict__ cnt, int n, int k, int* __restrict__ flag) {
    const int P = blockIdx.x * blockDim.x + threadIdx.x;
    if (P < n) flag[P] = cnt[P] < k ? 1 : 0;
}
__global__ void k_batch_compact_flagged(const int* __restrict__ flag, const int* __restrict__ pos, const int* __restrict__ pos_of_point, int n, int* __restrict__ qsel) {
    const int P = blockIdx.x * blockDim.x + threadIdx.x;
    if (P < n && flag[P]) qsel[pos[P]] = pos_of_point[P];
}

// d_xyz: all clouds back to back (cloud b = points [h_voff[b], h_voff[b + 1]), every cloud with at least k points); d_voff: the
// same offsets on the device.  d_tie_ids / d_tie_ids_inv (optional, both or neither), GLOBAL: neighbour lists are ordered by
// (d2, d_tie_ids[point]) and receive d_tie_ids_inv[id] - ids of one cloud must be ordered as the wanted positions inside it.
int normals_fpfh_batch_dev(tdv_ctx* ctx, const float* d_xyz, const int* h_voff, const int* d_voff, int n_clouds, int k, float radius,
                           float* d_normals, float* d_desc, const int* d_tie_ids, const int* d_tie_ids_inv) {
    if (!ctx || !d_xyz || !h_voff || !d_voff || n_clouds < 1 || k <= 0 || k > FP_MAXNN || !d_normals || !d_desc) return TDV_ERR_BAD_ARG;
    if ((d_tie_ids == nullptr) != (d_tie_ids_inv == nullptr)) return TDV_ERR_BAD_ARG;
    const int n = h_voff[n_clouds];
    if (n == 0) return TDV_OK;
    std::vector<int> ppos((size_t)n_clouds + 1, 0), leaf((size_t)n_clouds + 1, 0);
    for (int b = 0; b < n_clouds; ++b) {
        const int v = h_voff[b + 1] - h_voff[b];
        if (v > 0 && v < k) return TDV_ERR_BAD_ARG;          // (min(k, n) of registration.cpp:74 would differ per cloud: the caller takes the per-cloud path)
        ppos[b + 1] = ppos[b] + (int)align_up((size_t)v, 64);
        leaf[b + 1] = ppos[b + 1] / 64;
    }
    const int total_pos = (int)align_up((size_t)ppos[n_clouds], 256);
    hipStream_t s = ctx->stream;
    const float r2 = radius * radius;  // registration.cpp:89
    float* soa; int *orig, *okey = nullptr, *pos_of_point, *iota, *d_ppos, *d_leaf;
    TDV_TRY(ws_alloc(ctx, (size_t)3 * total_pos, &soa));
    TDV_TRY(ws_alloc(ctx, (size_t)total_pos, &orig));
    if (d_tie_ids) TDV_TRY(ws_alloc(ctx, (size_t)total_pos, &okey));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &pos_of_point));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &iota));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds + 1, &d_ppos));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds + 1, &d_leaf));
    TDV_HIP(ctx, hipMemcpyAsync(d_ppos, ppos.data(), ((size_t)n_clouds + 1) * 4, hipMemcpyHostToDevice, s));
    TDV_HIP(ctx, hipMemcpyAsync(d_leaf, leaf.data(), ((size_t)n_clouds + 1) * 4, hipMemcpyHostToDevice, s));
    Sorted so;
    so.sx = soa; so.sy = soa + total_pos; so.sz = soa + 2 * (size_t)total_pos; so.orig = orig;
    k_batch_layout<<<(total_pos + 255) / 256, 256, 0, s>>>(d_xyz, d_voff, d_ppos, n_clouds, total_pos, d_tie_ids, so.sx, so.sy, so.sz, orig, okey, pos_of_point, iota);
    so.n_leaf = total_pos / 64; so.n_top = (so.n_leaf + 63) / 64;
    TDV_TRY(ws_alloc(ctx, (size_t)6 * so.n_leaf, &so.lbox));
    TDV_TRY(ws_alloc(ctx, (size_t)6 * so.n_top, &so.tbox));
    k_leaf_boxes<<<(so.n_leaf + 3) / 4, 256, 0, s>>>(so.sx, so.sy, so.sz, so.n_leaf, so.lbox);
    k_top_boxes<<<so.n_top, 64, 0, s>>>(so.lbox, so.n_leaf, so.n_top, so.tbox);
    TDV_CHECK_LAUNCH(ctx);
    if (d_tie_ids) { so.okey = okey; so.key2idx = d_tie_ids_inv; }
    const int n_pad = (int)align_up((size_t)n, KN_BLOCK);
    int *nbr, *cnt, *flag, *pos, *qsel, *d_total, *listsK, *cntK;
    TDV_TRY(ws_alloc(ctx, (size_t)FP_MAXNN * n_pad, &nbr));
    TDV_TRY(ws_alloc(ctx, (size_t)n_pad, &cnt));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &flag));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &pos));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &qsel));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(ws_alloc(ctx, (size_t)k * n_pad, &listsK));
    TDV_TRY(ws_alloc(ctx, (size_t)n_pad, &cntK));
    auto query = [&](int kk, float bound0, int seed_own, int timer, const int* sel, int nsel, int* lists, int* counts) -> int {
        if (nsel <= 0) return TDV_OK;
        const unsigned grid = (unsigned)((nsel + QW_WAVES - 1) / QW_WAVES);
        ScopedTimer tm(ctx, timer);
#define TDV_QWB(RR) k_query_wave<RR, QW_SEED_SPAN, QW_BEST_FIRST><<<grid, 64 * QW_WAVES, 0, s>>>(so.sx, so.sy, so.sz, so.orig, total_pos, so.n_leaf, so.lbox, so.n_top, so.tbox, \
                                                            sel, nsel, nullptr, bound0, seed_own, kk, kk, lists, counts, so.okey, so.key2idx, d_leaf, n_clouds)
        if (kk <= 64) TDV_QWB(2); else TDV_QWB(4);
#undef TDV_QWB
        TDV_CHECK_LAUNCH(ctx);
        return TDV_OK;
    };
    TDV_TRY(query(FP_MAXNN, r2, 0, TDV_TIMER_RADIUS, pos_of_point, n, nbr, cnt));                       // radius lists of every point
    k_batch_flag_deficient<<<(n + 255) / 256, 256, 0, s>>>(cnt, n, k, flag);
    TDV_TRY(exclusive_scan_dev(ctx, flag, n, pos, d_total));
    k_batch_compact_flagged<<<(n + 255) / 256, 256, 0, s>>>(flag, pos, pos_of_point, n, qsel);
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, 64));
    int* h_total = reinterpret_cast<int*>(ctx->pin);
    TDV_HIP(ctx, hipMemcpyAsync(h_total, d_total, 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));                                                              // (also: ppos / leaf are host temporaries)
    TDV_TRY(query(k, INFINITY, 1, TDV_TIMER_KNN, qsel, *h_total, listsK, cntK));                         // kNN lists of the points with fewer than k in radius
    k_normals_from_lists<<<n_pad / KN_BLOCK, KN_BLOCK, 0, s>>>(d_xyz, n, iota, k, nbr, FP_MAXNN, cnt, listsK, k, cntK, d_normals, nullptr, 0);
    TDV_CHECK_LAUNCH(ctx);
    ScanPlan p = make_scan_plan(n);
    return fpfh_from_lists(ctx, d_xyz, d_normals, n, p, iota, nbr, cnt, d_desc, nullptr, nullptr);
}

}  // namespace tdv
