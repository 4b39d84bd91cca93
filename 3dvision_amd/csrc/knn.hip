// Exact brute-force neighbour searches on gfx950 and the per-point estimators built on them.
//
// Replaces (no GPU entry point exists in the reference; src/pipeline.cpp:93-95 calls the CPU statics):
//   findKNN + Registration::estimateNormals   /root/reference/src/registration.cpp:63-81, :105-130
//   findRadiusNN + Registration::computeFPFH  /root/reference/src/registration.cpp:83-102, :133-201
//
// Every query still evaluates its distance to EVERY point (the reference's O(N^2) scan, 8 VALU ops per pair,
// no FMA: d2 = dx*dx + (dy*dy + dz*dz)); what is engineered is what surrounds the scan, so that selection work
// nearly vanishes and the scan itself has the shape of the ICP nearest-neighbour kernel:
//   0. the cloud is sorted along a 30-bit Morton curve (bitonic sort shared with voxel.hip): the 64 queries of
//      a wave are spatial neighbours, and so are the targets of a chunk;
//   kNN (normals, k <= 32) — two phases, no selection state inside the scan:
//   A. k_window_bound : an upper bound on each query's k-th neighbour distance from a 768-point window around
//                       its own curve position (a subset of the cloud, so the bound can only be too large);
//   B. k_collect_scan : the full scan with that FIXED bound — two queries per lane in VGPRs, 16 targets per
//                       wave-uniform s_load step, min3 tree + one compare per chunk; the rare hits are appended
//                       to the query's candidate row (integer atomic slot counter; row order is irrelevant);
//   C. k_select_topk  : exact top-k of the row in the reference's (d2, index) order.  A row that overflowed still
//                       holds >= k real candidates, whose k-th distance is a tighter valid bound: those few
//                       queries repeat B/C as a subset until none overflows; exact ties beyond the row size end
//                       in the streaming kernel below.
//   radius search (FPFH, cap 100) and the kNN fallback — streaming selection: chunks are visited INSIDE-OUT from
//   the workgroup's own curve position (splits take interleaved positions), so the per-lane bound is tight almost
//   immediately and >99 % of chunks take the 70-instruction fast path; candidates queue per lane (8 entries) and
//   are merged into a sorted per-lane list — in VGPRs with static indexing (k <= 32) or in global memory by a
//   rank-merge (larger k).
// Results are independent of the scan order and of the atomic arrival order; lists, normals and descriptors are
// bit-identical to the CPU code's (tests/test_gpu_features.py, incl. massive exact ties).
// In the batched chain normals_fpfh_dev shares ONE radius scan between normals and FPFH: a radius list is sorted by
// (d2, idx), so its first k entries are the k nearest neighbours wherever it holds >= k; only the deficient points
// go through A-C as a subset.
// Per-point estimators run one lane per point with sequential sums in neighbour order, i.e. the same f32
// expression trees as the CPU loops; atan2 is evaluated in f64 and rounded once.
#include "tdv_internal.hpp"
#include "device_linalg.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <algorithm>
#include <cstdlib>

namespace tdv {

constexpr int KN_BLOCK = 256;
constexpr int KN_CH = 8;
constexpr int KN_PB = 8;
constexpr int KN_MAXSPLIT = 8;

__device__ __forceinline__ bool lex_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

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
__global__ void k_morton_records(const float* __restrict__ xyz, int n, int n_pow2, const float* __restrict__ bbox, uint4* __restrict__ rec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    if (i >= n) { rec[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu); return; }
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
    rec[i] = make_uint4(code, 0u, 0u, (unsigned)i);
}
__global__ void k_gather_sorted(const float* __restrict__ xyz, const uint4* __restrict__ rec, int n, int n_pad,
                                float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz, int* __restrict__ orig) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    if (i < n) {
        unsigned o = rec[i].w;
        sx[i] = xyz[3 * (size_t)o]; sy[i] = xyz[3 * (size_t)o + 1]; sz[i] = xyz[3 * (size_t)o + 2]; orig[i] = (int)o;
    } else { sx[i] = INFINITY; sy[i] = INFINITY; sz[i] = INFINITY; orig[i] = INT_MAX; }
}

// Bounding boxes of every 16-target chunk and of every super-chunk of 16 chunks (256 targets) of the sorted cloud.
// box layout: 6 arrays [minx|miny|minz|maxx|maxy|maxz][count]
__global__ void k_chunk_boxes(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                              int n_chunks, float* __restrict__ cb) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int t = 0; t < 16; ++t) {
        float v[3] = {sx[c * 16 + t], sy[c * 16 + t], sz[c * 16 + t]};
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], v[a]); mx[a] = fmaxf(mx[a], v[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { cb[(size_t)a * n_chunks + c] = mn[a]; cb[(size_t)(3 + a) * n_chunks + c] = mx[a]; }
}
__global__ void k_super_boxes(const float* __restrict__ cb, int n_chunks, int n_super, float* __restrict__ sb) {
    int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_super) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = u * 16; c < min(n_chunks, u * 16 + 16); ++c)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], cb[(size_t)a * n_chunks + c]); mx[a] = fmaxf(mx[a], cb[(size_t)(3 + a) * n_chunks + c]); }
#pragma unroll
    for (int a = 0; a < 3; ++a) { sb[(size_t)a * n_super + u] = mn[a]; sb[(size_t)(3 + a) * n_super + u] = mx[a]; }
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

// chunk visited at position v of the inside-out order centred at chunk cc (bijection onto [0, n_chunks))
__device__ __forceinline__ int visit_chunk(int v, int cc, int n_chunks) {
    const int L = cc, R = n_chunks - 1 - cc;
    const int m = min(L, R);
    if (v <= 2 * m) { int k = (v + 1) >> 1; return (v & 1) ? cc + k : cc - k; }
    return R > L ? cc + (v - m) : cc - (v - m);
}

// ------------------------------------------------------------------ register-resident top-k (k <= 32)
template <int K>
__device__ __forceinline__ void reg_insert(float (&Ld)[K], int (&Li)[K], float nd, int ni) {
    bool lt_cur = lex_less(nd, ni, Ld[K - 1], Li[K - 1]);
#pragma unroll
    for (int e = K - 1; e >= 1; --e) {
        bool lt_prev = lex_less(nd, ni, Ld[e - 1], Li[e - 1]);
        float d_keep = lt_cur ? nd : Ld[e];
        int i_keep = lt_cur ? ni : Li[e];
        Ld[e] = lt_prev ? Ld[e - 1] : d_keep;
        Li[e] = lt_prev ? Li[e - 1] : i_keep;
        lt_cur = lt_prev;
    }
    Ld[0] = lt_cur ? nd : Ld[0];
    Li[0] = lt_cur ? ni : Li[0];
}

template <int K>
__device__ __forceinline__ void reg_merge(float (&Ld)[K], int (&Li)[K], float (&pd)[KN_PB], int (&pi)[KN_PB],
                                          int k, int& cnt, int& pcnt, float bound0, float& bound) {
    int maxp = pcnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxp = max(maxp, __shfl_xor(maxp, off, 64));
#pragma unroll
    for (int s = 0; s < KN_PB; ++s) {
        if (s < maxp) {  // wave-uniform
            float nd = s < pcnt ? pd[s] : INFINITY;
            int ni = s < pcnt ? pi[s] : INT_MAX;
            reg_insert<K>(Ld, Li, nd, ni);
        }
    }
    cnt = min(k, cnt + pcnt);
    pcnt = 0;
    if (k < K) {
#pragma unroll
        for (int e = 0; e < K; ++e) if (e >= k) { Ld[e] = INFINITY; Li[e] = INT_MAX; }
    }
    float kth = Ld[K - 1];
    if (k < K) {
#pragma unroll
        for (int e = 0; e < K; ++e) if (e == k - 1) kth = Ld[e];
    }
    // inclusive bound: an equal-d2 candidate with a smaller original index must still get in
    bound = (cnt == k) ? fminf(bound0, kth) : bound0;
}

// out lists: element e of sorted query position i in split s at [(s*k + e) * nq_pad + i]; counts at [s*nq_pad + i]
template <int K>
__global__ __launch_bounds__(KN_BLOCK, 4)   // 4 waves per SIMD: keeps the K = 30 instance within 128 VGPRs
void k_topk_scan_reg(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                     const int* __restrict__ orig, int nq, int nq_pad, int n_chunks16, int n_super, int nsplit, int k, float bound0,
                     const float* __restrict__ cbox, const float* __restrict__ sbox, int prune,
                     const int* __restrict__ qsel, int nsel,   // optional: the queries are sorted positions qsel[0..nsel)
                     float* __restrict__ out_d, int* __restrict__ out_i, int* __restrict__ out_cnt) {
    const int split = blockIdx.y;
    const int qi = blockIdx.x * KN_BLOCK + threadIdx.x;
    const int nqq = qsel ? nsel : nq;                       // number of queries of this launch
    const int qslot = min(qi, nqq - 1);
    const int qc = qsel ? qsel[qslot] : qslot;               // sorted position of this lane's query
    const float qx = sx[qc], qy = sy[qc], qz = sz[qc];
    const int mid = min(blockIdx.x * KN_BLOCK + KN_BLOCK / 2, nqq - 1);
    const int cc = min((qsel ? qsel[mid] : mid) / 256, n_super - 1);   // the workgroup's own super-chunk
    float Ld[K]; int Li[K];
#pragma unroll
    for (int e = 0; e < K; ++e) { Ld[e] = INFINITY; Li[e] = INT_MAX; }
    float pd[KN_PB]; int pi[KN_PB];
#pragma unroll
    for (int s = 0; s < KN_PB; ++s) { pd[s] = INFINITY; pi[s] = INT_MAX; }
    int cnt = 0, pcnt = 0;
    float bound = bound0;
    // wave query box and the wave's largest bound (refreshed after every merge) for the exact box pruning
    float qmin[3] = {qx, qy, qz}, qmax[3] = {qx, qy, qz};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { qmin[a] = fminf(qmin[a], __shfl_xor(qmin[a], off, 64)); qmax[a] = fmaxf(qmax[a], __shfl_xor(qmax[a], off, 64)); }
    float Bmax = bound0;
    for (int vs = split; vs < n_super; vs += nsplit) {
        const int u = visit_chunk(vs, cc, n_super);   // super-chunks (256 targets) inside-out from the workgroup's own
        if (prune && !__any(box_lower_bound(sbox, n_super, u, qmin, qmax) <= Bmax)) continue;
        const int c16_end = min(n_chunks16, u * 16 + 16);
        for (int c16 = u * 16; c16 < c16_end; ++c16) {
            if (prune && !__any(box_lower_bound(cbox, n_chunks16, c16, qmin, qmax) <= Bmax)) continue;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int j = c16 * 16 + h * KN_CH;
                float d2[KN_CH];
#pragma unroll
                for (int t = 0; t < KN_CH; ++t) {
                    float dx = sx[j + t] - qx, dy = sy[j + t] - qy, dz = sz[j + t] - qz;   // (points[i] - query)
                    d2[t] = dx * dx + (dy * dy + dz * dz);
                }
                float m = fminf(fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])), fminf(fminf(d2[4], d2[5]), fminf(d2[6], d2[7])));
                if (!__any(m <= bound)) continue;
                // slow path: make room ONCE per chunk (a single merge site keeps the 8 appends statically indexed)
                int nacc = 0;
#pragma unroll
                for (int t = 0; t < KN_CH; ++t) nacc += (d2[t] <= bound) ? 1 : 0;
                if (__any(pcnt + nacc > KN_PB)) {
                    reg_merge<K>(Ld, Li, pd, pi, k, cnt, pcnt, bound0, bound);
                    Bmax = bound;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) Bmax = fmaxf(Bmax, __shfl_xor(Bmax, off, 64));
                }
#pragma unroll
                for (int t = 0; t < KN_CH; ++t) {
                    // padding targets (j + t >= nq, d2 = +inf) are never candidates; the bound only tightens, so what
                    // is refused now could never enter the list
                    const bool acc = (j + t < nq) && d2[t] <= bound;
                    if (__any(acc)) {
                        const int oi = orig[j + t];
#pragma unroll
                        for (int s = KN_PB - 1; s >= 1; --s) { pd[s] = acc ? pd[s - 1] : pd[s]; pi[s] = acc ? pi[s - 1] : pi[s]; }
                        pd[0] = acc ? d2[t] : pd[0];
                        pi[0] = acc ? oi : pi[0];
                        pcnt += acc ? 1 : 0;
                    }
                }
            }
        }
    }
    if (__any(pcnt > 0)) reg_merge<K>(Ld, Li, pd, pi, k, cnt, pcnt, bound0, bound);
    if (qi < nq_pad) {
        out_cnt[(size_t)split * nq_pad + qi] = cnt;
#pragma unroll
        for (int e = 0; e < K; ++e) {
            if (e < k) {
                out_d[((size_t)split * k + e) * nq_pad + qi] = Ld[e];
                out_i[((size_t)split * k + e) * nq_pad + qi] = Li[e];
            }
        }
    }
}

// ------------------------------------------------------------------ two-phase kNN: window bound -> collect scan -> select
// Phase A: an upper bound on each query's k-th neighbour distance from a small window around its own position
// on the Morton curve (a subset of the cloud, so its k-th smallest distance can only be >= the true one).
// One lane per query, per-lane (vector) loads, a register-resident sorted list of distances only.
template <int K>
__global__ __launch_bounds__(KN_BLOCK)
void k_window_bound(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                    int n, const int* __restrict__ qsel, int nqq, int k, int half_window, float* __restrict__ bound) {
    const int slot = blockIdx.x * KN_BLOCK + threadIdx.x;
    const int sl = min(slot, nqq - 1);
    const int sp = qsel ? qsel[sl] : sl;
    const float qx = sx[sp], qy = sy[sp], qz = sz[sp];
    int lo = max(0, sp - half_window), hi = min(n, sp + half_window);
    if (hi - lo < 2 * half_window) { if (lo == 0) hi = min(n, 2 * half_window); else lo = max(0, n - 2 * half_window); }
    float Ld[K];
#pragma unroll
    for (int e = 0; e < K; ++e) Ld[e] = INFINITY;
    const int len = hi - lo;
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
    for (int it = 0; it < maxlen; ++it) {
        const bool in = it < len;
        const int p = in ? lo + it : sp;
        float dx = sx[p] - qx, dy = sy[p] - qy, dz = sz[p] - qz;
        float d2 = dx * dx + (dy * dy + dz * dz);
        if (!in) d2 = INFINITY;
        if (__any(d2 < Ld[K - 1])) {   // compare-and-shift insertion of a distance (indices are irrelevant for a bound)
            bool lt_cur = d2 < Ld[K - 1];
#pragma unroll
            for (int e = K - 1; e >= 1; --e) {
                bool lt_prev = d2 < Ld[e - 1];
                Ld[e] = lt_prev ? Ld[e - 1] : (lt_cur ? d2 : Ld[e]);
                lt_cur = lt_prev;
            }
            Ld[0] = lt_cur ? d2 : Ld[0];
        }
    }
    float kth = Ld[K - 1];
    if (k < K) {
#pragma unroll
        for (int e = 0; e < K; ++e) if (e == k - 1) kth = Ld[e];
    }
    if (slot < nqq) bound[slot] = kth;   // +inf when the window holds fewer than k points
}

// Phase B: the full brute-force scan with a FIXED per-query bound and no selection state in the loop:
// two queries per lane, 16 targets per scalar-load step (the ICP scan's shape).  Every target with d2 <= bound is
// appended to the query's candidate row (64-bit key = d2 bits : original index; order in the row is irrelevant).
constexpr int CS_SPL = 2;
constexpr int CS_CH = 16;
__global__ __launch_bounds__(KN_BLOCK)
void k_collect_scan(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                    const int* __restrict__ orig, int n, int n_chunks, int n_super, int supers_per_split,
                    const float* __restrict__ cbox, const float* __restrict__ sbox, int prune,
                    const int* __restrict__ qsel, int nqq, const float* __restrict__ bound, int cap,
                    int* __restrict__ cnt, unsigned long long* __restrict__ cand) {
    const int split = blockIdx.y;
    const int u0 = split * supers_per_split, u1 = min(n_super, u0 + supers_per_split);
    float qx[CS_SPL], qy[CS_SPL], qz[CS_SPL], B[CS_SPL];
    int slot[CS_SPL];
#pragma unroll
    for (int s = 0; s < CS_SPL; ++s) {
        slot[s] = blockIdx.x * (KN_BLOCK * CS_SPL) + s * KN_BLOCK + threadIdx.x;
        const int sl = min(slot[s], nqq - 1);
        const int sp = qsel ? qsel[sl] : sl;
        qx[s] = sx[sp]; qy[s] = sy[sp]; qz[s] = sz[sp];
        B[s] = slot[s] < nqq ? bound[sl] : -1.f;   // padding lanes never accept
    }
    // the wave's query box and largest bound (padding lanes duplicate a live query, their bound is -1)
    float qmin[3] = {fminf(qx[0], qx[1]), fminf(qy[0], qy[1]), fminf(qz[0], qz[1])};
    float qmax[3] = {fmaxf(qx[0], qx[1]), fmaxf(qy[0], qy[1]), fmaxf(qz[0], qz[1])};
    float Bmax = fmaxf(B[0], B[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { qmin[a] = fminf(qmin[a], __shfl_xor(qmin[a], off, 64)); qmax[a] = fmaxf(qmax[a], __shfl_xor(qmax[a], off, 64)); }
        Bmax = fmaxf(Bmax, __shfl_xor(Bmax, off, 64));
    }
    for (int u = u0; u < u1; ++u) {
        if (prune && !__any(box_lower_bound(sbox, n_super, u, qmin, qmax) <= Bmax)) continue;   // 256 targets skipped
        const int cend = min(n_chunks, u * 16 + 16);
        for (int c = u * 16; c < cend; ++c) {
            if (prune && !__any(box_lower_bound(cbox, n_chunks, c, qmin, qmax) <= Bmax)) continue;   // 16 targets skipped
            const int j = c * CS_CH;
            float tx[CS_CH], ty[CS_CH], tz[CS_CH];   // wave-uniform: three s_load_dwordx16
#pragma unroll
            for (int t = 0; t < CS_CH; ++t) { tx[t] = sx[j + t]; ty[t] = sy[j + t]; tz[t] = sz[j + t]; }
            bool hit = false;
#pragma unroll
            for (int s = 0; s < CS_SPL; ++s) {
                float m = INFINITY;
#pragma unroll
                for (int t = 0; t < CS_CH; ++t) {
                    float dx = tx[t] - qx[s], dy = ty[t] - qy[s], dz = tz[t] - qz[s];   // (points[i] - query)
                    float d2 = dx * dx + (dy * dy + dz * dz);
                    m = fminf(m, d2);
                }
                hit |= m <= B[s];
            }
            if (!__any(hit)) continue;
#pragma unroll
            for (int s = 0; s < CS_SPL; ++s) {
#pragma unroll
                for (int t = 0; t < CS_CH; ++t) {
                    float dx = tx[t] - qx[s], dy = ty[t] - qy[s], dz = tz[t] - qz[s];
                    float d2 = dx * dx + (dy * dy + dz * dz);
                    const bool acc = (j + t < n) && d2 <= B[s];
                    if (__any(acc)) {
                        const int oi = orig[j + t];
                        if (acc) {
                            int at = atomicAdd(&cnt[slot[s]], 1);
                            if (at < cap) cand[(size_t)slot[s] * cap + at] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)oi;
                        }
                    }
                }
            }
        }
    }
}

// Phase C: exact top-k of a query's candidates in (d2, idx) order; the final list goes to lists[r*n_pad + i], cnt_out[i]
// (by original index).  overflow[slot] = 1 when the row was too small (the caller re-runs those queries with
// bound_next, the k-th smallest of the cap candidates it did keep: a valid, tighter bound).
// One WAVE per query: the row (<= 128 keys, 64-bit = d2 bits : original index, so unsigned order is (d2, idx) order)
// is sorted by a bitonic network across the lanes, two keys per lane.
__device__ __forceinline__ void cmpx_keys(unsigned& hi, unsigned& lo, unsigned phi, unsigned plo, bool keep_min) {
    const bool p_less = (phi < hi) || (phi == hi && plo < lo);
    const bool take = keep_min ? p_less : !p_less;   // keys are distinct (distinct indices) except the ~0 padding
    hi = take ? phi : hi; lo = take ? plo : lo;
}
constexpr int SEL_WAVES = KN_BLOCK / 64;
__global__ __launch_bounds__(KN_BLOCK)
void k_select_topk(const int* __restrict__ orig, const int* __restrict__ qsel, int nqq, int k, int cap, int n_pad,
                   const int* __restrict__ cnt, const unsigned long long* __restrict__ cand,
                   int* __restrict__ lists, int* __restrict__ cnt_out, int* __restrict__ overflow, float* __restrict__ bound_next) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * SEL_WAVES + (threadIdx.x >> 6);
    if (slot >= nqq) return;   // wave-uniform
    const int m_all = cnt[slot];
    const int m = min(m_all, cap);
    unsigned hi[2], lo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = r * 64 + lane;
        const unsigned long long key = e < m ? cand[(size_t)slot * cap + e] : ~0ull;
        hi[r] = (unsigned)(key >> 32); lo[r] = (unsigned)key;
    }
    // element index i = r*64 + lane; ascending bitonic sort of 128 (or of the first 64 when the row is short)
    const int nsort = m > 64 ? 128 : 64;   // wave-uniform
    for (int kk = 2; kk <= nsort; kk <<= 1) {
        for (int j = kk >> 1; j >= 1; j >>= 1) {
            if (j == 64) {   // partner is the lane's other key (only in the 128 sort, kk == 128: ascending everywhere)
                const bool swap = (hi[1] < hi[0]) || (hi[1] == hi[0] && lo[1] < lo[0]);
                const unsigned th = hi[0], tl = lo[0];
                hi[0] = swap ? hi[1] : hi[0]; lo[0] = swap ? lo[1] : lo[0];
                hi[1] = swap ? th : hi[1]; lo[1] = swap ? tl : lo[1];
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (r == 1 && nsort == 64) continue;
                    const int i = r * 64 + lane;
                    const unsigned phi = __shfl_xor(hi[r], j, 64), plo = __shfl_xor(lo[r], j, 64);
                    const bool asc = (i & kk) == 0;
                    const bool lower = (i & j) == 0;
                    cmpx_keys(hi[r], lo[r], phi, plo, asc == lower);
                }
            }
        }
    }
    const int i0 = orig[qsel ? qsel[slot] : slot];
    const int c = min(k, m);
    if (lane < c) lists[(size_t)lane * n_pad + i0] = (int)lo[0];   // k <= 32: the output ranks sit in the first key of lanes 0..k-1
    const unsigned kbits = __shfl(hi[0], k - 1, 64);
    const float kth = kbits == 0xffffffffu ? INFINITY : __uint_as_float(kbits);   // padding: fewer than k candidates
    if (lane == 0) {
        cnt_out[i0] = c;
        overflow[slot] = m_all > cap ? 1 : 0;
        bound_next[slot] = kth;
    }
}
__global__ void k_compact_overflow(const int* __restrict__ flag, const int* __restrict__ pos, const int* __restrict__ qsel, int nqq,
                                   const float* __restrict__ bound_next, int* __restrict__ qsel2, float* __restrict__ bound2) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot < nqq && flag[slot]) { qsel2[pos[slot]] = qsel ? qsel[slot] : slot; bound2[pos[slot]] = bound_next[slot]; }  // ascending sorted positions
}

// ------------------------------------------------------------------ global-list top-k (any k; FPFH radius search)
// Sorted per-lane list with elements at base[e * stride] in global memory; pending queue in LDS.
__device__ __forceinline__ void glob_merge(float* __restrict__ ld, int* __restrict__ li, size_t stride,
                                           const float* __restrict__ pd_lds, const int* __restrict__ pi_lds,
                                           int k, int& cnt_list, int& pcnt, float bound0, float& bound) {
    float pd[KN_PB]; int pi[KN_PB]; int prank[KN_PB];
#pragma unroll
    for (int p = 0; p < KN_PB; ++p) {
        bool v = p < pcnt;
        pd[p] = v ? pd_lds[p * KN_BLOCK] : INFINITY;
        pi[p] = v ? pi_lds[p * KN_BLOCK] : INT_MAX;
    }
#pragma unroll
    for (int a = 0; a < KN_PB; ++a) {
        int r = 0;
#pragma unroll
        for (int b = 0; b < KN_PB; ++b) r += (b != a && lex_less(pd[b], pi[b], pd[a], pi[a])) ? 1 : 0;
        prank[a] = r;
    }
    int maxc = cnt_list;  // wave-uniform trip count: the longest list in the wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off, 64));
    // descending e: an entry only moves right (new position >= e), onto slots already vacated
    for (int e = maxc - 1; e >= 0; --e) {
        bool valid = e < cnt_list;
        float d = valid ? ld[e * stride] : INFINITY;
        int i = valid ? li[e * stride] : INT_MAX;
        int shift = 0;
#pragma unroll
        for (int p = 0; p < KN_PB; ++p) {
            bool lt = lex_less(pd[p], pi[p], d, i);
            shift += lt ? 1 : 0;
            prank[p] += (valid && !lt) ? 1 : 0;
        }
        int np = e + shift;
        if (valid && shift > 0 && np < k) { ld[np * stride] = d; li[np * stride] = i; }
    }
#pragma unroll
    for (int p = 0; p < KN_PB; ++p) {
        if (p < pcnt && prank[p] < k) { ld[prank[p] * stride] = pd[p]; li[prank[p] * stride] = pi[p]; }
    }
    cnt_list = min(k, cnt_list + pcnt);
    pcnt = 0;
    if (cnt_list == k) bound = fminf(bound0, ld[(k - 1) * stride]);
}

__global__ __launch_bounds__(KN_BLOCK)
void k_topk_scan_glob(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                      const int* __restrict__ orig, int nq, int nq_pad, int n_chunks16, int n_super, int nsplit, int k, float bound0,
                      const float* __restrict__ cbox, const float* __restrict__ sbox, int prune,
                      float* __restrict__ out_d, int* __restrict__ out_i, int* __restrict__ out_cnt) {
    __shared__ float s_pd[KN_PB][KN_BLOCK];
    __shared__ int s_pi[KN_PB][KN_BLOCK];
    const int tid = threadIdx.x;
    const int split = blockIdx.y;
    const int qi = blockIdx.x * KN_BLOCK + tid;
    const int qc = min(qi, nq - 1);
    const float qx = sx[qc], qy = sy[qc], qz = sz[qc];
    const int cc = min((blockIdx.x * KN_BLOCK + KN_BLOCK / 2) / 256, n_super - 1);
    float* ld = out_d + (size_t)split * k * nq_pad + qi;
    int* li = out_i + (size_t)split * k * nq_pad + qi;
    const size_t stride = (size_t)nq_pad;
    const float* pdl = &s_pd[0][tid];
    const int* pil = &s_pi[0][tid];
    int cnt_list = 0, pcnt = 0;
    float bound = bound0;
    float qmin[3] = {qx, qy, qz}, qmax[3] = {qx, qy, qz};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { qmin[a] = fminf(qmin[a], __shfl_xor(qmin[a], off, 64)); qmax[a] = fmaxf(qmax[a], __shfl_xor(qmax[a], off, 64)); }
    float Bmax = bound0;
    for (int vs = split; vs < n_super; vs += nsplit) {
        const int u = visit_chunk(vs, cc, n_super);
        if (prune && !__any(box_lower_bound(sbox, n_super, u, qmin, qmax) <= Bmax)) continue;
        const int c16_end = min(n_chunks16, u * 16 + 16);
        for (int c16 = u * 16; c16 < c16_end; ++c16) {
            if (prune && !__any(box_lower_bound(cbox, n_chunks16, c16, qmin, qmax) <= Bmax)) continue;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int j = c16 * 16 + h * KN_CH;
                float d2[KN_CH];
#pragma unroll
                for (int t = 0; t < KN_CH; ++t) {
                    float dx = sx[j + t] - qx, dy = sy[j + t] - qy, dz = sz[j + t] - qz;
                    d2[t] = dx * dx + (dy * dy + dz * dz);
                }
                float m = fminf(fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])), fminf(fminf(d2[4], d2[5]), fminf(d2[6], d2[7])));
                if (!__any(m <= bound)) continue;
                int nacc = 0;
#pragma unroll
                for (int t = 0; t < KN_CH; ++t) nacc += (d2[t] <= bound) ? 1 : 0;
                if (__any(pcnt + nacc > KN_PB)) {
                    glob_merge(ld, li, stride, pdl, pil, k, cnt_list, pcnt, bound0, bound);
                    Bmax = bound;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) Bmax = fmaxf(Bmax, __shfl_xor(Bmax, off, 64));
                }
#pragma unroll
                for (int t = 0; t < KN_CH; ++t) {
                    if ((j + t < nq) && d2[t] <= bound) { s_pd[pcnt][tid] = d2[t]; s_pi[pcnt][tid] = orig[j + t]; pcnt++; }  // never a padding target
                }
            }
        }
    }
    if (__any(pcnt > 0)) glob_merge(ld, li, stride, pdl, pil, k, cnt_list, pcnt, bound0, bound);
    if (qi < nq_pad) out_cnt[(size_t)split * nq_pad + qi] = cnt_list;
}

// k-way merge of the per-split sorted lists of one query; calls emit(rank, d2, idx) in order.
template <class F>
__device__ __forceinline__ int merge_splits(const float* __restrict__ pd, const int* __restrict__ pi,
                                            const int* __restrict__ pc, int nsplit, int k, int nq_pad, int qi, F emit) {
    float hd[KN_MAXSPLIT]; int hi[KN_MAXSPLIT]; int pos[KN_MAXSPLIT]; int cn[KN_MAXSPLIT];
#pragma unroll
    for (int s = 0; s < KN_MAXSPLIT; ++s) {
        pos[s] = 0; cn[s] = s < nsplit ? pc[(size_t)s * nq_pad + qi] : 0;
        bool v = cn[s] > 0;
        hd[s] = v ? pd[((size_t)s * k) * nq_pad + qi] : INFINITY;
        hi[s] = v ? pi[((size_t)s * k) * nq_pad + qi] : INT_MAX;
    }
    int out = 0;
    for (; out < k; ++out) {
        int bs = -1; float bd = INFINITY; int bi = INT_MAX;
#pragma unroll
        for (int s = 0; s < KN_MAXSPLIT; ++s) {
            bool v = pos[s] < cn[s];
            if (v && (bs < 0 || lex_less(hd[s], hi[s], bd, bi))) { bd = hd[s]; bi = hi[s]; bs = s; }
        }
        if (bs < 0) break;
        emit(out, bd, bi);
#pragma unroll
        for (int s = 0; s < KN_MAXSPLIT; ++s) {
            if (s == bs) {
                pos[s]++;
                bool v = pos[s] < cn[s];
                hd[s] = v ? pd[((size_t)s * k + pos[s]) * nq_pad + qi] : INFINITY;
                hi[s] = v ? pi[((size_t)s * k + pos[s]) * nq_pad + qi] : INT_MAX;
            }
        }
    }
    return out;
}

// ------------------------------------------------------------------ list finishing (shared by kNN and radius)
// k-way merge of the per-split lists of each query of the launch (all sorted positions, or the subset qsel);
// the final list is stored BY ORIGINAL POINT INDEX: lists[r * n_pad + i], cnt[i].
__global__ __launch_bounds__(KN_BLOCK)
void k_lists_finish(const int* __restrict__ orig, const int* __restrict__ qsel, int nqq, int nq_pad, int n_pad, int k, int nsplit,
                    const float* __restrict__ pd, const int* __restrict__ pi, const int* __restrict__ pc,
                    int* __restrict__ lists, int* __restrict__ cnt) {
    const int slot = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (slot >= nqq) return;
    const int i = orig[qsel ? qsel[slot] : slot];
    int c = merge_splits(pd, pi, pc, nsplit, k, nq_pad, slot, [&](int r, float, int idx) { lists[(size_t)r * n_pad + i] = idx; });
    cnt[i] = c;
}

// ------------------------------------------------------------------ normals (registration.cpp:105-130)
// One lane per point (original index).  The k nearest neighbours come from listsA when it holds at least k
// entries (FPFH's radius list: sorted by (d2, idx), so its first k ARE the k nearest), else from listsB (kNN scan).
__global__ __launch_bounds__(KN_BLOCK)
void k_normals_from_lists(const float* __restrict__ xyz, int n, int n_pad, int k,
                          const int* __restrict__ listsA, const int* __restrict__ cntA,
                          const int* __restrict__ listsB, const int* __restrict__ cntB,
                          float* __restrict__ normals, int* __restrict__ knn_out, int knn_stride) {
    const int i = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const bool useA = listsA && cntA[i] >= k;
    const int* __restrict__ L = useA ? listsA : listsB;
    const int cnt = useA ? k : cntB[i];
    if (knn_out) for (int r = 0; r < knn_stride; ++r) knn_out[(size_t)i * knn_stride + r] = r < cnt ? L[(size_t)r * n_pad + i] : -1;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = L[(size_t)r * n_pad + i];
        cx += xyz[3 * (size_t)j]; cy += xyz[3 * (size_t)j + 1]; cz += xyz[3 * (size_t)j + 2];
    }
    const float fc = (float)cnt;
    cx /= fc; cy /= fc; cz /= fc;
    float c00 = 0.f, c10 = 0.f, c20 = 0.f, c11 = 0.f, c21 = 0.f, c22 = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = L[(size_t)r * n_pad + i];
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

__global__ __launch_bounds__(KN_BLOCK)
void k_spfh(const float* __restrict__ xyz, const float* __restrict__ nrm, int n, int n_pad,
            const int* __restrict__ nbr, const int* __restrict__ nbr_cnt, float* __restrict__ spfh) {
    __shared__ float hist[33][KN_BLOCK];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * KN_BLOCK + tid;
#pragma unroll
    for (int b = 0; b < 33; ++b) hist[b][tid] = 0.f;
    if (i >= n) return;
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    const float ux = nrm[3 * (size_t)i], uy = nrm[3 * (size_t)i + 1], uz = nrm[3 * (size_t)i + 2];
    const int cnt = nbr_cnt[i];
    for (int r = 0; r < cnt; ++r) {
        const int j = nbr[(size_t)r * n_pad + i];
        if (j == i) continue;
        float dx = xyz[3 * (size_t)j] - px, dy = xyz[3 * (size_t)j + 1] - py, dz = xyz[3 * (size_t)j + 2] - pz;
        float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
        if (dist < 1e-8f) continue;
        float ex = dx / dist, ey = dy / dist, ez = dz / dist;
        float vx = uy * ez - uz * ey, vy = uz * ex - ux * ez, vz = ux * ey - uy * ex;   // v = u x d
        float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;   // w = u x v
        float njx = nrm[3 * (size_t)j], njy = nrm[3 * (size_t)j + 1], njz = nrm[3 * (size_t)j + 2];
        float alpha = vx * njx + (vy * njy + vz * njz);
        float phi = ux * ex + (uy * ey + uz * ez);
        float wn = wx * njx + (wy * njy + wz * njz);
        float un = ux * njx + (uy * njy + uz * njz);
        float theta = (float)atan2((double)wn, (double)un);
        int bin_a = min(max((int)((alpha + 1.0f) * 5.5f), 0), 10);
        int bin_p = min(max((int)((phi + 1.0f) * 5.5f), 0), 10);
        int bin_t = min(max((int)(((double)theta / 3.14159265358979323846 + (double)1.0f) * (double)5.5f), 0), 10);
        hist[bin_a][tid] += 1.0f;
        hist[11 + bin_p][tid] += 1.0f;
        hist[22 + bin_t][tid] += 1.0f;
    }
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < 33; ++b) sum += hist[b][tid];
#pragma unroll
    for (int b = 0; b < 33; ++b) {
        float v = hist[b][tid];
        if (sum > 0.f) v /= sum;
        spfh[(size_t)i * 33 + b] = v;
    }
}

__global__ __launch_bounds__(KN_BLOCK)
void k_fpfh(const float* __restrict__ xyz, int n, int n_pad, const int* __restrict__ nbr, const int* __restrict__ nbr_cnt,
            const float* __restrict__ spfh, float* __restrict__ desc, int* __restrict__ nbr_out /* [n][100] or null */) {
    const int i = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (i >= n) return;
    float f[33];
#pragma unroll
    for (int d = 0; d < 33; ++d) f[d] = spfh[(size_t)i * 33 + d];
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    const int cnt = nbr_cnt[i];
    for (int r = 0; r < cnt; ++r) {
        const int j = nbr[(size_t)r * n_pad + i];
        if (nbr_out) nbr_out[(size_t)i * FP_MAXNN + r] = j;
        if (j == i) continue;
        float dx = xyz[3 * (size_t)j] - px, dy = xyz[3 * (size_t)j + 1] - py, dz = xyz[3 * (size_t)j + 2] - pz;
        float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
        if (dist < 1e-8f) continue;
        float w = 1.0f / dist;
        const float* sj = spfh + (size_t)j * 33;
#pragma unroll
        for (int d = 0; d < 33; ++d) f[d] += w * sj[d];
    }
    if (nbr_out) for (int r = cnt; r < FP_MAXNN; ++r) nbr_out[(size_t)i * FP_MAXNN + r] = -1;
    float sum = 0.f;
#pragma unroll
    for (int d = 0; d < 33; ++d) sum += f[d];
#pragma unroll
    for (int d = 0; d < 33; ++d) {
        float v = f[d];
        if (sum > 0.f) v /= sum;
        desc[(size_t)i * 33 + d] = v;
    }
}

namespace {

struct ScanPlan { int n_pad, nt_pad, n_chunks, blocks_x, nsplit; };

int pick_nsplit(int blocks_x, int n_super) {
    // interleaved splits over the 256-target super-chunks: enough workgroups for a short tail (>= ~6k),
    // each split keeping >= 2 super-chunks
    int want = (6144 + blocks_x - 1) / blocks_x;
    int max_split = std::max(1, n_super / 2);
    int nsplit = std::max(1, std::min(std::min(want, max_split), KN_MAXSPLIT));
    if (const char* e = getenv("TDV_KNN_NSPLIT")) nsplit = std::max(1, std::min(std::min(atoi(e), max_split), KN_MAXSPLIT));  // tuning knob
    return nsplit;
}

ScanPlan make_scan_plan(int n) {
    ScanPlan p;
    p.n_pad = (int)align_up((size_t)n, KN_BLOCK);
    p.nt_pad = (int)align_up((size_t)n, 16);
    p.n_chunks = p.nt_pad / 16;          // 16-target chunks
    p.blocks_x = p.n_pad / KN_BLOCK;
    p.nsplit = pick_nsplit(p.blocks_x, (p.n_chunks + 15) / 16);
    return p;
}

struct Sorted { float *sx, *sy, *sz; int* orig; float *cbox, *sbox; int n_chunks16, n_super; };

// Morton sort of the cloud: sorted SoA coordinates (padded with +inf) and the original index of each position
int spatial_sort(tdv_ctx* ctx, const float* d_xyz, int n, const ScanPlan& p, Sorted& so) {
    hipStream_t s = ctx->stream;
    const size_t n_pow2 = sort_pow2((size_t)n);
    const int pad = std::max(p.n_pad, (int)align_up((size_t)p.nt_pad, 16));   // multiple of 256 >= n: covers 8- and 16-target chunks
    float* soa; uint4* rec; float *part, *bbox;
    TDV_TRY(ws_alloc(ctx, (size_t)3 * pad, &soa));
    TDV_TRY(ws_alloc(ctx, (size_t)pad, &so.orig));
    TDV_TRY(ws_alloc(ctx, n_pow2, &rec));
    const int bblocks = std::min(1024, (n + 255) / 256);
    TDV_TRY(ws_alloc(ctx, (size_t)bblocks * 6, &part));
    TDV_TRY(ws_alloc(ctx, 6, &bbox));
    so.sx = soa; so.sy = soa + pad; so.sz = soa + 2 * (size_t)pad;
    k_bbox_partial<<<bblocks, 256, 0, s>>>(d_xyz, n, part);
    k_bbox_final<<<1, 64, 0, s>>>(part, bblocks, bbox);
    k_morton_records<<<(unsigned)((n_pow2 + 255) / 256), 256, 0, s>>>(d_xyz, n, (int)n_pow2, bbox, rec);
    TDV_TRY(sort_records_dev(ctx, rec, n_pow2));
    k_gather_sorted<<<(pad + 255) / 256, 256, 0, s>>>(d_xyz, rec, n, pad, so.sx, so.sy, so.sz, so.orig);
    // bounding boxes of the 16-target chunks and 256-target super-chunks (exact pruning of the scans)
    so.n_chunks16 = (int)(align_up((size_t)n, 16) / 16);
    so.n_super = (so.n_chunks16 + 15) / 16;
    TDV_TRY(ws_alloc(ctx, (size_t)6 * so.n_chunks16, &so.cbox));
    TDV_TRY(ws_alloc(ctx, (size_t)6 * so.n_super, &so.sbox));
    k_chunk_boxes<<<(so.n_chunks16 + 255) / 256, 256, 0, s>>>(so.sx, so.sy, so.sz, so.n_chunks16, so.cbox);
    k_super_boxes<<<(so.n_super + 255) / 256, 256, 0, s>>>(so.cbox, so.n_chunks16, so.n_super, so.sbox);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

}  // namespace

int spatial_sort_cloud(tdv_ctx* ctx, const float* d_xyz, int n, SortedCloud& out) {
    const ScanPlan p = make_scan_plan(n);
    Sorted so;
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    out.sx = so.sx; out.sy = so.sy; out.sz = so.sz; out.orig = so.orig; out.cbox = so.cbox; out.sbox = so.sbox;
    out.n = n; out.pad = std::max(p.n_pad, (int)align_up((size_t)p.nt_pad, 16));
    out.n_chunks16 = so.n_chunks16; out.n_super = so.n_super; out.n_top = (so.n_super + 15) / 16;
    TDV_TRY(ws_alloc(ctx, (size_t)6 * out.n_top, &out.tbox));
    k_super_boxes<<<(out.n_top + 255) / 256, 256, 0, ctx->stream>>>(so.sbox, so.n_super, out.n_top, out.tbox);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

namespace {

// Scan + per-split lists + merge into lists[r * p.n_pad + original index] / cnt[original index].
// qsel == nullptr: all n queries; else the nsel sorted positions in qsel (device).
int scan_to_lists(tdv_ctx* ctx, const Sorted& so, int n, const ScanPlan& p, int k, float bound0, int timer,
                  const int* qsel, int nsel, int* lists, int* cnt) {
    const int nqq = qsel ? nsel : n;
    if (nqq <= 0) return TDV_OK;
    const int nq_pad = (int)align_up((size_t)nqq, KN_BLOCK);
    const int blocks_x = nq_pad / KN_BLOCK;
    const int nsplit = qsel ? pick_nsplit(blocks_x, so.n_super) : p.nsplit;
    static const int prune = getenv("TDV_NO_PRUNE") ? 0 : 1;   // A/B knob: 0 = evaluate every pair (pure brute force)
    float* pd; int *pi, *pc;
    TDV_TRY(ws_alloc(ctx, (size_t)nsplit * k * nq_pad, &pd));
    TDV_TRY(ws_alloc(ctx, (size_t)nsplit * k * nq_pad, &pi));
    TDV_TRY(ws_alloc(ctx, (size_t)nsplit * nq_pad, &pc));
    hipStream_t s = ctx->stream;
    {
        ScopedTimer tm(ctx, timer);
        dim3 grid(blocks_x, nsplit);
#define TDV_REG_SCAN(KK) k_topk_scan_reg<KK><<<grid, KN_BLOCK, 0, s>>>(so.sx, so.sy, so.sz, so.orig, n, nq_pad, so.n_chunks16, so.n_super, nsplit, k, bound0, \
                                                                     so.cbox, so.sbox, prune, qsel, nsel, pd, pi, pc)
        if (k <= 8) TDV_REG_SCAN(8);
        else if (k <= 16) TDV_REG_SCAN(16);
        else if (k <= 30) TDV_REG_SCAN(30);
        else if (k <= 32) TDV_REG_SCAN(32);
        else {
            if (qsel) return TDV_ERR_INTERNAL;  // subset scans are only issued for k <= 32
            k_topk_scan_glob<<<grid, KN_BLOCK, 0, s>>>(so.sx, so.sy, so.sz, so.orig, n, nq_pad, so.n_chunks16, so.n_super, nsplit, k, bound0,
                                                       so.cbox, so.sbox, prune, pd, pi, pc);
        }
#undef TDV_REG_SCAN
    }
    k_lists_finish<<<blocks_x, KN_BLOCK, 0, s>>>(so.orig, qsel, nqq, nq_pad, p.n_pad, k, nsplit, pd, pi, pc, lists, cnt);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

}  // namespace

namespace {

constexpr int CS_CAP = 128;       // candidate row per query (k <= 32): typical fill 30-70
constexpr int CS_HALF_WINDOW = 384;

// exact kNN lists (k <= 32) of all queries (qsel == nullptr) or of the subset qsel, by the two-phase scheme;
// queries whose candidate row overflows (or whose window gave no finite bound) are redone by the streaming scan.
int knn_to_lists(tdv_ctx* ctx, const Sorted& so, int n, const ScanPlan& p, int k, const int* qsel, int nsel, int* lists, int* cnt) {
    int nqq = qsel ? nsel : n;
    if (nqq <= 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    const int nq_pad = (int)align_up((size_t)nqq, KN_BLOCK * CS_SPL);
    float *bound, *bnext, *bound2; int *ccnt, *ovf, *pos, *qselA, *qselB, *d_total; unsigned long long* cand;
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &bound));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &bnext));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &bound2));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &ccnt));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &ovf));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &pos));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &qselA));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad, &qselB));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(ws_alloc(ctx, (size_t)nq_pad * CS_CAP, &cand));
    TDV_TRY(pin_reserve(ctx, 64));
    int* h_total = reinterpret_cast<int*>(ctx->pin);
    static const int prune = getenv("TDV_NO_PRUNE") ? 0 : 1;   // A/B knob: 0 = evaluate every pair (pure brute force)
    const int* cur_q = qsel;
    float* cur_b = bound;
    int* next_q = qselA;
    {   // phase A: bounds from the local window
        ScopedTimer tm(ctx, TDV_TIMER_KNN);
        const int qblocks = (nqq + KN_BLOCK - 1) / KN_BLOCK;
#define TDV_WB(KK) k_window_bound<KK><<<qblocks, KN_BLOCK, 0, s>>>(so.sx, so.sy, so.sz, n, cur_q, nqq, k, CS_HALF_WINDOW, cur_b)
        if (k <= 8) TDV_WB(8); else if (k <= 16) TDV_WB(16); else if (k <= 30) TDV_WB(30); else TDV_WB(32);
#undef TDV_WB
    }
    for (int round = 0; round < 8 && nqq > 0; ++round) {
        const int qblocks = (nqq + KN_BLOCK - 1) / KN_BLOCK;
        const int cblocks = (nqq + KN_BLOCK * CS_SPL - 1) / (KN_BLOCK * CS_SPL);
        int want = (8192 + cblocks - 1) / cblocks;
        int csplit = std::max(1, std::min(want, so.n_super));
        int ups = (so.n_super + csplit - 1) / csplit;      // super-chunks per split
        csplit = (so.n_super + ups - 1) / ups;
        TDV_HIP(ctx, hipMemsetAsync(ccnt, 0, (size_t)nqq * 4, s));
        {
            ScopedTimer tm(ctx, TDV_TIMER_KNN);
            k_collect_scan<<<dim3(cblocks, csplit), KN_BLOCK, 0, s>>>(so.sx, so.sy, so.sz, so.orig, n, so.n_chunks16, so.n_super, ups, so.cbox, so.sbox, prune,
                                                                      cur_q, nqq, cur_b, CS_CAP, ccnt, cand);
            static_assert(CS_CAP == 128, "k_select_topk sorts rows of at most 128 keys");
            k_select_topk<<<(nqq + SEL_WAVES - 1) / SEL_WAVES, KN_BLOCK, 0, s>>>(so.orig, cur_q, nqq, k, CS_CAP, p.n_pad, ccnt, cand, lists, cnt, ovf, bnext);
        }
        TDV_CHECK_LAUNCH(ctx);
        // queries whose candidate row overflowed go another round with the tightened bound
        TDV_TRY(exclusive_scan_dev(ctx, ovf, nqq, pos, d_total));
        float* nb = (cur_b == bound2) ? bound : bound2;
        k_compact_overflow<<<qblocks, KN_BLOCK, 0, s>>>(ovf, pos, cur_q, nqq, bnext, next_q, nb);
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipMemcpyAsync(h_total, d_total, 4, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipStreamSynchronize(s));
        if (getenv("TDV_DEBUG")) fprintf(stderr, "[tdv] knn_to_lists: round %d queries=%d overflow=%d (k=%d)\n", round, nqq, *h_total, k);
        nqq = *h_total;
        cur_q = next_q; next_q = (next_q == qselA) ? qselB : qselA;
        cur_b = nb;
    }
    // pathological leftovers (e.g. more than CS_CAP points at exactly the k-th distance): streaming scan
    if (nqq > 0) TDV_TRY(scan_to_lists(ctx, so, n, p, k, INFINITY, TDV_TIMER_KNN, cur_q, nqq, lists, cnt));
    return TDV_OK;
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
    static const bool streaming_only = getenv("TDV_KNN_STREAMING") != nullptr;  // tuning / A-B knob
    if (kk <= 32 && !streaming_only) TDV_TRY(knn_to_lists(ctx, so, n, p, kk, nullptr, 0, lists, cnt));
    else TDV_TRY(scan_to_lists(ctx, so, n, p, kk, INFINITY, TDV_TIMER_KNN, nullptr, 0, lists, cnt));
    k_normals_from_lists<<<p.blocks_x, KN_BLOCK, 0, ctx->stream>>>(d_xyz, n, p.n_pad, kk, nullptr, nullptr, lists, cnt, d_normals, d_knn, k);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

namespace {
int fpfh_from_lists(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, const ScanPlan& p, const int* nbr, const int* cnt,
                    float* d_desc, int* d_nbr, int* d_nbr_cnt) {
    float* spfh;
    TDV_TRY(ws_alloc(ctx, (size_t)n * 33, &spfh));
    hipStream_t s = ctx->stream;
    k_spfh<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, d_normals, n, p.n_pad, nbr, cnt, spfh);
    k_fpfh<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, nbr, cnt, spfh, d_desc, d_nbr);
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
    TDV_TRY(scan_to_lists(ctx, so, n, p, FP_MAXNN, r2, TDV_TIMER_RADIUS, nullptr, 0, nbr, cnt));
    return fpfh_from_lists(ctx, d_xyz, d_normals, n, p, nbr, cnt, d_desc, d_nbr, d_nbr_cnt);
}

// estimateNormals(k) followed by computeFPFH(radius) on the same cloud (src/pipeline.cpp:93-95), sharing one
// spatial sort and ONE full scan: the radius lists are sorted by (d2, idx), so wherever a point has >= k
// neighbours in radius its k nearest neighbours are the first k entries; only the deficient points (isolated
// points, silhouette edges) go through a kNN scan, as a subset.  Results are identical to the two separate calls.
int normals_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float radius, float* d_normals, float* d_desc) {
    if (!ctx || n < 0 || k <= 0 || k > 255 || (n > 0 && (!d_xyz || !d_normals || !d_desc))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const int kk = std::min(k, n);
    if (kk > 32 || kk > FP_MAXNN) {  // subset scans use the register kernel (k <= 32): otherwise the two plain calls
        TDV_TRY(estimate_normals_dev(ctx, d_xyz, n, k, d_normals, nullptr));
        return compute_fpfh_dev(ctx, d_xyz, d_normals, n, radius, d_desc, nullptr, nullptr);
    }
    const float r2 = radius * radius;
    const ScanPlan p = make_scan_plan(n);
    hipStream_t s = ctx->stream;
    Sorted so; int *nbr, *cnt, *flag, *pos, *qsel, *d_total, *listsK, *cntK;
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    TDV_TRY(ws_alloc(ctx, (size_t)FP_MAXNN * p.n_pad, &nbr));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cnt));
    TDV_TRY(scan_to_lists(ctx, so, n, p, FP_MAXNN, r2, TDV_TIMER_RADIUS, nullptr, 0, nbr, cnt));
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
    k_normals_from_lists<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, kk, nbr, cnt, listsK, cntK, d_normals, nullptr, 0);
    TDV_CHECK_LAUNCH(ctx);
    return fpfh_from_lists(ctx, d_xyz, d_normals, n, p, nbr, cnt, d_desc, nullptr, nullptr);
}

}  // namespace tdv
