// Exact brute-force neighbour searches on gfx950 and the per-point estimators built on them.
//
// Replaces (no GPU entry point exists in the reference; src/pipeline.cpp:93-95 calls the CPU statics):
//   findKNN + Registration::estimateNormals   /root/reference/src/registration.cpp:63-81, :105-130
//   findRadiusNN + Registration::computeFPFH  /root/reference/src/registration.cpp:83-102, :133-201
//
// Every query still evaluates its distance to EVERY point (the reference's O(N^2) scan, 8 VALU ops per
// pair, no FMA: d2 = dx*dx + (dy*dy + dz*dz)); what is engineered is the ORDER of the scan, so that the
// selection work around it nearly vanishes:
//   1. the cloud is sorted along a 30-bit Morton curve (bitonic sort shared with voxel.hip), so the 64
//      queries of a wave are spatial neighbours and so are the 8 targets of a chunk;
//   2. each workgroup scans the chunks INSIDE-OUT, starting at its own position on the curve and
//      alternating right/left; the true neighbours arrive within the first few hundred chunks, the
//      per-lane bound (radius^2, or the k-th best distance so far) is tight almost immediately, and
//      the remaining >99 % of the chunks take the fast path: 64 distance ops + a min3 tree + one
//      compare per chunk, targets broadcast through the scalar data path (wave-uniform s_load);
//   3. target splits (more workgroups, shorter tail) take INTERLEAVED positions of that visit order,
//      so every split sees near chunks first; per-split sorted lists are k-way merged afterwards.
// Selection keeps the reference's (d2, original index) lexicographic order of std::partial_sort /
// std::sort on pair<float,size_t>: candidates with d2 <= bound are queued (8 per lane), and when a
// lane's queue would overflow the wave inserts the queued entries into its sorted per-lane list —
// in VGPRs with static indexing for k <= 32 (normals, k = 30), in global memory (the output buffer
// itself, rank-merge without dependent chains) for larger k (FPFH: cap 100).
// Results are independent of the scan order; lists, normals and descriptors are bit-identical to the
// CPU code's (see tests/test_gpu_features.py).
// Per-point estimators run one lane per point with sequential sums in neighbour order, i.e. the same
// f32 expression trees as the CPU loops; atan2 is evaluated in f64 and rounded once.
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
__global__ void k_bbox_final(const float* __restrict__ part, int nblocks, float* __restrict__ bbox /* min xyz, max xyz */) {
    int c = threadIdx.x;
    if (c >= 6) return;
    float v = part[c];
    for (int b = 1; b < nblocks; ++b) v = c < 3 ? fminf(v, part[b * 6 + c]) : fmaxf(v, part[b * 6 + c]);
    bbox[c] = v;
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
                     const int* __restrict__ orig, int nq, int nq_pad, int n_chunks, int nsplit, int k, float bound0,
                     float* __restrict__ out_d, int* __restrict__ out_i, int* __restrict__ out_cnt) {
    const int split = blockIdx.y;
    const int qi = blockIdx.x * KN_BLOCK + threadIdx.x;
    const int qc = min(qi, nq - 1);
    const float qx = sx[qc], qy = sy[qc], qz = sz[qc];
    const int cc = min((blockIdx.x * KN_BLOCK + KN_BLOCK / 2) / KN_CH, n_chunks - 1);
    float Ld[K]; int Li[K];
#pragma unroll
    for (int e = 0; e < K; ++e) { Ld[e] = INFINITY; Li[e] = INT_MAX; }
    float pd[KN_PB]; int pi[KN_PB];
#pragma unroll
    for (int s = 0; s < KN_PB; ++s) { pd[s] = INFINITY; pi[s] = INT_MAX; }
    int cnt = 0, pcnt = 0;
    float bound = bound0;
    for (int v = split; v < n_chunks; v += nsplit) {
        const int j = visit_chunk(v, cc, n_chunks) * KN_CH;
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
        if (__any(pcnt + nacc > KN_PB)) reg_merge<K>(Ld, Li, pd, pi, k, cnt, pcnt, bound0, bound);
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
                      const int* __restrict__ orig, int nq, int nq_pad, int n_chunks, int nsplit, int k, float bound0,
                      float* __restrict__ out_d, int* __restrict__ out_i, int* __restrict__ out_cnt) {
    __shared__ float s_pd[KN_PB][KN_BLOCK];
    __shared__ int s_pi[KN_PB][KN_BLOCK];
    const int tid = threadIdx.x;
    const int split = blockIdx.y;
    const int qi = blockIdx.x * KN_BLOCK + tid;
    const int qc = min(qi, nq - 1);
    const float qx = sx[qc], qy = sy[qc], qz = sz[qc];
    const int cc = min((blockIdx.x * KN_BLOCK + KN_BLOCK / 2) / KN_CH, n_chunks - 1);
    float* ld = out_d + (size_t)split * k * nq_pad + qi;
    int* li = out_i + (size_t)split * k * nq_pad + qi;
    const size_t stride = (size_t)nq_pad;
    const float* pdl = &s_pd[0][tid];
    const int* pil = &s_pi[0][tid];
    int cnt_list = 0, pcnt = 0;
    float bound = bound0;
    for (int v = split; v < n_chunks; v += nsplit) {
        const int j = visit_chunk(v, cc, n_chunks) * KN_CH;
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
        if (__any(pcnt + nacc > KN_PB)) glob_merge(ld, li, stride, pdl, pil, k, cnt_list, pcnt, bound0, bound);
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) {
            if ((j + t < nq) && d2[t] <= bound) { s_pd[pcnt][tid] = d2[t]; s_pi[pcnt][tid] = orig[j + t]; pcnt++; }  // never a padding target
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

// ------------------------------------------------------------------ normals (registration.cpp:105-130)
// one lane per SORTED position sp; everything is written at the point's original index
__global__ __launch_bounds__(KN_BLOCK)
void k_normals_finish(const float* __restrict__ xyz, const int* __restrict__ orig, int n, int n_pad, int k, int nsplit,
                      const float* __restrict__ pd, const int* __restrict__ pi, const int* __restrict__ pc,
                      int* __restrict__ nbr /* [k][n_pad] scratch, by sorted position */, float* __restrict__ normals,
                      int* __restrict__ knn_out, int knn_stride) {
    const int sp = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (sp >= n) return;
    const int i = orig[sp];
    int cnt = merge_splits(pd, pi, pc, nsplit, k, n_pad, sp, [&](int r, float, int idx) { nbr[(size_t)r * n_pad + sp] = idx; });
    if (knn_out) for (int r = 0; r < knn_stride; ++r) knn_out[(size_t)i * knn_stride + r] = r < cnt ? nbr[(size_t)r * n_pad + sp] : -1;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = nbr[(size_t)r * n_pad + sp];
        cx += xyz[3 * (size_t)j]; cy += xyz[3 * (size_t)j + 1]; cz += xyz[3 * (size_t)j + 2];
    }
    const float fc = (float)cnt;
    cx /= fc; cy /= fc; cz /= fc;
    float c00 = 0.f, c10 = 0.f, c20 = 0.f, c11 = 0.f, c21 = 0.f, c22 = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = nbr[(size_t)r * n_pad + sp];
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

// ------------------------------------------------------------------ FPFH (registration.cpp:133-201)
constexpr int FP_MAXNN = 100;

// lists are re-indexed by ORIGINAL point index here so that the SPFH / FPFH passes run in input order
__global__ __launch_bounds__(KN_BLOCK)
void k_radius_finish(const int* __restrict__ orig, int n, int n_pad, int nsplit, const float* __restrict__ pd, const int* __restrict__ pi,
                     const int* __restrict__ pc, int* __restrict__ nbr /* [100][n_pad] by original index */, int* __restrict__ nbr_cnt) {
    const int sp = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (sp >= n) return;
    const int i = orig[sp];
    int cnt = merge_splits(pd, pi, pc, nsplit, FP_MAXNN, n_pad, sp, [&](int r, float, int idx) { nbr[(size_t)r * n_pad + i] = idx; });
    nbr_cnt[i] = cnt;
}

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

ScanPlan make_scan_plan(int n) {
    ScanPlan p;
    p.n_pad = (int)align_up((size_t)n, KN_BLOCK);
    p.nt_pad = (int)align_up((size_t)n, KN_CH);
    p.n_chunks = p.nt_pad / KN_CH;
    p.blocks_x = p.n_pad / KN_BLOCK;
    // interleaved splits: enough workgroups for a short tail (>= ~6k), each split keeping >= 64 chunks
    int want = (6144 + p.blocks_x - 1) / p.blocks_x;
    int max_split = std::max(1, p.n_chunks / 64);
    p.nsplit = std::max(1, std::min(std::min(want, max_split), KN_MAXSPLIT));
    if (const char* e = getenv("TDV_KNN_NSPLIT")) p.nsplit = std::max(1, std::min(std::min(atoi(e), max_split), KN_MAXSPLIT));  // tuning knob
    return p;
}

struct Sorted { float *sx, *sy, *sz; int* orig; };

// Morton sort of the cloud: sorted SoA coordinates (padded with +inf) and the original index of each position
int spatial_sort(tdv_ctx* ctx, const float* d_xyz, int n, const ScanPlan& p, Sorted& so) {
    hipStream_t s = ctx->stream;
    const size_t n_pow2 = sort_pow2((size_t)n);
    const int pad = std::max(p.n_pad, p.nt_pad);
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
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

int run_scan(tdv_ctx* ctx, const float* d_xyz, int n, int k, float bound0, int timer, const ScanPlan& p, Sorted& so,
             float** pd, int** pi, int** pc) {
    TDV_TRY(spatial_sort(ctx, d_xyz, n, p, so));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * k * p.n_pad, pd));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * k * p.n_pad, pi));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * p.n_pad, pc));
    hipStream_t s = ctx->stream;
    {
        ScopedTimer tm(ctx, timer);
        dim3 grid(p.blocks_x, p.nsplit);
#define TDV_REG_SCAN(KK) k_topk_scan_reg<KK><<<grid, KN_BLOCK, 0, s>>>(so.sx, so.sy, so.sz, so.orig, n, p.n_pad, p.n_chunks, p.nsplit, k, bound0, *pd, *pi, *pc)
        if (k <= 8) TDV_REG_SCAN(8);
        else if (k <= 16) TDV_REG_SCAN(16);
        else if (k <= 30) TDV_REG_SCAN(30);
        else if (k <= 32) TDV_REG_SCAN(32);
        else k_topk_scan_glob<<<grid, KN_BLOCK, 0, s>>>(so.sx, so.sy, so.sz, so.orig, n, p.n_pad, p.n_chunks, p.nsplit, k, bound0, *pd, *pi, *pc);
#undef TDV_REG_SCAN
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

}  // namespace

int estimate_normals_dev(tdv_ctx* ctx, const float* d_xyz, int n, int k, float* d_normals, int* d_knn) {
    if (!ctx || n < 0 || k <= 0 || k > 255 || (n > 0 && (!d_xyz || !d_normals))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const int kk = std::min(k, n);  // std::min(k, dists.size()), registration.cpp:74
    const ScanPlan p = make_scan_plan(n);
    Sorted so; float* pd; int *pi, *pc, *nbr;
    TDV_TRY(run_scan(ctx, d_xyz, n, kk, INFINITY, TDV_TIMER_KNN, p, so, &pd, &pi, &pc));
    TDV_TRY(ws_alloc(ctx, (size_t)kk * p.n_pad, &nbr));
    k_normals_finish<<<p.blocks_x, KN_BLOCK, 0, ctx->stream>>>(d_xyz, so.orig, n, p.n_pad, kk, p.nsplit, pd, pi, pc, nbr, d_normals, d_knn, k);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

int compute_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, float radius,
                     float* d_desc, int* d_nbr, int* d_nbr_cnt) {
    if (!ctx || n < 0 || (n > 0 && (!d_xyz || !d_normals || !d_desc))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const float r2 = radius * radius;  // registration.cpp:89
    const ScanPlan p = make_scan_plan(n);
    Sorted so; float* pd; int *pi, *pc, *nbr, *cnt; float* spfh;
    TDV_TRY(run_scan(ctx, d_xyz, n, FP_MAXNN, r2, TDV_TIMER_RADIUS, p, so, &pd, &pi, &pc));
    TDV_TRY(ws_alloc(ctx, (size_t)FP_MAXNN * p.n_pad, &nbr));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cnt));
    TDV_TRY(ws_alloc(ctx, (size_t)n * 33, &spfh));
    hipStream_t s = ctx->stream;
    k_radius_finish<<<p.blocks_x, KN_BLOCK, 0, s>>>(so.orig, n, p.n_pad, p.nsplit, pd, pi, pc, nbr, cnt);
    k_spfh<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, d_normals, n, p.n_pad, nbr, cnt, spfh);
    k_fpfh<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, nbr, cnt, spfh, d_desc, d_nbr);
    TDV_CHECK_LAUNCH(ctx);
    if (d_nbr_cnt) TDV_HIP(ctx, hipMemcpyAsync(d_nbr_cnt, cnt, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return TDV_OK;
}

}  // namespace tdv
