// Exact brute-force neighbour searches on gfx950 and the per-point estimators built on them.
//
// Replaces (no GPU entry point exists in the reference; src/pipeline.cpp:93-95 calls the CPU statics):
//   findKNN + Registration::estimateNormals   /root/reference/src/registration.cpp:63-81, :105-130
//   findRadiusNN + Registration::computeFPFH  /root/reference/src/registration.cpp:83-102, :133-201
//
// k_topk_scan — streaming top-k selection over a brute-force scan (VALU-bound, ~9 ops per pair):
//   one query point per lane in VGPRs; the cloud is broadcast as SoA through the scalar data path
//   (wave-uniform s_load_dwordx8 per coordinate, 8 targets per step).  The fast path only
//   evaluates d2 = dx*dx + (dy*dy + dz*dz) (no FMA) and compares it with the lane's current
//   inclusive bound (radius^2, or just below the k-th best so far).  Accepted candidates are
//   appended to a small per-lane pending buffer in LDS; when any lane's buffer is full the whole
//   wave merges pending entries into its sorted per-lane list by RANK (each element's final
//   position = number of smaller elements in the union), an in-place, latency-tolerant O(k*P)
//   step with no dependent chains.  Order is the reference's (d2, idx) lexicographic order of
//   std::partial_sort / std::sort on pair<float,size_t>, so the neighbour SETS AND ORDER are
//   exactly those of the CPU code.  Lists live in LDS for k <= 32 (normals, k = 30) and in
//   global memory (the output buffer itself) for larger k (FPFH, cap 100, where accepts are rare
//   because the radius bounds them from the start).
//   The target range is split over up to 8 workgroup columns; k_*_finish k-way merges the
//   per-split sorted lists.
// Per-point estimators run one lane per point with sequential sums in neighbour order, i.e. the
// same f32 expression trees as the CPU loops (centroid, covariance, Darboux features, histograms),
// so they are reproducible bit for bit; atan2 is evaluated in f64 and rounded once.
#include "tdv_internal.hpp"
#include "device_linalg.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <algorithm>

namespace tdv {

constexpr int KN_BLOCK = 256;
constexpr int KN_CH = 8;
constexpr int KN_PB = 8;
constexpr int KN_KCAP_LDS = 32;
constexpr int KN_MAXSPLIT = 8;

__global__ void k_aos_to_soa_pad2(const float* __restrict__ aos, int n, int n_pad, float pad,
                                  float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    if (i < n) { x[i] = aos[3 * i]; y[i] = aos[3 * i + 1]; z[i] = aos[3 * i + 2]; }
    else { x[i] = pad; y[i] = pad; z[i] = pad; }
}

__device__ __forceinline__ bool lex_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

__device__ __forceinline__ float float_pred(float x) {  // largest float < x, for x >= 0 finite or +inf
    if (x == 0.f) return -FLT_TRUE_MIN;
    return __int_as_float(__float_as_int(x) - 1);
}

// Sorted per-lane list with elements at base[e * stride]; LDS or global.
template <bool LDSLIST>
__device__ __forceinline__ void topk_merge(float* __restrict__ ld, int* __restrict__ li, size_t stride,
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
    // wave-uniform trip count: the longest list in the wave
    int maxc = cnt_list;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off, 64));
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
    if (cnt_list == k) bound = fminf(bound0, float_pred(ld[(k - 1) * stride]));
}

// out lists: element e of query i in split s at [(s*k + e) * nq_pad + i]; counts at [s*nq_pad + i]
template <bool LDSLIST>
__global__ __launch_bounds__(KN_BLOCK)
void k_topk_scan(const float* __restrict__ q_aos, int nq, int nq_pad,
                 const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                 int n_chunks, int chunks_per_split, int k, float bound0,
                 float* __restrict__ out_d, int* __restrict__ out_i, int* __restrict__ out_cnt) {
    __shared__ float s_pd[KN_PB][KN_BLOCK];
    __shared__ int s_pi[KN_PB][KN_BLOCK];
    __shared__ float s_ld[LDSLIST ? KN_KCAP_LDS : 1][KN_BLOCK];
    __shared__ int s_li[LDSLIST ? KN_KCAP_LDS : 1][KN_BLOCK];
    const int tid = threadIdx.x;
    const int split = blockIdx.y;
    const int c0 = split * chunks_per_split;
    const int c1 = min(n_chunks, c0 + chunks_per_split);
    const int qi = blockIdx.x * KN_BLOCK + tid;
    const int qc = min(qi, nq - 1);
    const float qx = q_aos[3 * qc], qy = q_aos[3 * qc + 1], qz = q_aos[3 * qc + 2];
    float* ld; int* li; size_t stride;
    if (LDSLIST) { ld = &s_ld[0][tid]; li = &s_li[0][tid]; stride = KN_BLOCK; }
    else { ld = out_d + (size_t)split * k * nq_pad + qi; li = out_i + (size_t)split * k * nq_pad + qi; stride = (size_t)nq_pad; }
    const float* pdl = &s_pd[0][tid];
    const int* pil = &s_pi[0][tid];
    int cnt_list = 0, pcnt = 0;
    float bound = bound0;
    for (int c = c0; c < c1; ++c) {
        const int j = c * KN_CH;
        float d2[KN_CH];
        int nacc = 0;
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) {
            float dx = tx[j + t] - qx, dy = ty[j + t] - qy, dz = tz[j + t] - qz;  // (points[i] - query)
            d2[t] = dx * dx + (dy * dy + dz * dz);
        }
        float m = fminf(fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])), fminf(fminf(d2[4], d2[5]), fminf(d2[6], d2[7])));
        if (!__any(m <= bound)) continue;
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) nacc += (d2[t] <= bound) ? 1 : 0;
        if (__any(pcnt + nacc > KN_PB)) topk_merge<LDSLIST>(ld, li, stride, pdl, pil, k, cnt_list, pcnt, bound0, bound);
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) {
            if (d2[t] <= bound) { s_pd[pcnt][tid] = d2[t]; s_pi[pcnt][tid] = j + t; pcnt++; }
        }
    }
    if (__any(pcnt > 0)) topk_merge<LDSLIST>(ld, li, stride, pdl, pil, k, cnt_list, pcnt, bound0, bound);
    if (qi < nq_pad) {
        out_cnt[(size_t)split * nq_pad + qi] = cnt_list;
        if (LDSLIST) {
            for (int e = 0; e < cnt_list; ++e) {
                out_d[((size_t)split * k + e) * nq_pad + qi] = s_ld[e][tid];
                out_i[((size_t)split * k + e) * nq_pad + qi] = s_li[e][tid];
            }
        }
    }
}

// Register-resident variant for small k (normals: k = 30): the sorted list (K entries) and an
// 8-entry pending queue live in VGPRs with static indexing only, so the kernel uses no LDS and runs
// at 4-5 waves per SIMD.  Fast path per chunk of 8 targets: 64 distance ops + a min3 tree + one
// compare.  Accepted candidates are pushed into the pending queue by predicated shifts; when a
// lane's queue is full the wave inserts the queued entries, oldest (lowest index) first, into the
// sorted list by a compare-and-shift sweep.  Because targets are scanned in ascending index, a new
// entry never precedes an equal-d2 entry already in the list, so `nd < Ld[e]` IS the (d2, idx)
// lexicographic test.
template <int K>
__device__ __forceinline__ void reg_insert(float (&Ld)[K], int (&Li)[K], float nd, int ni) {
    bool lt_cur = nd < Ld[K - 1];
#pragma unroll
    for (int e = K - 1; e >= 1; --e) {
        bool lt_prev = nd < Ld[e - 1];
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
    for (int s = KN_PB - 1; s >= 0; --s) {   // oldest (smallest target index) first
        if (s < maxp) {                      // wave-uniform
            float nd = s < pcnt ? pd[s] : INFINITY;
            reg_insert<K>(Ld, Li, nd, pi[s]);
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
    bound = (cnt == k) ? fminf(bound0, float_pred(kth)) : bound0;
}

template <int K>
__global__ __launch_bounds__(KN_BLOCK, 4)   // 4 waves per SIMD: keeps the K = 30 instance within 128 VGPRs
void k_topk_scan_reg(const float* __restrict__ q_aos, int nq, int nq_pad,
                     const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                     int n_chunks, int chunks_per_split, int k, float bound0,
                     float* __restrict__ out_d, int* __restrict__ out_i, int* __restrict__ out_cnt) {
    const int split = blockIdx.y;
    const int c0 = split * chunks_per_split;
    const int c1 = min(n_chunks, c0 + chunks_per_split);
    const int qi = blockIdx.x * KN_BLOCK + threadIdx.x;
    const int qc = min(qi, nq - 1);
    const float qx = q_aos[3 * qc], qy = q_aos[3 * qc + 1], qz = q_aos[3 * qc + 2];
    float Ld[K]; int Li[K];
#pragma unroll
    for (int e = 0; e < K; ++e) { Ld[e] = INFINITY; Li[e] = INT_MAX; }
    float pd[KN_PB]; int pi[KN_PB];
#pragma unroll
    for (int s = 0; s < KN_PB; ++s) { pd[s] = INFINITY; pi[s] = INT_MAX; }
    int cnt = 0, pcnt = 0;
    float bound = bound0;
    for (int c = c0; c < c1; ++c) {
        const int j = c * KN_CH;
        float d2[KN_CH];
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) {
            float dx = tx[j + t] - qx, dy = ty[j + t] - qy, dz = tz[j + t] - qz;
            d2[t] = dx * dx + (dy * dy + dz * dz);
        }
        float m = fminf(fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])), fminf(fminf(d2[4], d2[5]), fminf(d2[6], d2[7])));
        if (!__any(m <= bound)) continue;
        // slow path: make room ONCE per chunk (single merge site keeps the 8 appends statically indexed)
        int nacc = 0;
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) nacc += (d2[t] <= bound) ? 1 : 0;
        if (__any(pcnt + nacc > KN_PB)) reg_merge<K>(Ld, Li, pd, pi, k, cnt, pcnt, bound0, bound);
#pragma unroll
        for (int t = 0; t < KN_CH; ++t) {
            const bool acc = d2[t] <= bound;   // bound only tightens: entries refused now could never enter the list
            if (__any(acc)) {
#pragma unroll
                for (int s = KN_PB - 1; s >= 1; --s) { pd[s] = acc ? pd[s - 1] : pd[s]; pi[s] = acc ? pi[s - 1] : pi[s]; }
                pd[0] = acc ? d2[t] : pd[0];
                pi[0] = acc ? (j + t) : pi[0];
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
            if (v && lex_less(hd[s], hi[s], bd, bi)) { bd = hd[s]; bi = hi[s]; bs = s; }
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
__global__ __launch_bounds__(KN_BLOCK)
void k_normals_finish(const float* __restrict__ xyz, int n, int n_pad, int k, int nsplit,
                      const float* __restrict__ pd, const int* __restrict__ pi, const int* __restrict__ pc,
                      int* __restrict__ nbr /* [k][n_pad] scratch */, float* __restrict__ normals, int* __restrict__ knn_out) {
    const int i = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (i >= n) return;
    int cnt = merge_splits(pd, pi, pc, nsplit, k, n_pad, i, [&](int r, float, int idx) { nbr[(size_t)r * n_pad + i] = idx; });
    if (knn_out) for (int r = 0; r < k; ++r) knn_out[(size_t)i * k + r] = r < cnt ? nbr[(size_t)r * n_pad + i] : -1;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = nbr[(size_t)r * n_pad + i];
        cx += xyz[3 * j]; cy += xyz[3 * j + 1]; cz += xyz[3 * j + 2];
    }
    const float fc = (float)cnt;
    cx /= fc; cy /= fc; cz /= fc;
    float c00 = 0.f, c10 = 0.f, c20 = 0.f, c11 = 0.f, c21 = 0.f, c22 = 0.f;
    for (int r = 0; r < cnt; ++r) {
        int j = nbr[(size_t)r * n_pad + i];
        float dx = xyz[3 * j] - cx, dy = xyz[3 * j + 1] - cy, dz = xyz[3 * j + 2] - cz;
        c00 += dx * dx; c10 += dy * dx; c20 += dz * dx; c11 += dy * dy; c21 += dz * dy; c22 += dz * dz;
    }
    c00 /= fc; c10 /= fc; c20 /= fc; c11 /= fc; c21 /= fc; c22 /= fc;
    float nx, ny, nz;
    dl::smallest_eigvec3(c00, c10, c20, c11, c21, c22, nx, ny, nz);
    const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    float dot = nx * (-px) + (ny * (-py) + nz * (-pz));  // normals[i].dot(-points[i])
    if (dot < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    normals[3 * i] = nx; normals[3 * i + 1] = ny; normals[3 * i + 2] = nz;
}

// ------------------------------------------------------------------ FPFH (registration.cpp:133-201)
constexpr int FP_MAXNN = 100;

__global__ __launch_bounds__(KN_BLOCK)
void k_radius_finish(int n, int n_pad, int nsplit, const float* __restrict__ pd, const int* __restrict__ pi,
                     const int* __restrict__ pc, int* __restrict__ nbr /* [100][n_pad] */, int* __restrict__ nbr_cnt) {
    const int i = blockIdx.x * KN_BLOCK + threadIdx.x;
    if (i >= n) return;
    int cnt = merge_splits(pd, pi, pc, nsplit, FP_MAXNN, n_pad, i, [&](int r, float, int idx) { nbr[(size_t)r * n_pad + i] = idx; });
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
    const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    const float ux = nrm[3 * i], uy = nrm[3 * i + 1], uz = nrm[3 * i + 2];
    const int cnt = nbr_cnt[i];
    for (int r = 0; r < cnt; ++r) {
        const int j = nbr[(size_t)r * n_pad + i];
        if (j == i) continue;
        float dx = xyz[3 * j] - px, dy = xyz[3 * j + 1] - py, dz = xyz[3 * j + 2] - pz;
        float dist = sqrtf(dx * dx + (dy * dy + dz * dz));
        if (dist < 1e-8f) continue;
        float ex = dx / dist, ey = dy / dist, ez = dz / dist;
        float vx = uy * ez - uz * ey, vy = uz * ex - ux * ez, vz = ux * ey - uy * ex;   // v = u x d
        float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;   // w = u x v
        float njx = nrm[3 * j], njy = nrm[3 * j + 1], njz = nrm[3 * j + 2];
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
    const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    const int cnt = nbr_cnt[i];
    for (int r = 0; r < cnt; ++r) {
        const int j = nbr[(size_t)r * n_pad + i];
        if (nbr_out) nbr_out[(size_t)i * FP_MAXNN + r] = j;
        if (j == i) continue;
        float dx = xyz[3 * j] - px, dy = xyz[3 * j + 1] - py, dz = xyz[3 * j + 2] - pz;
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

struct ScanPlan { int n_pad, nt_pad, n_chunks, blocks_x, nsplit, chunks_per_split; };

ScanPlan make_scan_plan(int n) {
    ScanPlan p;
    p.n_pad = (int)align_up((size_t)n, KN_BLOCK);
    p.nt_pad = (int)align_up((size_t)n, KN_CH);
    p.n_chunks = p.nt_pad / KN_CH;
    p.blocks_x = p.n_pad / KN_BLOCK;
    int want = (2048 + p.blocks_x - 1) / p.blocks_x;
    int max_split = std::max(1, p.n_chunks / 16);
    p.nsplit = std::max(1, std::min(std::min(want, max_split), KN_MAXSPLIT));
    p.chunks_per_split = (p.n_chunks + p.nsplit - 1) / p.nsplit;
    p.nsplit = (p.n_chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    return p;
}

int run_scan(tdv_ctx* ctx, const float* d_xyz, int n, int k, float bound0, int timer, const ScanPlan& p,
             float** pd, int** pi, int** pc) {
    float* soa;
    TDV_TRY(ws_alloc(ctx, (size_t)3 * p.nt_pad, &soa));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * k * p.n_pad, pd));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * k * p.n_pad, pi));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * p.n_pad, pc));
    hipStream_t s = ctx->stream;
    k_aos_to_soa_pad2<<<(p.nt_pad + 255) / 256, 256, 0, s>>>(d_xyz, n, p.nt_pad, INFINITY, soa, soa + p.nt_pad, soa + 2 * (size_t)p.nt_pad);
    {
        ScopedTimer tm(ctx, timer);
        dim3 grid(p.blocks_x, p.nsplit);
#define TDV_REG_SCAN(KK) k_topk_scan_reg<KK><<<grid, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, soa, soa + p.nt_pad, soa + 2 * (size_t)p.nt_pad, \
                                                                     p.n_chunks, p.chunks_per_split, k, bound0, *pd, *pi, *pc)
        if (k <= 8) TDV_REG_SCAN(8);
        else if (k <= 16) TDV_REG_SCAN(16);
        else if (k <= 30) TDV_REG_SCAN(30);
        else if (k <= 32) TDV_REG_SCAN(32);
#undef TDV_REG_SCAN
        else
            k_topk_scan<false><<<grid, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, soa, soa + p.nt_pad, soa + 2 * (size_t)p.nt_pad,
                                                         p.n_chunks, p.chunks_per_split, k, bound0, *pd, *pi, *pc);
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
    float* pd; int *pi, *pc, *nbr;
    TDV_TRY(run_scan(ctx, d_xyz, n, kk, INFINITY, TDV_TIMER_KNN, p, &pd, &pi, &pc));
    TDV_TRY(ws_alloc(ctx, (size_t)kk * p.n_pad, &nbr));
    hipStream_t s = ctx->stream;
    if (d_knn && kk < k) TDV_HIP(ctx, hipMemsetAsync(d_knn, 0xff, (size_t)n * k * 4, s));
    if (kk == k)
        k_normals_finish<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, kk, p.nsplit, pd, pi, pc, nbr, d_normals, d_knn);
    else {
        // n < k: the lists are shorter than the caller's row stride; write rows through a strided pass
        int* knn_tmp = nullptr;
        if (d_knn) TDV_TRY(ws_alloc(ctx, (size_t)n * kk, &knn_tmp));
        k_normals_finish<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, kk, p.nsplit, pd, pi, pc, nbr, d_normals, knn_tmp);
        if (d_knn) TDV_HIP(ctx, hipMemcpy2DAsync(d_knn, (size_t)k * 4, knn_tmp, (size_t)kk * 4, (size_t)kk * 4, n, hipMemcpyDeviceToDevice, s));
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

int compute_fpfh_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_normals, int n, float radius,
                     float* d_desc, int* d_nbr, int* d_nbr_cnt) {
    if (!ctx || n < 0 || (n > 0 && (!d_xyz || !d_normals || !d_desc))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    const float r2 = radius * radius;  // registration.cpp:89
    const ScanPlan p = make_scan_plan(n);
    float* pd; int *pi, *pc, *nbr, *cnt; float* spfh;
    TDV_TRY(run_scan(ctx, d_xyz, n, FP_MAXNN, r2, TDV_TIMER_RADIUS, p, &pd, &pi, &pc));
    TDV_TRY(ws_alloc(ctx, (size_t)FP_MAXNN * p.n_pad, &nbr));
    TDV_TRY(ws_alloc(ctx, (size_t)p.n_pad, &cnt));
    TDV_TRY(ws_alloc(ctx, (size_t)n * 33, &spfh));
    hipStream_t s = ctx->stream;
    k_radius_finish<<<p.blocks_x, KN_BLOCK, 0, s>>>(n, p.n_pad, p.nsplit, pd, pi, pc, nbr, cnt);
    k_spfh<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, d_normals, n, p.n_pad, nbr, cnt, spfh);
    k_fpfh<<<p.blocks_x, KN_BLOCK, 0, s>>>(d_xyz, n, p.n_pad, nbr, cnt, spfh, d_desc, d_nbr);
    TDV_CHECK_LAUNCH(ctx);
    if (d_nbr_cnt) TDV_HIP(ctx, hipMemcpyAsync(d_nbr_cnt, cnt, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return TDV_OK;
}

}  // namespace tdv
