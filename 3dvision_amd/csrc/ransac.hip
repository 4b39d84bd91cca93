// RANSAC coarse alignment on gfx950.
//
// Replaces Registration::ransacRegistration (/root/reference/src/registration.cpp:204-295), which
// has no GPU entry point in the reference (src/pipeline.cpp:97-102 calls the CPU static directly).
// Sub-steps and their kernels:
//  (i)   feature correspondences (registration.cpp:216-232): csrc/fmatch.hip.
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
#include <cstring>
#include <algorithm>
#include <vector>

namespace tdv {

// ------------------------------------------------------------------ hypotheses
// pq layout: 8 floats per point: px py pz qx qy qz 0 0  (q = tgt[corr[i]]); padding points have
// p = 0 and q = +inf so that d2 = +inf and they are never inliers.
// A correspondence outside [0, nt) (caller-supplied lists are not trusted) raises *bad and reads target 0 instead of
// faulting; the host turns the flag into TDV_ERR_BAD_ARG at its first synchronisation.
// *pmax receives (integer atomic max on the bits of a non-negative float) the largest |source coordinate|, +inf for a
// non-finite one or a NaN in a matched target: it scales the rounding band of the fast scoring pass (k_ransac_score_fast).
__global__ void k_gather_pq(const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ corr,
                            int ns, int ns_pad, int nt, float* __restrict__ pq, int* __restrict__ bad, unsigned* __restrict__ pmax) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float am = 0.f;
    if (i < ns_pad) {
        float4 a, b;
        if (i < ns) {
            int c = corr[i];
            if ((unsigned)c >= (unsigned)nt) { *bad = 1; c = 0; }
            a = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], tgt[3 * c]);
            b = make_float4(tgt[3 * c + 1], tgt[3 * c + 2], 0.f, 0.f);
            am = fmaxf(fabsf(a.x), fmaxf(fabsf(a.y), fabsf(a.z)));
            if (!(am <= FLT_MAX)) am = INFINITY;      // NaN or inf
            // a NaN target coordinate makes d2 NaN, whose sign bit the fast pass would read as "inlier": such a cloud is
            // scored with the reference arithmetic throughout (an infinite one gives d2 = +inf in both and is harmless)
            if (a.w != a.w || b.x != b.x || b.y != b.y) am = INFINITY;
        } else {
            a = make_float4(0.f, 0.f, 0.f, INFINITY);
            b = make_float4(INFINITY, INFINITY, 0.f, 0.f);
        }
        reinterpret_cast<float4*>(pq)[2 * (size_t)i] = a;
        reinterpret_cast<float4*>(pq)[2 * (size_t)i + 1] = b;
    }
    // one atomic per workgroup: thousands of them on one address would cost more than the gather itself
    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((threadIdx.x & 63) == 0 && am > 0.f) atomicMax(&s_max, __float_as_uint(am));
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(pmax, s_max);
}

__device__ __forceinline__ float tau_mid_default(float sqrt_tau) { return sqrt_tau * sqrt_tau; }

// hyp layout: SoA [14][h_pad]: r00 r10 r20 r01 r11 r21 r02 r12 r22 t0 t1 t2 (column-major R).
// Invalid (skipped) iterations get NaN so that no comparison is ever true -> 0 inliers.
// Rows 12, 13 of hyp: the band of the fast scoring pass for this hypothesis (RansacBand below): mid and half-width of
// the d2 interval inside which the FMA arithmetic and the reference's arithmetic might disagree on `d2 < tau`.
__global__ void k_ransac_hypotheses(const float* __restrict__ pq, const int4* __restrict__ triples, int count, int h_pad,
                                    float* __restrict__ hyp, const unsigned* __restrict__ pmax, float sqrt_tau, int* __restrict__ counts) {
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= h_pad) return;
    counts[h] = 0;                       // the scoring kernel adds its point-splits' counts here (one memset launch less per batch)
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
    // RansacBand.  x_ref = fl((r0 px + (r1 py + r2 pz)) + t) and x_fma = fma(r0, px, fma(r1, py, fma(r2, pz, t))) are both
    // within gamma_4 resp. gamma_3 of the real value relative to A = |r0||px| + |r1||py| + |r2||pz| + |t| (u = 2^-24), so
    // they differ by at most 7.1 u A per component; the subtraction of q, the squared norm (three terms, either order)
    // and the square root add relative errors of a few u.  In distance: |sqrt(d2_ref) - sqrt(d2_fma)| <= 12.3 u A + 12 u s
    // for distances up to 2 s, s = sqrt(tau).  E = 16 u (A + s) with A bounded over all points by the largest |source
    // coordinate|; outside [(s - E)^2, (s + E)^2] (widened by 1e-6) both arithmetics give the same side of `d2 < tau`.
    // A band that is not small against s (coordinates far from the origin, non-finite data, an invalid hypothesis) is
    // stored as NaN: every chunk of such a lane's wave is then scored with the reference arithmetic.
    // (a skipped iteration or a padding lane has no band at all - half = 0 - so that it never makes its wave score a chunk
    // twice; its count is garbage and the host never reads it)
    float mid = tau_mid_default(sqrt_tau), half = valid ? __builtin_nanf("") : 0.f;
    if (valid) {
        const float P = __uint_as_float(*pmax);
        float A = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) A = fmaxf(A, (fabsf(o[c]) + fabsf(o[3 + c]) + fabsf(o[6 + c])) * P + fabsf(o[9 + c]));
        const float E = (9.5367431640625e-07f * A + 9.5367431640625e-07f * sqrt_tau) * 1.0001f;
        if (E < 0.25f * sqrt_tau) {
            const float lo = sqrt_tau - E, hi = sqrt_tau + E;
            const float tlo = lo * lo * (1.0f - 1e-6f), thi = hi * hi * (1.0f + 1e-6f);
            mid = 0.5f * (tlo + thi);
            half = 0.5f * (thi - tlo) * (1.0f + 1e-5f) + mid * 1e-6f;
        }
    }
    hyp[(size_t)12 * h_pad + h] = mid;
    hyp[(size_t)13 * h_pad + h] = half;
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

// The same counts from half the arithmetic.  Parity forbids FMA contraction in the reference's expression, but only
// its RESULT - which side of tau each d2 falls on - has to be reproduced.  This pass evaluates every (hypothesis, point)
// with fused multiply-adds (15 packed instructions per two points instead of 26) and classifies by the sign of
// d2_fma - mid; a test whose d2_fma lies inside the hypothesis' rounding band (RansacBand in k_ransac_hypotheses: the
// two arithmetics provably agree outside it) makes its wave score that chunk of 8 points again with the reference
// arithmetic.  The band is about 1e-3 of the threshold wide at metre-scale coordinates, so this happens for a percent or
// so of the chunks; counts are identical to k_ransac_score's (tests/test_gpu_ransac.py holds the two against each other
// and against the oracle).
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__global__ __launch_bounds__(RS_BLOCK)
void k_ransac_score_fast(const float* __restrict__ hyp, int h_pad, const float* __restrict__ pq2,
                         int n_pchunks, int pchunks_per_split, float tau, int* __restrict__ counts, unsigned long long* __restrict__ rescored) {
    const int split = blockIdx.y;
    const int c0 = split * pchunks_per_split;
    const int c1 = min(n_pchunks, c0 + pchunks_per_split);
    const int base = blockIdx.x * RS_BLOCK + threadIdx.x;
    v2f r[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) { const float t = hyp[(size_t)e * h_pad + base]; r[e] = (v2f){t, t}; }
    const float mid = hyp[(size_t)12 * h_pad + base], half = hyp[(size_t)13 * h_pad + base];
    const v2f nmid = {-mid, -mid};
    int cnt = 0;
    unsigned n_rescored = 0;     // wave-uniform
    for (int c = c0; c < c1; ++c) {
        const float* __restrict__ g = pq2 + (size_t)c * (6 * RS_PCH);  // RS_PCH points = RS_PCH/2 records of 12 floats, wave-uniform
        float v[6 * RS_PCH];
#pragma unroll
        for (int e = 0; e < 6 * RS_PCH; ++e) v[e] = g[e];
        float m = INFINITY;      // smallest |d2_fma - mid| of this lane in the chunk
        unsigned sgn = 0u;       // the signs of d2_fma - mid, shifted in one per test (1 = below mid = inlier)
        int cf = 0;
#pragma unroll
        for (int p = 0; p < RS_PCH / 2; ++p) {
            const v2f px = {v[12 * p + 0], v[12 * p + 1]}, py = {v[12 * p + 2], v[12 * p + 3]}, pz = {v[12 * p + 4], v[12 * p + 5]};
            const v2f qx = {v[12 * p + 6], v[12 * p + 7]}, qy = {v[12 * p + 8], v[12 * p + 9]}, qz = {v[12 * p + 10], v[12 * p + 11]};
            const v2f dx = fma2(r[0], px, fma2(r[3], py, fma2(r[6], pz, r[9]))) - qx;
            const v2f dy = fma2(r[1], px, fma2(r[4], py, fma2(r[7], pz, r[10]))) - qy;
            const v2f dz = fma2(r[2], px, fma2(r[5], py, fma2(r[8], pz, r[11]))) - qz;
            // d2_fma - mid as one chain ending in -mid: its own rounding, at most 3 u mid = 1.5 u s in distance, sits inside
            // the 3.7 u A + 4 u s that the band's E keeps in reserve over the proven bound
            const v2f t = fma2(dx, dx, fma2(dy, dy, fma2(dz, dz, nmid)));
            m = fminf(fminf(m, fabsf(t.x)), fabsf(t.y));                       // one v_min3_f32; a NaN (invalid hypothesis) leaves m alone: half is NaN there
            sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(t.x), 31);      // sgn = sgn << 1 | sign(t.x)
            sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(t.y), 31);
        }
        cf = __popc(sgn);
        if (__any(!(m >= half))) {      // some lane of the wave is inside its band (or has none): the reference arithmetic decides this chunk
            ++n_rescored;
            cf = 0;
#pragma unroll
            for (int p = 0; p < RS_PCH / 2; ++p) {
                const v2f px = {v[12 * p + 0], v[12 * p + 1]}, py = {v[12 * p + 2], v[12 * p + 3]}, pz = {v[12 * p + 4], v[12 * p + 5]};
                const v2f qx = {v[12 * p + 6], v[12 * p + 7]}, qy = {v[12 * p + 8], v[12 * p + 9]}, qz = {v[12 * p + 10], v[12 * p + 11]};
                const v2f x = (r[0] * px + (r[3] * py + r[6] * pz)) + r[9];
                const v2f y = (r[1] * px + (r[4] * py + r[7] * pz)) + r[10];
                const v2f z = (r[2] * px + (r[5] * py + r[8] * pz)) + r[11];
                const v2f dx = x - qx, dy = y - qy, dz = z - qz;
                const v2f d2 = dx * dx + (dy * dy + dz * dz);
                cf += (d2.x < tau) ? 1 : 0;
                cf += (d2.y < tau) ? 1 : 0;
            }
        }
        cnt += cf;
    }
    atomicAdd(&counts[base], cnt);
    // statistics only (tdv_ctx_last_ransac_rescore): one atomic per workgroup
    __shared__ unsigned s_rescored;
    if (threadIdx.x == 0) s_rescored = 0u;
    __syncthreads();
    if (n_rescored && (threadIdx.x & 63) == 0) atomicAdd(&s_rescored, n_rescored);
    __syncthreads();
    if (threadIdx.x == 0 && s_rescored) atomicAdd(rescored, (unsigned long long)s_rescored);
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
    // one device block: [0] bad index flag, [1] largest |source coordinate|, [2..3] rescored chunks (u64), [16..27] the winning
    // hypothesis, [32..35] its error sum and inlier count (2 doubles): one memset at the start, one copy back at the end
    int* d_bad = nullptr;
    TDV_TRY(ws_alloc(ctx, 40, &d_bad));
    unsigned* d_pmax = reinterpret_cast<unsigned*>(d_bad + 1);
    unsigned long long* d_rescored = reinterpret_cast<unsigned long long*>(d_bad + 2);
    TDV_HIP(ctx, hipMemsetAsync(d_bad, 0, 160, s));
    double wave_chunks = 0.0;    // wave x chunk pairs scored by the fast pass in this call
    k_gather_pq<<<(ns_pad + 255) / 256, 256, 0, s>>>(d_src, d_tgt, d_corr, ns, ns_pad, nt, pq, d_bad, d_pmax);
    // sqrt(tau) rounded up: the boundary of `d2 < tau` in distance, for the band of the fast scoring pass
    const float sqrt_tau = std::nextafter((float)std::sqrt((double)tau), INFINITY);
    static const bool score_exact_env = getenv("TDV_RANSAC_SCORE") && !strcmp(getenv("TDV_RANSAC_SCORE"), "exact");
    const bool score_fast = !score_exact_env && !ctx->ransac_score_exact;
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
        TDV_TRY(ws_alloc(ctx, (size_t)14 * h_pad, &hyp[q]));
        TDV_TRY(ws_alloc(ctx, (size_t)h_pad, &counts[q]));
        TDV_TRY(ws_alloc(ctx, (size_t)batch, &d_tri[q]));
    }
    const int rblocks = (ns + 255) / 256;
    TDV_TRY(ws_alloc(ctx, (size_t)2 * rblocks, &slabs));
    d_best12 = reinterpret_cast<float*>(d_bad + 16);
    d_out2 = reinterpret_cast<double*>(d_bad + 32);
    // pinned: 2 x triples (int4 * batch) | 2 x counts (int * batch) | best12 (12 floats) | out2 (2 doubles)
    const size_t sz_tri = align_up((size_t)batch * 16, 64), sz_cnt = align_up((size_t)batch * 4, 64);
    const size_t pin_b12 = 2 * sz_tri + 2 * sz_cnt, pin_bad = pin_b12 + 192, pin_total = pin_bad + 64;   // pin_b12: the 160-byte result block
    TDV_TRY(pin_reserve(ctx, pin_total));
    int* h_bad = reinterpret_cast<int*>(ctx->pin + pin_bad);
    *h_bad = 0;
    TDV_HIP(ctx, hipMemcpyAsync(h_bad, d_bad, 4, hipMemcpyDeviceToHost, s));   // lands before the first batch's counts
    int4* h_tri[2] = {reinterpret_cast<int4*>(ctx->pin), reinterpret_cast<int4*>(ctx->pin + sz_tri)};
    int* h_cnt[2] = {reinterpret_cast<int*>(ctx->pin + 2 * sz_tri), reinterpret_cast<int*>(ctx->pin + 2 * sz_tri + sz_cnt)};
    const int* h_block = reinterpret_cast<const int*>(ctx->pin + pin_b12);      // host copy of d_bad[0..40): same layout
    const float* h_b12 = reinterpret_cast<const float*>(h_block + 16);
    const double* h_o2 = reinterpret_cast<const double*>(h_block + 32);
    hipEvent_t ev[2] = {event_acquire(ctx), event_acquire(ctx)};   // from the ctx's pool: no create/destroy per call
    if (!ev[0] || !ev[1]) { event_release(ctx, ev[0]); event_release(ctx, ev[1]); return TDV_ERR_OOM; }

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
        k_ransac_hypotheses<<<(h_pad + 255) / 256, 256, 0, s>>>(pq, d_tri[q], cnt, h_pad, hyp[q], d_pmax, sqrt_tau, counts[q]);
        const int hb = (int)(align_up((size_t)cnt, RS_HYP_PER_BLOCK) / RS_HYP_PER_BLOCK);
        {
            ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
            if (score_fast) {
                k_ransac_score_fast<<<dim3(hb, psplit), RS_BLOCK, 0, s>>>(hyp[q], h_pad, pq2, n_pchunks, pchunks_per_split, tau, counts[q], d_rescored);
                wave_chunks += (double)hb * (RS_BLOCK / 64) * (double)n_pchunks;
            }
            else k_ransac_score<<<dim3(hb, psplit), RS_BLOCK, 0, s>>>(hyp[q], h_pad, pq2, n_pchunks, pchunks_per_split, tau, counts[q]);
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
        if (*h_bad) { std::snprintf(ctx->err, sizeof(ctx->err), "ransac: a correspondence index lies outside [0, %d)", nt); status = TDV_ERR_BAD_ARG; break; }
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
    ctx->last_ransac_rescore = -1.0;
    for (int q = 0; q < 2; ++q) event_release(ctx, ev[q]);
    if (status != TDV_OK) return status;
    out->iterations_run = done_iters;
    // the result block comes back in one copy: winning hypothesis, its error sum and count, the fast pass's statistics
    const unsigned long long* h_res = reinterpret_cast<const unsigned long long*>(h_block + 2);
    const bool want_stats = score_fast && wave_chunks > 0.0;
    if (best_iter < 0 && want_stats) {
        TDV_HIP(ctx, hipMemcpyAsync(ctx->pin + pin_b12, d_bad, 160, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipStreamSynchronize(s));
    }
    if (best_iter >= 0) {
        k_ransac_rmse_partial<<<rblocks, 256, 0, s>>>(pq, ns, d_best12, tau, slabs);
        k_ransac_rmse_final<<<1, 256, 0, s>>>(slabs, rblocks, d_out2);
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipMemcpyAsync(ctx->pin + pin_b12, d_bad, 160, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipStreamSynchronize(s));
        if (want_stats) ctx->last_ransac_rescore = (double)*h_res / wave_chunks;
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
    } else if (want_stats) {
        ctx->last_ransac_rescore = (double)*h_res / wave_chunks;
    }
    return TDV_OK;
}

}  // namespace tdv
