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
__device__ __forceinline__ void ransac_hypothesis_lane(const float* __restrict__ pq, const int4 tr, const bool valid, const int h, const int h_pad,
                                                       float* __restrict__ hyp, const unsigned* __restrict__ pmax, const float sqrt_tau, const float band_u) {
    float o[12];
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
        const float E = (band_u * A + band_u * sqrt_tau) * 1.0001f;
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
__global__ void k_ransac_hypotheses(const float* __restrict__ pq, const int4* __restrict__ triples, int count, int h_pad,
                                    float* __restrict__ hyp, const unsigned* __restrict__ pmax, float sqrt_tau, int* __restrict__ counts,
                                    float band_u /* E = band_u (A + s): 16 u for the FMA pass, 24 u for the matrix-core pass */) {
    int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= h_pad) return;
    counts[h] = 0;                       // the scoring kernel adds its point-splits' counts here (one memset launch less per batch)
    bool valid = false;
    int4 tr = make_int4(0, 0, 0, 0);
    if (h < count) { tr = triples[h]; valid = tr.w != 0; }
    ransac_hypothesis_lane(pq, tr, valid, h, h_pad, hyp, pmax, sqrt_tau, band_u);
}

__device__ __forceinline__ unsigned long long shfl_u64_down(unsigned long long v, int o) {
    const unsigned lo = __shfl_down((unsigned)v, o, 64), hi = __shfl_down((unsigned)(v >> 32), o, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// ------------------------------------------------------------------ scoring
#ifndef RS_BLOCK_VALUE
#define RS_BLOCK_VALUE 1024
#endif
#ifndef RS_XCD_R
#define RS_XCD_R 8           // XCD groups over the point ranges (1, 2, 4 or 8); 8 / RS_XCD_R groups over the hypothesis blocks
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
// Exact bail-out (RansacPlan, below).  One dispatch carries up to two JOBS, decoded from the workgroup id:
//   job A (ids below g1): a batch's hypotheses, one block of RS_BLOCK per id, over the first plan[0] chunks of the points
//                         (plan == nullptr: over all of them) cut EVENLY into a.ps ranges - every workgroup of the job has the
//                         same amount of work, whatever the prefix (the first version cut the whole point range and let the
//                         workgroups past the prefix exit: 3,648 busy workgroups on 512 slots, an eighth of the last round idle);
//   job B (the ids after): phase 2 of the PREVIOUS batch - the hypotheses its selection listed (plan[1] of them: those that
//                         can still beat the best count known) over the chunks its phase 1 left out, in ranges of plan[3]
//                         chunks (= what a job-A workgroup of that batch walked, so all workgroups of a dispatch cost the same).
// By default the two jobs are dispatched one after the other (job B alone, see ransac_run_dev); with TDV_RANSAC_MERGE=1 job B
// rides behind the NEXT batch's job A.
struct ScoreJob {
    const float* hyp; int* counts; const int* plan; const int* list;
    int hb;      // hypothesis blocks (A: of the batch; B: upper bound - the real number comes from plan[1])
    int ps;      // A: point ranges
};
// One block of hypotheses (lane = hypothesis `base`, -1: none) over the chunks [c0, c1) of the point pairs; returns the lane's
// inlier count, adds the point PAIRS the wave scored twice to n_rescored (wave-uniform; RS_PCH / 2 per chunk that was re-scored whole).
// (A finer band test - per PAIR of points instead of per chunk of eight - was built for the batch's small clouds, half of whose
// hypotheses are decent, so that some of a wave's 512 tests per chunk nearly always sit at the threshold and 49 % of the chunks are
// re-scored (k_rb_score, C5): the share stayed at 49 % and the pass got slower, 2.52 against 2.30 ms.  Removed in round 4.)
// ADAPT: a wave that had to re-score three of its first eight chunks stops trying the FMA pass and scores the rest of its range with
// the reference arithmetic alone (28 ops per test instead of 16.6 + 28: cheaper from a re-scoring share of 0.4 on).  Measured on C5:
// per-chunk band test 2.30 ms, per-pair band test 2.52 ms (the share stays at 49 % even for 128 tests), adaptive exact: see k_rb_score.
template <bool ADAPT = false>
__device__ __forceinline__ int score_range_fast(const float* __restrict__ hyp, const int h_pad, const int base, const float* __restrict__ pq2,
                                                const int c0, const int c1, const float tau, unsigned& n_rescored) {
    v2f r[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) { const float t = base >= 0 ? hyp[(size_t)e * h_pad + base] : __builtin_nanf(""); r[e] = (v2f){t, t}; }
    const float mid = base >= 0 ? hyp[(size_t)12 * h_pad + base] : tau, half = base >= 0 ? hyp[(size_t)13 * h_pad + base] : 0.f;   // a lane without a hypothesis has no band
    const v2f nmid = {-mid, -mid};
    int cnt = 0;
    int c_fast_end = c1;         // ADAPT: where the FMA pass gives up (wave-uniform)
    for (int c = c0; c < c_fast_end; ++c) {
        if (ADAPT && c == c0 + 8 && n_rescored >= 3u * (RS_PCH / 2)) { c_fast_end = c; break; }      // (n_rescored counts pairs: three whole chunks)
        const float* __restrict__ g = pq2 + (size_t)c * (6 * RS_PCH);  // RS_PCH points = RS_PCH/2 records of 12 floats, wave-uniform
        float v[6 * RS_PCH];
#pragma unroll
        for (int e = 0; e < 6 * RS_PCH; ++e) v[e] = g[e];
        float m = INFINITY;      // smallest |d2_fma - mid| of this lane in the chunk
        unsigned sgn = 0u;       // the signs of d2_fma - mid, shifted in one per test (1 = below mid = inlier)
        int cf = 0;
        v2f tt[RS_PCH / 2];      // d2_fma - mid of every test, kept for the re-scoring branch (which pairs are inside a band)
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
            tt[p] = t;
            m = fminf(fminf(m, fabsf(t.x)), fabsf(t.y));            // one v_min3_f32; a NaN (invalid hypothesis) leaves m alone: half is NaN there
            sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(t.x), 31);      // sgn = sgn << 1 | sign(t.x)
            sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(t.y), 31);
        }
        cf = __popc(sgn);
        if (__any(!(m >= half))) {      // some lane of the wave is inside its band (or has none): the reference arithmetic decides
            // ... the PAIRS of points that some lane has inside its band (round 4; until then the whole chunk: a hit is nearly always one
            // (lane, point), so three quarters of the second scoring were spent on pairs nobody doubted).  n_rescored counts pairs.
#pragma unroll
            for (int p = 0; p < RS_PCH / 2; ++p) {
                const float mp = fminf(fminf(INFINITY, fabsf(tt[p].x)), fabsf(tt[p].y));       // (NaN - an invalid hypothesis - leaves INFINITY)
                if (!ADAPT && !__any(!(mp >= half))) continue;                                 // (the small-cloud pass keeps re-scoring whole chunks: its ADAPT rule counts them)
                ++n_rescored;
                const v2f px = {v[12 * p + 0], v[12 * p + 1]}, py = {v[12 * p + 2], v[12 * p + 3]}, pz = {v[12 * p + 4], v[12 * p + 5]};
                const v2f qx = {v[12 * p + 6], v[12 * p + 7]}, qy = {v[12 * p + 8], v[12 * p + 9]}, qz = {v[12 * p + 10], v[12 * p + 11]};
                const v2f x = (r[0] * px + (r[3] * py + r[6] * pz)) + r[9];
                const v2f y = (r[1] * px + (r[4] * py + r[7] * pz)) + r[10];
                const v2f z = (r[2] * px + (r[5] * py + r[8] * pz)) + r[11];
                const v2f dx = x - qx, dy = y - qy, dz = z - qz;
                const v2f d2 = dx * dx + (dy * dy + dz * dz);
                // the pair's two sign bits in sgn: test 2p at bit RS_PCH - 1 - 2p, test 2p + 1 right below it
                cf -= __popc((sgn >> (RS_PCH - 2 - 2 * p)) & 3u);
                cf += (d2.x < tau) ? 1 : 0;
                cf += (d2.y < tau) ? 1 : 0;
            }
        }
        cnt += cf;
    }
    if (ADAPT) {
        for (int c = c_fast_end; c < c1; ++c) {          // the reference arithmetic alone (k_ransac_score's loop)
            const float* __restrict__ g = pq2 + (size_t)c * (6 * RS_PCH);
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
    }
    return cnt;
}

// Occupancy (round 4).  The loop keeps a chunk's 48 floats in SGPRs and the compiler took 106 of them: 7 waves per SIMD on paper, but a
// workgroup is 16 waves (4 per SIMD), so ONE workgroup per CU - 4 waves per SIMD to hide the scalar loads of a loop whose waves walk the
// same chunks nearly in step.  Capped at the 8-wave budget (80 SGPRs; 43 values spilled to VGPR lanes, all outside the chunk loop) two
// workgroups share a CU: 1,098 -> 1,211 steps/s of bench.py on the same box, `frac` 0.70 -> 0.78.  Also measured: chunks of 4 points
// with it (16 spills: 1,209-1,223, within a point of this), chunks of 2 (1,140), 512-thread workgroups (1,177), chunks of 4 without it (1,034);
// and, at 8 waves, a loop that loads the NEXT pair of points while it computes one (12 SGPRs in flight instead of 48, two waits per chunk
// instead of one wait on everything): 1,196-1,210 against 1,213-1,221 on one box - with two workgroups per CU the scalar loads are hidden already.
__global__ __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_ransac_score_fast(const ScoreJob a, const ScoreJob b, const int g1, const int h_pad, const float* __restrict__ pq2,
                         const int n_pchunks, const float tau, unsigned long long* __restrict__ rescored) {
    const int id = blockIdx.x;
    unsigned n_rescored = 0;     // wave-uniform
    unsigned chunks = 0;         // chunks this workgroup walked (workgroup-uniform)
    if (id < g1) {
        // XCD-aware deal (round 4).  An XCD (workgroup id mod 8) has its own L2.  Round 3 gave every XCD an eighth of the hypothesis blocks
        // and ALL point ranges: every L2 pulled the whole pair array (PMC: 38.4 MB of HBM traffic per dispatch for 4.8 MB of pairs).  Now
        // the eight XCDs form RS_XCD_R groups over the point ranges x 8 / RS_XCD_R groups over the hypothesis blocks: an L2 holds
        // 1 / RS_XCD_R of the pairs and 1 / (8 / RS_XCD_R) of the hypotheses (52 B each).
        const int xcd = id & 7, k = id >> 3;
        constexpr int XR = RS_XCD_R, XH = 8 / RS_XCD_R;
        const int hbl = (a.hb + XH - 1) / XH;
        const int hblock = (k % hbl) * XH + xcd / XR, split = (k / hbl) * XR + xcd % XR;
        if (hblock >= a.hb || split >= a.ps) return;                // (counts that are not multiples of the group sizes: the last groups are short)
        const int c_split = a.plan ? a.plan[0] : n_pchunks;
        const int per = (c_split + a.ps - 1) / a.ps;
        const int c0 = split * per, c1 = min(c_split, c0 + per);
        if (c0 >= c1) return;                                    // workgroup-uniform (ids past hb * ps, a prefix shorter than ps chunks)
        const int base = hblock * RS_BLOCK + threadIdx.x;
        const int cnt = score_range_fast(a.hyp, h_pad, base, pq2, c0, c1, tau, n_rescored);
        atomicAdd(&a.counts[base], cnt);
        chunks = (unsigned)(c1 - c0);
    } else {
        // Few hypothesis blocks are left, and the points reach the scalar cache through the XCD's L2: a workgroup that walks a
        // point range alone misses on every chunk (measured 1.2 ms for 5 % of the work).  Items (point range, surviving block)
        // are therefore dealt so that an XCD (workgroup id mod 8) runs ALL surviving blocks of one range next to each other; a
        // workgroup takes every (job-B workgroups / 8)-th item of its XCD, so any grid of at least 8 workgroups covers any
        // number of survivors and any prefix.
        const int n_list = b.plan[1];
        const int n_blk = (n_list + RS_BLOCK - 1) / RS_BLOCK;
        if (n_blk == 0) return;
        const int j0 = id - g1, xcd = j0 & 7, stride = ((int)gridDim.x - g1) >> 3;
        const int c_split = b.plan[0], per = max(b.plan[3], 1);
        const int ranges = (n_pchunks - c_split + per - 1) / per;
        for (int t = j0 >> 3; ; t += stride) {
            const int split = (t / n_blk) * 8 + xcd, hblock = t % n_blk;
            if (split >= ranges) break;                              // workgroup-uniform
            const int c0 = c_split + split * per, c1 = min(n_pchunks, c0 + per);
            const int slot = hblock * RS_BLOCK + threadIdx.x;
            const int base = slot < n_list ? b.list[slot] : -1;
            const int cnt = score_range_fast(b.hyp, h_pad, base, pq2, c0, c1, tau, n_rescored);
            if (base >= 0) atomicAdd(&b.counts[base], cnt);
            chunks += (unsigned)(c1 - c0);
        }
        if (!chunks) return;
    }
    // statistics only (tdv_ctx_last_ransac_rescore / _scored): two atomics per workgroup — (wave, chunk) pairs scored, and scored twice
    __shared__ unsigned s_rescored;
    if (threadIdx.x == 0) s_rescored = 0u;
    __syncthreads();
    if (n_rescored && (threadIdx.x & 63) == 0) atomicAdd(&s_rescored, n_rescored);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_rescored) atomicAdd(rescored, (unsigned long long)s_rescored);
        atomicAdd(rescored + 1, (unsigned long long)(RS_BLOCK / 64) * (unsigned long long)chunks);
    }
}

// RansacPlan — exact bail-out.  The loop of ransacRegistration (registration.cpp:284-290) uses an iteration's inlier count only
// to ask whether it beats the best so far (strictly) — the early exit `fitness > confidence` can fire only on such a new best,
// the loop having stopped otherwise when the earlier best passed it.  So a hypothesis whose count over the first chunks plus
// ALL the remaining points cannot exceed the best count of the EARLIER batches needs no exact count: it is scored over the
// prefix only, and whatever partial count the host reads for it compares as the true one would.  The best hypothesis itself
// always survives, so the returned transform, inlier count, fitness, rmse and iteration are the reference's.  The running best
// stays on the device (no host round trip between batches): plan = { chunks in phase 1, survivors, best count so far }.
// Used only when the caller asked for no per-iteration trace.
// state[0] = best count known so far (a lower bound of the best count of every batch already enqueued: full counts of the
// batches that are complete, prefix counts of the one whose phase 2 is still to run).  plan = { chunks in phase 1, survivors,
// largest PREFIX count of this batch, chunks per workgroup } - one plan per batch buffer, the state shared.
__global__ void k_ransac_plan(int* __restrict__ state, int* __restrict__ plan, int ns, int n_pchunks, int ps, int drop_permille) {
    const int best = state[0];
    int c_split = n_pchunks;
    const int rest = best - max((int)((long long)best * drop_permille / 1000), 1);   // points left to phase 2: a hypothesis with under that share of the best count in the prefix is dropped
    if (rest >= ns / 8)                                      // (below an eighth of the points a second phase costs more than it saves)
        c_split = min(n_pchunks, (ns - rest + RS_PCH - 1) / RS_PCH);
    plan[0] = c_split; plan[1] = 0; plan[2] = 0; plan[3] = (c_split + ps - 1) / ps;
}
// largest count of a batch (prefix counts after phase 1, full counts after phase 2) -> *dst by atomic max, one atomic per
// workgroup (one per wave on the same address cost 12 us for a 65,536-hypothesis batch)
__global__ __launch_bounds__(1024)
void k_ransac_best(const int4* __restrict__ triples, int count, const int* __restrict__ counts, int* __restrict__ dst) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    int c = (h < count && triples[h].w != 0) ? counts[h] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c = max(c, __shfl_xor(c, off, 64));
    __shared__ int s_max[16];
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) c = max(c, s_max[w]);
        if (c > 0) atomicMax(dst, c);
    }
}
// Which hypotheses of a batch go on to phase 2.  With rest = the points phase 1 left out, ub = count + rest bounds a
// hypothesis' full count from above.  It is dropped when it provably is neither the result nor the iteration the loop of
// registration.cpp:284-290 stops at:
//   (a) ub <= best of the EARLIER batches.  It cannot be a new best at its iteration (strict >, :284), and the early exit
//       (:290) can only fire on a new best: had an earlier iteration reached `fitness > confidence` the loop would have ended
//       there.  Ties with the earlier best lose by the strict comparison, so <= is enough.  [round 2]
//   (b) ub < L, L = the largest PREFIX count inside this very batch, and float(ub)/ns is not > confidence.  Some hypothesis
//       h* of the batch has a full count >= L > ub, so this one is not the final result whatever the order of the two (strict
//       <: with ub == L and h* LATER than it, a tie would go to the earlier iteration, i.e. to the dropped one).  It could
//       still be a new best at its own iteration when h* comes later, and an exit firing there would return it - hence the
//       second condition: with no count up to ub passing the confidence test, the exit cannot fire on it.  Conversely an exit
//       that fires at a kept iteration e returns e itself: anything earlier with at least its count would have ended the loop
//       before, and every dropped iteration has a count below the confidence bar that e passed.  [round 3]
// Either way the counts of the dropped hypotheses stay partial and compare as the true ones would: below the result's.
__global__ void k_ransac_select(const int4* __restrict__ triples, int count, const int* __restrict__ counts, int ns, float confidence,
                                const int* __restrict__ state, int* __restrict__ plan, int* __restrict__ list) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    const int c_split = plan[0], best = state[0], in_batch = plan[2];
    const int rest = max(0, ns - min(ns, c_split * RS_PCH));     // (the padding past ns is never an inlier)
    bool keep = h < count && triples[h].w != 0 && rest > 0;
    if (keep) {
        const int ub = counts[h] + rest;
        const bool passes = static_cast<float>(ub) / static_cast<float>((size_t)ns) > confidence;   // registration.cpp:281,290 on the bound
        keep = ub > best && (ub >= in_batch || passes);
    }
    const unsigned long long m = __ballot(keep);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    int at = 0;
    if (lane == 0) at = atomicAdd(&plan[1], __popcll(m));
    at = __shfl(at, 0, 64);
    if (keep) list[at + __popcll(m & ((1ull << lane) - 1ull))] = h;
}

#ifdef TDV_STUDY
// ------------------------------------------------------------------ scoring on the matrix cores (A/B variant, not the default)
// R p + t is a [3H x 4] x [4 x N] product, so the transform can run on the matrix cores (v_mfma_f32_32x32x2_f32, twice for
// K = 4 with a row of ones under the points for t) and leave the vector ALUs the subtraction of q, the squared norm and the
// classification: 27 vector instructions per 32 points x 10 hypotheses (320 tests) instead of 73 per 8 points x 64
// hypotheses (512 tests).  Built, parity-green (tests/test_gpu_ransac.py runs every scoring test in this mode too) and
// MEASURED SLOWER than k_ransac_score_fast: 5.4 ms against 4.0 ms per 65,536 hypotheses x 200k points on the same box
// (profiles/r2/history/ransac_score_matrix_cores.md).  The probes recorded there show why: the f32 matrix instruction and
// the vector instructions of a SIMD do not overlap — the kernel's time is the SUM of its matrix time (3.1 ms alone) and its
// vector time, from one wave or from four per SIMD — and the K = 4 product spends a quarter of its multiply-adds on the
// constant row and a sixteenth on the unused accumulator row, so per test the matrix pipe is slower than nine packed FMAs.
// Kept behind TDV_RANSAC_SCORE_MATRIX / TDV_RANSAC_SCORE=mfma as the record of that experiment.
//
// Counts stay the reference's: the classification is the band scheme of k_ransac_score_fast (sign of d2 - mid outside the
// rounding band, the reference arithmetic inside it), with the band widened from 16 u to 24 u (A + s) for the accumulation
// of the matrix core — taken as at most one rounding per product and per addition of the K = 4 chain, i.e. within gamma_8 of
// the real value where the FMA chain is within gamma_3: 5 u A more per component, 8.7 u A in distance; an assumption about
// the hardware's arithmetic that only the count-for-count tests against the exact kernel back — and taken as the union over
// the wave's 10 hypotheses.
//
// Accumulator layout (32 x 32 tile, 16 registers per lane): lane l holds column l % 32 (a point), rows
// 8 (v / 4) + 4 (l / 32) + v % 4 for v = 0..15.  Rows are assigned so that the x, y, z of one (hypothesis, point) meet in
// one lane and two hypotheses share aligned register pairs (packed f32 operations): each half of the wave owns 5
// hypotheses a..e: v0..5 = ax bx ay by az bz, v6..11 = cx dx cy dy cz dz, v12..14 = ex ey ez, v15 unused.
typedef float v16f __attribute__((ext_vector_type(16)));
#ifndef RM_WAVES_VALUE
#define RM_WAVES_VALUE 8
#endif
constexpr int RM_WAVES = RM_WAVES_VALUE;          // hypothesis groups per workgroup, walking the same points
constexpr int RM_HPW = 10;           // hypotheses per wave
constexpr int RM_REC_FLOATS = 1280;  // per record of 4 tiles (128 points): [b0 | b1 | qx | qy | qz][lane][tile] — one 16-B load per lane and array
constexpr unsigned long long RM_E_OF_V = 0xF444323232101010ull, RM_C_OF_V = 0x0210221100221100ull;   // nibble v: hypothesis a..e, component
typedef float v4f __attribute__((ext_vector_type(4)));

// B operands as the lanes read them: lane l of tile j holds b0 = (l < 32 ? px : py), b1 = (l < 32 ? pz : 1) of point l % 32, and q of that point
__global__ void k_pack_pq3(const float* __restrict__ pq, int ns_pad, int n_rec, float* __restrict__ pq3) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (point, half)
    if (idx >= n_rec * 256) return;
    const int i = idx >> 1, hf = idx & 1;
    float px = 0.f, py = 0.f, pz = 0.f, qx = INFINITY, qy = INFINITY, qz = INFINITY;   // padding: never an inlier
    if (i < ns_pad) { const float* a = pq + (size_t)i * 8; px = a[0]; py = a[1]; pz = a[2]; qx = a[3]; qy = a[4]; qz = a[5]; }
    float* o = pq3 + (size_t)(i >> 7) * RM_REC_FLOATS + (hf * 32 + (i & 31)) * 4 + ((i >> 5) & 3);
    o[0] = hf ? py : px; o[256] = hf ? 1.f : pz; o[512] = qx; o[768] = qy; o[1024] = qz;
}

__global__ __launch_bounds__(64 * RM_WAVES)
void k_ransac_score_mfma(const float* __restrict__ hyp, int h_pad, const float* __restrict__ pq3, int n_rec, int rec_per_split,
                         float tau, int* __restrict__ counts, unsigned long long* __restrict__ rescored) {
    __shared__ float s_hyp[RM_WAVES][RM_HPW][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hf = lane >> 5, col = lane & 31;
    const int hb = (blockIdx.x * RM_WAVES + wave) * RM_HPW;
    const int g0 = blockIdx.y * rec_per_split, g1 = min(n_rec, g0 + rec_per_split);
    // A operand: lane l supplies row l % 32, k = l / 32 (first instruction: k = 0, 1; second: k = 2, 3 with t as column 3)
    float a0 = 0.f, a1 = 0.f;
    {
        const int v = 4 * (col >> 3) + (col & 3), hfrow = (col >> 2) & 1;
        const int e = (int)((RM_E_OF_V >> (4 * v)) & 15), c = (int)((RM_C_OF_V >> (4 * v)) & 15);
        const int h = hb + hfrow * 5 + e;
        if (e < 5 && h < h_pad) { a0 = hyp[(size_t)(c + 3 * hf) * h_pad + h]; a1 = hyp[(size_t)(c + 3 * (hf + 2)) * h_pad + h]; }
    }
    for (int idx = lane; idx < RM_HPW * 12; idx += 64) {
        const int h = idx / 12, e = idx - 12 * h;
        s_hyp[wave][h][e] = (hb + h < h_pad) ? hyp[(size_t)e * h_pad + hb + h] : __builtin_nanf("");
    }
    __syncthreads();
    // one band for the wave: the union of its hypotheses' bands (a skipped iteration has none; an unbounded one makes every tile exact)
    float lo = INFINITY, hi = -INFINITY; bool unbounded = false;
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        const int h = hb + hf * 5 + e;
        if (h < h_pad) {
            const float mid_h = hyp[(size_t)12 * h_pad + h], half_h = hyp[(size_t)13 * h_pad + h];
            if (half_h != 0.f) { lo = fminf(lo, mid_h - half_h); hi = fmaxf(hi, mid_h + half_h); unbounded |= half_h != half_h; }
        }
    }
    lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
    unbounded = __any(unbounded);
    float mid = tau, half = 0.f;
    if (lo <= hi) { mid = 0.5f * (lo + hi); half = 0.5f * (hi - lo) * (1.0f + 1e-5f) + mid * 1e-6f; }
    if (unbounded) half = __builtin_nanf("");
    const v2f nmid2 = {-mid, -mid};
    const float (*rt)[12] = &s_hyp[wave][hf * 5];

    int cnt[5] = {0, 0, 0, 0, 0};
    unsigned sgn[5] = {0u, 0u, 0u, 0u, 0u};     // signs of d2 - mid, one bit per tile (1 = below mid = inlier), harvested every 32 tiles
    unsigned n_rescored = 0;
    const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // subtract q, square, classify: 25 vector instructions for the lane's 5 hypotheses x 1 point
    auto classify = [&](const v16f& d, float b0, float b1, float qx, float qy, float qz) {
        const v2f q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
        const v2f dxab = (v2f){d[0], d[1]} - q2x, dyab = (v2f){d[2], d[3]} - q2y, dzab = (v2f){d[4], d[5]} - q2z;
        const v2f dxcd = (v2f){d[6], d[7]} - q2x, dycd = (v2f){d[8], d[9]} - q2y, dzcd = (v2f){d[10], d[11]} - q2z;
        const v2f dxye = (v2f){d[12], d[13]} - (v2f){qx, qy};
        const float dze = d[14] - qz;
        const v2f tab = fma2(dxab, dxab, fma2(dyab, dyab, fma2(dzab, dzab, nmid2)));
        const v2f tcd = fma2(dxcd, dxcd, fma2(dycd, dycd, fma2(dzcd, dzcd, nmid2)));
        const float tee = __builtin_fmaf(dxye.x, dxye.x, __builtin_fmaf(dxye.y, dxye.y, __builtin_fmaf(dze, dze, -mid)));
        float m = fminf(fminf(fabsf(tab.x), fabsf(tab.y)), fabsf(tee));
        m = fminf(fminf(m, fabsf(tcd.x)), fabsf(tcd.y));
        sgn[0] = __builtin_amdgcn_alignbit(sgn[0], __float_as_uint(tab.x), 31);
        sgn[1] = __builtin_amdgcn_alignbit(sgn[1], __float_as_uint(tab.y), 31);
        sgn[2] = __builtin_amdgcn_alignbit(sgn[2], __float_as_uint(tcd.x), 31);
        sgn[3] = __builtin_amdgcn_alignbit(sgn[3], __float_as_uint(tcd.y), 31);
        sgn[4] = __builtin_amdgcn_alignbit(sgn[4], __float_as_uint(tee), 31);
        if (__any(!(m >= half))) {     // a test of this tile lies inside the band: the reference arithmetic decides the tile
            ++n_rescored;
            const float px = __shfl(b0, col, 64), py = __shfl(b0, col + 32, 64), pz = __shfl(b1, col, 64);
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                const float* r = rt[e];
                const float x = (r[0] * px + (r[3] * py + r[6] * pz)) + r[9];
                const float y = (r[1] * px + (r[4] * py + r[7] * pz)) + r[10];
                const float z = (r[2] * px + (r[5] * py + r[8] * pz)) + r[11];
                const float dx = x - qx, dy = y - qy, dz = z - qz;
                const float d2 = dx * dx + (dy * dy + dz * dz);
                sgn[e] = (sgn[e] & ~1u) | ((d2 < tau) ? 1u : 0u);
            }
        }
    };
    auto transform = [&](float b0, float b1) {
        v16f d = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, zero16, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, d, 0, 0, 0);
    };
    auto load5 = [&](int g, v4f (&r)[5]) {
        const v4f* __restrict__ rec = reinterpret_cast<const v4f*>(pq3 + (size_t)g * RM_REC_FLOATS) + lane;
#pragma unroll
        for (int a = 0; a < 5; ++a) r[a] = rec[a * 64];
    };
    if (g0 < g1) {
        v4f cur[5], nxt[5];
        load5(g0, cur);
        int since = 0;
        for (int g = g0; g < g1; ++g) {
            load5(min(g + 1, g1 - 1), nxt);       // the next record's operands are in flight while this one is scored
            // the matrix pipe works on tile j + 1 while the vector pipe classifies tile j
            v16f dA = transform(cur[0][0], cur[1][0]);
            v16f dB = transform(cur[0][1], cur[1][1]);
            classify(dA, cur[0][0], cur[1][0], cur[2][0], cur[3][0], cur[4][0]);
            dA = transform(cur[0][2], cur[1][2]);
            classify(dB, cur[0][1], cur[1][1], cur[2][1], cur[3][1], cur[4][1]);
            dB = transform(cur[0][3], cur[1][3]);
            classify(dA, cur[0][2], cur[1][2], cur[2][2], cur[3][2], cur[4][2]);
            classify(dB, cur[0][3], cur[1][3], cur[2][3], cur[3][3], cur[4][3]);
            if (++since == 8) {
                since = 0;
#pragma unroll
                for (int e = 0; e < 5; ++e) { cnt[e] += __popc(sgn[e]); sgn[e] = 0u; }
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) cur[a] = nxt[a];
        }
#pragma unroll
        for (int e = 0; e < 5; ++e) cnt[e] += __popc(sgn[e]);
    }
    // a hypothesis' count: the sum over the 32 lanes (points) of its half
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        int c = cnt[e];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        if (col == 0 && hb + hf * 5 + e < h_pad) atomicAdd(&counts[hb + hf * 5 + e], c);
    }
    if (n_rescored && lane == 0) atomicAdd(rescored, (unsigned long long)n_rescored);
}

// error sum of one hypothesis (column-major R in T[0..8], t in T[9..11]) over all points
#endif  // TDV_STUDY

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
    // one device block: [0] bad index flag, [1] largest |source coordinate|, [2..3] rescored chunks, [4..5] scored chunks (u64 each),
    // [8] the best count known so far, [10..13] and [12+..] the two batch buffers' bail-out plans (4 ints each at [10] and [14]), [16..27] the winning
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
    const bool score_fast = !score_exact_env && ctx->ransac_score_mode != TDV_RANSAC_SCORE_EXACT;
    const bool score_mfma_env = study_env("TDV_RANSAC_SCORE") && !strcmp(study_env("TDV_RANSAC_SCORE"), "mfma");
    const bool score_mfma = kStudyBuild && score_fast && (score_mfma_env || ctx->ransac_score_mode == TDV_RANSAC_SCORE_MATRIX);   // study build only
    static const bool bailout_env_off = getenv("TDV_RANSAC_BAILOUT") && atoi(getenv("TDV_RANSAC_BAILOUT")) == 0;   // A/B knob
    // RansacPlan; short calls run as one batch without it (C4's 10,000 iterations: a short first batch was tried for them and lost
    // 2 % - their best fitness is 0.1-0.2, so at most a fifth of the points could be left out)
    const bool bailout = score_fast && !score_mfma && !trace_inliers && !bailout_env_off && max_iterations > 16384;
    int* d_state = d_bad + 8;                       // [0] best count known so far
    int* d_plan[2] = {d_bad + 10, d_bad + 28};   // per batch buffer: phase-1 chunks, survivors, largest prefix count, chunks per workgroup
    const int drop_permille = study_env("TDV_RANSAC_DROP_PERMILLE") ? atoi(study_env("TDV_RANSAC_DROP_PERMILLE")) : 100;   // tuning knob (5 to 100 measured equal)
    float* pq2 = nullptr;
#ifdef TDV_STUDY
    float* pq3 = nullptr;
    const int n_rec = (ns + 127) / 128;
    if (score_mfma) {
        TDV_TRY(ws_alloc(ctx, (size_t)n_rec * RM_REC_FLOATS, &pq3));
        k_pack_pq3<<<n_rec, 256, 0, s>>>(pq, ns_pad, n_rec, pq3);
    } else
#endif
    {
        TDV_TRY(ws_alloc(ctx, (size_t)ns_pad * 6, &pq2));
        k_pack_pq2<<<(ns_pad / 2 + 255) / 256, 256, 0, s>>>(pq, ns_pad, pq2);
    }
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
    int* d_list[2] = {nullptr, nullptr};
    for (int q = 0; q < 2; ++q) {
        TDV_TRY(ws_alloc(ctx, (size_t)14 * h_pad, &hyp[q]));
        TDV_TRY(ws_alloc(ctx, (size_t)h_pad, &counts[q]));
        TDV_TRY(ws_alloc(ctx, (size_t)batch, &d_tri[q]));
    }
    if (bailout) for (int q = 0; q < 2; ++q) TDV_TRY(ws_alloc(ctx, (size_t)h_pad, &d_list[q]));   // a batch's list lives until its phase 2 has run, behind the next batch's phase 1
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
    // with the bail-out a shorter first batch establishes a best count for the rest
    const int first_batch = bailout ? 8 * RS_HYP_PER_BLOCK : batch;
    auto prepare = [&](int q, int it0) -> int {    // host: draw + pack the triples of one batch
        const int cnt = std::min(it0 == 0 ? first_batch : batch, max_iterations - it0);
        uint64_t d[3];
        for (int k = 0; k < cnt; ++k) {
            stream_idx.next(d);
            int valid = !(d[0] == d[1] || d[1] == d[2] || d[0] == d[2]);  // registration.cpp:240
            h_tri[q][k] = make_int4((int)d[0], (int)d[1], (int)d[2], valid);
        }
        return cnt;
    };
    // point ranges of a scoring dispatch with hb hypothesis blocks (counts are accumulated by atomics, so the cut may differ per dispatch)
    auto score_grid = [](int hb, int ps) { return 8 * ((hb + 8 / RS_XCD_R - 1) / (8 / RS_XCD_R)) * ((ps + RS_XCD_R - 1) / RS_XCD_R); };   // k_ransac_score_fast's job-A workgroups
    auto ranges_for = [&](int hb) {
        const int ps = std::max(1, std::min(std::min((RS_WG_TARGET + hb - 1) / hb, std::max(1, n_pchunks / 32)), 512));
        const int per = (n_pchunks + ps - 1) / ps;
        return (n_pchunks + per - 1) / per;
    };
    int pending = -1, pending_cnt = 0;   // bail-out: the batch buffer whose phase 2 has not been enqueued yet
    // phase 2 of the pending batch (alone, or riding behind job A of the batch in `a`), then its final counts: best, copy, event
    auto finish_pending = [&](const ScoreJob* a, int g1) -> int {
        const int p = pending;
        const int hbp = (int)(align_up((size_t)pending_cnt, RS_HYP_PER_BLOCK) / RS_HYP_PER_BLOCK);
        ScoreJob jb{hyp[p], counts[p], d_plan[p], d_list[p], hbp, 0};
        // job B's grid: the host knows neither how many hypotheses survived nor how long phase 1 was; its workgroups stride over
        // the (range, block) items, so any multiple of 8 is enough - a quarter of a full grid covers the usual eighth of
        // survivors in one pass, surplus workgroups return at once
        const int g2 = std::max(8, (hbp * ranges_for(hbp) / 4 + 7) / 8 * 8);
        {
            ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
            if (a) k_ransac_score_fast<<<g1 + g2, RS_BLOCK, 0, s>>>(*a, jb, g1, h_pad, pq2, n_pchunks, tau, d_rescored);
            else k_ransac_score_fast<<<g2, RS_BLOCK, 0, s>>>(jb, jb, 0, h_pad, pq2, n_pchunks, tau, d_rescored);
        }
        k_ransac_best<<<(pending_cnt + 1023) / 1024, 1024, 0, s>>>(d_tri[p], pending_cnt, counts[p], d_state);
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipMemcpyAsync(h_cnt[p], counts[p], (size_t)pending_cnt * 4, hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipEventRecord(ev[p], s));
        pending = -1;
        return TDV_OK;
    };
    auto enqueue = [&](int q, int cnt) -> int {     // device: hypotheses + scoring + counts back to the host
        TDV_HIP(ctx, hipMemcpyAsync(d_tri[q], h_tri[q], (size_t)cnt * 16, hipMemcpyHostToDevice, s));
        k_ransac_hypotheses<<<(h_pad + 255) / 256, 256, 0, s>>>(pq, d_tri[q], cnt, h_pad, hyp[q], d_pmax, sqrt_tau, counts[q],
                                                                 (score_mfma ? 24.f : 16.f) * 5.9604644775390625e-08f);
        const int hb = (int)(align_up((size_t)cnt, RS_HYP_PER_BLOCK) / RS_HYP_PER_BLOCK);
        {   // TDV_TIMER_RANSAC_SCORE brackets every dispatch of a scoring kernel on its own
#ifdef TDV_STUDY
            if (score_mfma) {
                ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
                const int groups = (cnt + RM_HPW - 1) / RM_HPW, gblocks = (groups + RM_WAVES - 1) / RM_WAVES;
                int splits = std::max(1, std::min((16384 + groups - 1) / groups, std::max(1, n_rec / 16)));
                const int rec_per_split = (n_rec + splits - 1) / splits;
                splits = (n_rec + rec_per_split - 1) / rec_per_split;
                k_ransac_score_mfma<<<dim3(gblocks, splits), 64 * RM_WAVES, 0, s>>>(hyp[q], h_pad, pq3, n_rec, rec_per_split, tau, counts[q], d_rescored);
                wave_chunks += (double)gblocks * RM_WAVES * 4.0 * (double)n_rec;
            }
            else
#endif
            if (score_fast) {
                const int ps = ranges_for(hb);
                if (bailout) {
                    // One dispatch per batch: its phase 1 (job A) and, behind it, phase 2 of the batch before (job B).  The plan
                    // of this batch is made from the best count known now: full counts of the batches whose phase 2 has run,
                    // the prefix counts of the pending one (a lower bound of its full counts - a bound is all the rule needs).
                    k_ransac_plan<<<1, 1, 0, s>>>(d_state, d_plan[q], ns, n_pchunks, ps, drop_permille);
                    ScoreJob ja{hyp[q], counts[q], d_plan[q], nullptr, hb, ps};
                    const int g1 = score_grid(hb, ps);        // (a multiple of 8: job B's XCD numbering starts there)
                    if (pending >= 0) TDV_TRY(finish_pending(&ja, g1));
                    else {
                        ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
                        k_ransac_score_fast<<<g1, RS_BLOCK, 0, s>>>(ja, ja, g1, h_pad, pq2, n_pchunks, tau, d_rescored);
                    }
                    // survivors of this batch: the in-batch bound first (largest prefix count), then the list; the prefix counts
                    // also raise the best known for the batches after this one
                    const bool merge_on = study_env("TDV_RANSAC_MERGE") && atoi(study_env("TDV_RANSAC_MERGE")) == 1;     // (study build; read per call: the tests switch it)
                    k_ransac_best<<<(cnt + 1023) / 1024, 1024, 0, s>>>(d_tri[q], cnt, counts[q], d_plan[q] + 2);
                    k_ransac_select<<<(cnt + 255) / 256, 256, 0, s>>>(d_tri[q], cnt, counts[q], ns, confidence, d_state, d_plan[q], d_list[q]);
                    // (merged mode only: phase 2 comes a dispatch later, the prefix counts raise the bound for the batch in between;
                    //  otherwise the full counts do that right after phase 2)
                    if (merge_on) k_ransac_best<<<(cnt + 1023) / 1024, 1024, 0, s>>>(d_tri[q], cnt, counts[q], d_state);
                    pending = q; pending_cnt = cnt;
                    wave_chunks += (double)hb * (RS_BLOCK / 64) * (double)n_pchunks;
                    TDV_CHECK_LAUNCH(ctx);
                    // Phase 2 runs as a dispatch of its own right away.  Letting it ride behind the NEXT batch's phase 1 (TDV_RANSAC_MERGE=1:
                    // one scoring dispatch per batch, job B of k_ransac_score_fast) was built and measured: the dispatches gain a point
                    // of lane-op utilisation (0.667 vs 0.660 of the peak on the same box) but a batch's counts then reach the host one
                    // dispatch later, the host prepares the next index batch with nothing queued behind it, and the call loses 11 %
                    // end to end (19.0 vs 21.7 M hypotheses/s; profiles/r3/history/ransac_merged_dispatch.md).  Two batches in flight
                    // (three buffer sets) would hide that for one point of utilisation - not built.
                    if (!merge_on) return finish_pending(nullptr, 0);
                    return TDV_OK;                           // counts and event follow with this batch's phase 2 (finish_pending)
                } else {
                    ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
                    ScoreJob ja{hyp[q], counts[q], nullptr, nullptr, hb, ps};
                    const int gA = score_grid(hb, ps);
                    k_ransac_score_fast<<<gA, RS_BLOCK, 0, s>>>(ja, ja, gA, h_pad, pq2, n_pchunks, tau, d_rescored);
                }
                wave_chunks += (double)hb * (RS_BLOCK / 64) * (double)n_pchunks;
            }
            else {
                ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
                k_ransac_score<<<dim3(hb, psplit), RS_BLOCK, 0, s>>>(hyp[q], h_pad, pq2, n_pchunks, pchunks_per_split, tau, counts[q]);
            }
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
            status = enqueue(nxt, cnt_next);           // (with the bail-out this also runs phase 2 of `cur` and sends its counts)
            if (status != TDV_OK) break;
        } else if (pending == cur) {                  // last batch: its phase 2 runs alone
            status = finish_pending(nullptr, 0);
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
    ctx->last_ransac_rescore = -1.0; ctx->last_ransac_scored = 1.0;
    // statistics of the fast pass: (wave, chunk) pairs scored twice / scored (the FMA kernel counts the latter itself: the
    // bail-out leaves chunks out), and the scored share of all pairs
    auto stats = [&](const unsigned long long* r) {
        const double scored = r[1] ? (double)r[1] : wave_chunks;
        ctx->last_ransac_rescore = (double)r[0] / (scored * (score_mfma ? 1.0 : (double)(RS_PCH / 2)));      // the FMA kernel counts point pairs scored twice, the matrix-core study kernel chunks
        ctx->last_ransac_scored = scored / wave_chunks;
    };
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
        if (want_stats) stats(h_res);
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
        stats(h_res);
    }
    return TDV_OK;
}


// ---- many small clouds against one target: the whole coarse alignment in a handful of launches (round 3) ---------------------
// A batch of small instances (config C5: 1,024 clouds of ~400 voxels, 10,000 hypotheses each) is bound by what ransac_run_dev
// does on the HOST per call - 30,000 index draws, a dozen launches, two synchronisations - not by its kernels.  Here:
//   k_rb_sample      the reference's index stream on the device.  mt19937(42)'s raw outputs are the same for every cloud (the host
//                    draws them once per call); only libstdc++'s Lemire mapping to [0, n) depends on the cloud.  A workgroup per
//                    cloud maps the raw draws 1,024 at a time and compacts away the rejected ones (probability n / 2^32 each) with
//                    a workgroup scan, so a rejection shifts everything after it exactly as the sequential loop does.
//   k_rb_gather_pq   the (point, matched target) pairs of all clouds, each cloud padded to whole scoring chunks; k_pack_pq2 as usual
//   k_rb_hypotheses  one lane per (cloud, iteration): the lane function of k_ransac_hypotheses
//   k_rb_score       one workgroup per (cloud, 1,024 hypotheses): score_range_fast over all the cloud's chunks
//   k_rb_select      one workgroup per cloud: the loop of registration.cpp:281-290 as two reductions - the first iteration
//                    whose fitness passes the confidence bounds the prefix, then the first largest fitness inside it.
// Index stream, transforms, counts and the winner are ransac_run_dev's (tests/test_gpu_c5.py and test_gpu_chain.py hold the batch
// against the operator chain bit for bit).  The winner's rmse is not evaluated: the batch does not report it.
struct RbResult { float T[12]; int best_iter, inliers, iterations_run, pad; };

__global__ __launch_bounds__(1024)
void k_rb_sample(const unsigned* __restrict__ raw, int n_raw, const int* __restrict__ off, int H, int* __restrict__ idx_out /* [clouds][3 H] */, int* __restrict__ fail) {
    const int b = blockIdx.x, n = off[b + 1] - off[b];
    if (n <= 0) return;
    const unsigned range = (unsigned)n, thr = (0u - range) % range;        // libstdc++ 11 uniform_int_distribution (Lemire), see ctx.hip
    int* out = idx_out + (size_t)b * 3 * H;
    __shared__ int s_wave[16];
    __shared__ int s_produced;
    if (threadIdx.x == 0) s_produced = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r0 = 0; r0 < n_raw; r0 += 1024) {
        const int produced = s_produced;
        if (produced >= 3 * H) break;                                      // workgroup-uniform
        const int r = r0 + threadIdx.x;
        unsigned long long prod = 0ull; bool acc = false;
        if (r < n_raw) { prod = (unsigned long long)raw[r] * (unsigned long long)range; acc = !((unsigned)prod < thr); }
        const unsigned long long m = __ballot(acc);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
        const int q = produced + before + __popcll(m & ((1ull << lane) - 1ull));
        if (acc && q < 3 * H) out[q] = (int)(prod >> 32);
        __syncthreads();
        if (threadIdx.x == 0) s_produced = produced + total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_produced < 3 * H) *fail = 1;                  // ran out of raw draws (cannot happen with the slack the host adds)
}

// pos_off[b]: first padded pair slot of cloud b (multiples of 2 RS_PCH); grid over all padded slots
__global__ void k_rb_gather_pq(const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ corr, const int* __restrict__ off,
                               const int* __restrict__ pos_off, int n_clouds, int total_pos, int nt, float* __restrict__ pq, int* __restrict__ bad,
                               unsigned* __restrict__ pmax /* [clouds] */) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= total_pos) return;
    int a = 0, z = n_clouds;
    while (z - a > 1) { const int m = (a + z) >> 1; if (pos_off[m] <= pos) a = m; else z = m; }
    const int i = pos - pos_off[a], n = off[a + 1] - off[a];
    float4 A, Bq;
    if (i < n) {
        const size_t P = (size_t)off[a] + i;
        int c = corr[P];
        if ((unsigned)c >= (unsigned)nt) { *bad = 1; c = 0; }
        A = make_float4(src[3 * P], src[3 * P + 1], src[3 * P + 2], tgt[3 * c]);
        Bq = make_float4(tgt[3 * c + 1], tgt[3 * c + 2], 0.f, 0.f);
        float am = fmaxf(fabsf(A.x), fmaxf(fabsf(A.y), fabsf(A.z)));
        if (!(am <= FLT_MAX)) am = INFINITY;
        if (A.w != A.w || Bq.x != Bq.x || Bq.y != Bq.y) am = INFINITY;      // as k_gather_pq
        if (am > 0.f) atomicMax(&pmax[a], __float_as_uint(am));
    } else {
        A = make_float4(0.f, 0.f, 0.f, INFINITY);
        Bq = make_float4(INFINITY, INFINITY, 0.f, 0.f);
    }
    reinterpret_cast<float4*>(pq)[2 * (size_t)pos] = A;
    reinterpret_cast<float4*>(pq)[2 * (size_t)pos + 1] = Bq;
}

__global__ void k_rb_hypotheses(const float* __restrict__ pq, const int* __restrict__ pos_off, const int* __restrict__ off, const int* __restrict__ idx /* [clouds][3 H] */,
                                int H, int h_pad, float* __restrict__ hyp /* [clouds][14][h_pad] */, const unsigned* __restrict__ pmax, float sqrt_tau, float band_u) {
    const int b = blockIdx.y, h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= h_pad || off[b + 1] == off[b]) return;
    bool valid = false;
    int4 tr = make_int4(0, 0, 0, 0);
    if (h < H) {
        const int* t = idx + (size_t)b * 3 * H + 3 * (size_t)h;
        tr = make_int4(t[0], t[1], t[2], 0);
        valid = !(tr.x == tr.y || tr.y == tr.z || tr.x == tr.z);           // registration.cpp:240
    }
    ransac_hypothesis_lane(pq + (size_t)pos_off[b] * 8, tr, valid, h, h_pad, hyp + (size_t)b * 14 * h_pad, pmax + b, sqrt_tau, band_u);
}

__global__ __launch_bounds__(RS_BLOCK)
void k_rb_score(const float* __restrict__ hyp, int h_pad, const float* __restrict__ pq2, const int* __restrict__ pos_off, const int* __restrict__ off, float tau,
                int* __restrict__ counts /* [clouds][h_pad] */, unsigned long long* __restrict__ rescored /* [0] chunks scored twice, [1] chunks scored (per wave) */) {
    // consecutive workgroup ids = consecutive clouds of ONE hypothesis block: a cloud's 10 blocks land on the same XCD (id mod 8 =
    // cloud mod 8), so its pair records are pulled into one L2 only, and stay there for the next block
    const int b = blockIdx.x;
    if (off[b + 1] == off[b]) return;
    const int base = blockIdx.y * RS_BLOCK + threadIdx.x;
    const int chunks = (pos_off[b + 1] - pos_off[b]) / RS_PCH;
    unsigned n_rescored = 0;
    const int cnt = score_range_fast<true>(hyp + (size_t)b * 14 * h_pad, h_pad, base, pq2 + (size_t)pos_off[b] * 6, 0, chunks, tau, n_rescored);
    counts[(size_t)b * h_pad + base] = cnt;
    // statistics only (tdv_ctx_last_ransac_rescore): two atomics per workgroup
    __shared__ unsigned s_rescored;
    if (threadIdx.x == 0) s_rescored = 0u;
    __syncthreads();
    if (n_rescored && (threadIdx.x & 63) == 0) atomicAdd(&s_rescored, n_rescored);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_rescored) atomicAdd(rescored, (unsigned long long)s_rescored);
        atomicAdd(rescored + 1, (unsigned long long)(RS_BLOCK / 64) * (unsigned long long)chunks);
    }
}

__global__ __launch_bounds__(256)
void k_rb_select(const int* __restrict__ idx, const int* __restrict__ counts, const float* __restrict__ hyp, const int* __restrict__ off, int H, int h_pad,
                 float confidence, RbResult* __restrict__ res) {
    const int b = blockIdx.x, n = off[b + 1] - off[b];
    RbResult* r = res + b;
    if (n <= 0) { if (threadIdx.x == 0) { r->best_iter = -1; r->inliers = 0; r->iterations_run = 0; } return; }
    const int* t = idx + (size_t)b * 3 * H;
    const int* c = counts + (size_t)b * h_pad;
    const float fn = static_cast<float>((size_t)n);
    __shared__ int s_stop;
    __shared__ unsigned long long s_best[4];
    if (threadIdx.x == 0) s_stop = H;
    __syncthreads();
    // the first iteration whose fitness passes the confidence ends the loop (registration.cpp:290)
    int stop = H;
    for (int h = threadIdx.x; h < H; h += 256) {
        const bool valid = !(t[3 * h] == t[3 * h + 1] || t[3 * h + 1] == t[3 * h + 2] || t[3 * h] == t[3 * h + 2]);
        if (valid && static_cast<float>(c[h]) / fn > confidence) { stop = h; break; }
    }
    atomicMin(&s_stop, stop);
    __syncthreads();
    const int k_end = s_stop < H ? s_stop + 1 : H;
    // the first largest fitness among iterations [0, k_end): key = fitness bits (positive floats order as their bits), then
    // the EARLIEST iteration (largest H - h)
    unsigned long long best = 0ull;
    for (int h = threadIdx.x; h < k_end; h += 256) {
        const bool valid = !(t[3 * h] == t[3 * h + 1] || t[3 * h + 1] == t[3 * h + 2] || t[3 * h] == t[3 * h + 2]);
        if (!valid) continue;
        const float fit = static_cast<float>(c[h]) / fn;                     // registration.cpp:281
        if (!(fit > 0.f)) continue;                                          // has to beat the initial best fitness 0 (:284)
        const unsigned long long key = ((unsigned long long)__float_as_uint(fit) << 32) | (unsigned)(H - h);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = shfl_u64_down(best, o); best = x > best ? x : best; }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = s_best[w] > best ? s_best[w] : best;
        r->iterations_run = k_end;
        if (best == 0ull) { r->best_iter = -1; r->inliers = 0; }
        else {
            const int h = H - (int)(unsigned)(best & 0xffffffffull);
            r->best_iter = h; r->inliers = c[h];
            const float* hp = hyp + (size_t)b * 14 * h_pad;
            for (int e = 0; e < 12; ++e) r->T[e] = hp[(size_t)e * h_pad + h];
        }
    }
}

int ransac_small_batch_dev(tdv_ctx* ctx, const float* d_src, const int* h_off, const int* d_off, int n_clouds, const float* d_tgt, int nt, const int* d_corr,
                           float voxel, int max_iterations, float confidence, uint32_t seed, tdv_ransac_result* out, int* fell_back) {
    if (!ctx || !h_off || !d_off || !out || !fell_back || n_clouds < 0 || nt < 0 || max_iterations < 0) return TDV_ERR_BAD_ARG;
    *fell_back = 0;
    for (int b = 0; b < n_clouds; ++b) {      // RegistrationResult defaults (include/registration.hpp:26-30)
        for (int i = 0; i < 16; ++i) out[b].T[i] = (i % 5 == 0) ? 1.f : 0.f;
        out[b].fitness = 0.f; out[b].rmse = 0.f; out[b].inliers = 0; out[b].best_iteration = -1; out[b].iterations_run = 0;
    }
    const int total = n_clouds ? h_off[n_clouds] : 0;
    if (total == 0 || nt == 0 || max_iterations == 0) return TDV_OK;
    if (!d_src || !d_tgt || !d_corr) return TDV_ERR_BAD_ARG;
    const bool off_env = getenv("TDV_RANSAC_BATCH") && atoi(getenv("TDV_RANSAC_BATCH")) == 0;   // A/B knob (read per call: the tests switch it)
    const bool fast_mode = ctx->ransac_score_mode == TDV_RANSAC_SCORE_FAST && !(getenv("TDV_RANSAC_SCORE") && strcmp(getenv("TDV_RANSAC_SCORE"), "fast"));
    int v_max = 0;
    for (int b = 0; b < n_clouds; ++b) v_max = std::max(v_max, h_off[b + 1] - h_off[b]);
    if (off_env || !fast_mode || v_max > 4096 || max_iterations > 32768) { *fell_back = 1; return TDV_OK; }
    hipStream_t s = ctx->stream;
    const float thr = voxel * 1.5f;  // registration.cpp:213
    const float tau = tau_lt(thr);
    const float sqrt_tau = std::nextafter((float)std::sqrt((double)tau), INFINITY);
    const int H = max_iterations, h_pad = (int)align_up((size_t)H, RS_HYP_PER_BLOCK), hb = h_pad / RS_HYP_PER_BLOCK;
    // padded pair slots per cloud
    std::vector<int> pos_off((size_t)n_clouds + 1, 0);
    for (int b = 0; b < n_clouds; ++b) pos_off[b + 1] = pos_off[b] + (int)align_up((size_t)(h_off[b + 1] - h_off[b]), (size_t)RS_PCH * 2);   // whole chunks of RS_PCH pairs, 64-B aligned records
    const int total_pos = pos_off[n_clouds];
    const int n_raw = 3 * H + 4096;                                            // slack for rejected draws (each has probability n / 2^32)
    int* d_pos_off; unsigned* d_raw; int* d_idx; float *pq, *pq2, *hyp; int* counts; unsigned* d_pmax; int* d_flags; RbResult* d_res;
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds + 1, &d_pos_off));
    TDV_TRY(ws_alloc(ctx, (size_t)n_raw, &d_raw));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds * 3 * H, &d_idx));
    TDV_TRY(ws_alloc(ctx, (size_t)total_pos * 8, &pq));
    TDV_TRY(ws_alloc(ctx, (size_t)total_pos * 6, &pq2));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds * 14 * h_pad, &hyp));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds * h_pad, &counts));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds + 2, &d_pmax));                     // pmax[clouds] | bad | fail
    d_flags = reinterpret_cast<int*>(d_pmax + n_clouds);
    unsigned long long* d_stats;                                               // rescored, scored (their own allocation: 8-byte aligned whatever n_clouds is)
    TDV_TRY(ws_alloc(ctx, 2, &d_stats));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds, &d_res));
    const size_t pin_raw = align_up((size_t)n_raw * 4, 64), pin_res = align_up((size_t)n_clouds * sizeof(RbResult), 64);
    TDV_TRY(pin_reserve(ctx, pin_raw + pin_res + 64 + ((size_t)n_clouds + 1) * 4));
    unsigned* h_raw = reinterpret_cast<unsigned*>(ctx->pin);
    RbResult* h_res = reinterpret_cast<RbResult*>(ctx->pin + pin_raw);
    int* h_flags = reinterpret_cast<int*>(ctx->pin + pin_raw + pin_res);
    int* h_pos = h_flags + 16;
    mt19937_raw(seed, (size_t)n_raw, h_raw);
    std::memcpy(h_pos, pos_off.data(), ((size_t)n_clouds + 1) * 4);
    TDV_HIP(ctx, hipMemcpyAsync(d_raw, h_raw, (size_t)n_raw * 4, hipMemcpyHostToDevice, s));
    TDV_HIP(ctx, hipMemcpyAsync(d_pos_off, h_pos, ((size_t)n_clouds + 1) * 4, hipMemcpyHostToDevice, s));
    TDV_HIP(ctx, hipMemsetAsync(d_pmax, 0, ((size_t)n_clouds + 2) * 4, s));
    TDV_HIP(ctx, hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned long long), s));
    k_rb_sample<<<n_clouds, 1024, 0, s>>>(d_raw, n_raw, d_off, H, d_idx, d_flags + 1);
    k_rb_gather_pq<<<(total_pos + 255) / 256, 256, 0, s>>>(d_src, d_tgt, d_corr, d_off, d_pos_off, n_clouds, total_pos, nt, pq, d_flags, d_pmax);
    k_pack_pq2<<<(total_pos / 2 + 255) / 256, 256, 0, s>>>(pq, total_pos, pq2);
    k_rb_hypotheses<<<dim3((h_pad + 255) / 256, n_clouds), 256, 0, s>>>(pq, d_pos_off, d_off, d_idx, H, h_pad, hyp, d_pmax, sqrt_tau, 16.f * 5.9604644775390625e-08f);
    {
        ScopedTimer tm(ctx, TDV_TIMER_RANSAC_SCORE);
        k_rb_score<<<dim3(n_clouds, hb), RS_BLOCK, 0, s>>>(hyp, h_pad, pq2, d_pos_off, d_off, tau, counts, d_stats);
    }
    k_rb_select<<<n_clouds, 256, 0, s>>>(d_idx, counts, hyp, d_off, H, h_pad, confidence, d_res);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipMemcpyAsync(h_res, d_res, (size_t)n_clouds * sizeof(RbResult), hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipMemcpyAsync(h_flags, d_flags, 8, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipMemcpyAsync(h_flags + 4, d_stats, 16, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    {
        const unsigned long long* st = reinterpret_cast<const unsigned long long*>(h_flags + 4);
        ctx->last_ransac_rescore = st[1] ? (double)st[0] / ((double)st[1] * (RS_PCH / 2)) : 0.0; ctx->last_ransac_scored = 1.0;      // (pairs scored twice / pairs scored)
    }
    if (h_flags[0]) { std::snprintf(ctx->err, sizeof(ctx->err), "ransac: a correspondence index lies outside [0, %d)", nt); return TDV_ERR_BAD_ARG; }
    if (h_flags[1]) { *fell_back = 1; return TDV_OK; }
    for (int b = 0; b < n_clouds; ++b) {
        const int n = h_off[b + 1] - h_off[b];
        if (n == 0) continue;
        const RbResult& r = h_res[b];
        out[b].iterations_run = r.iterations_run;
        if (r.best_iter < 0) continue;
        for (int c = 0; c < 3; ++c) for (int q = 0; q < 3; ++q) out[b].T[c * 4 + q] = r.T[c * 3 + q];
        out[b].T[12] = r.T[9]; out[b].T[13] = r.T[10]; out[b].T[14] = r.T[11];
        out[b].inliers = r.inliers; out[b].best_iteration = r.best_iter;
        out[b].fitness = static_cast<float>(r.inliers) / static_cast<float>((size_t)n);
    }
    return TDV_OK;
}

}  // namespace tdv
