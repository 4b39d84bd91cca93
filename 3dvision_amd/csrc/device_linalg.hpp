// Per-lane small dense solvers for gfx950 (one lane = one problem; no MFMA: there is no dense
// contraction here).  They implement the solver semantics the reference gets from Eigen
// (not vendored; call sites: /root/reference/src/registration.cpp:122-123 SelfAdjointEigenSolver,
// :255-262 / :388-394 JacobiSVD + Kabsch, :366 LDLT solve, :369-371 AngleAxis product):
//   * two-sided Jacobi SVD of a 3x3 with the (1,0),(2,0),(2,1) sweep order, descending
//     singular values, V U^T with the det<0 column flip;
//   * symmetric 3x3 eigen-decomposition by Householder tridiagonalisation + implicit
//     Wilkinson-shift QR, ascending eigenvalues;
//   * 6x6 diagonally pivoted LDL^T with pseudo-inverse of D (zero pivot -> zero component);
//   * X*Y*Z Euler rotation through the quaternion product.
// Every 3-term sum is evaluated as c0 + (c1 + c2) and nothing is contracted into FMA
// (the TU is compiled with -ffp-contract=off), so results are reproducible bit for bit and
// match a CPU evaluation of the same expression tree; sqrt and division are the correctly
// rounded forms (-fhip-fp32-correctly-rounded-divide-sqrt).
// All matrices here are column-major: a[c*3 + r].
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include "libm_f32.hpp"

namespace tdv {
namespace dl {

#define TDV_DI __device__ __forceinline__

TDV_DI float s3(float a, float b, float c) { return a + (b + c); }

struct Mat3 { float a[9]; };
TDV_DI float& el(Mat3& m, int r, int c) { return m.a[c * 3 + r]; }
TDV_DI float el(const Mat3& m, int r, int c) { return m.a[c * 3 + r]; }

TDV_DI Mat3 ident3() { Mat3 m; for (int i = 0; i < 9; ++i) m.a[i] = 0.f; m.a[0] = m.a[4] = m.a[8] = 1.f; return m; }
TDV_DI Mat3 mul3(const Mat3& x, const Mat3& y) {
    Mat3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            el(r, i, j) = s3(el(x, i, 0) * el(y, 0, j), el(x, i, 1) * el(y, 1, j), el(x, i, 2) * el(y, 2, j));
    return r;
}
TDV_DI Mat3 transp3(const Mat3& x) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) el(r, i, j) = el(x, j, i);
    return r;
}
TDV_DI void mulv3(const Mat3& m, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
    ox = s3(el(m, 0, 0) * vx, el(m, 0, 1) * vy, el(m, 0, 2) * vz);
    oy = s3(el(m, 1, 0) * vx, el(m, 1, 1) * vy, el(m, 1, 2) * vz);
    oz = s3(el(m, 2, 0) * vx, el(m, 2, 1) * vy, el(m, 2, 2) * vz);
}
TDV_DI float det3(const Mat3& m) {
    float h0 = el(m, 0, 0) * (el(m, 1, 1) * el(m, 2, 2) - el(m, 1, 2) * el(m, 2, 1));
    float h1 = el(m, 0, 1) * (el(m, 1, 0) * el(m, 2, 2) - el(m, 1, 2) * el(m, 2, 0));
    float h2 = el(m, 0, 2) * (el(m, 1, 0) * el(m, 2, 1) - el(m, 1, 1) * el(m, 2, 0));
    return h0 - h1 + h2;
}

struct Rot2 { float c, s; };
TDV_DI Rot2 rtrans(Rot2 r) { return Rot2{r.c, -r.s}; }

// Rotation that diagonalises the symmetric 2x2 [[x,y],[y,z]].
TDV_DI Rot2 jacobi_sym(float x, float y, float z) {
    Rot2 r;
    float deno = 2.f * fabsf(y);
    if (deno < FLT_MIN) { r.c = 1.f; r.s = 0.f; return r; }
    float tau = (x - z) / deno;
    float w = sqrtf(tau * tau + 1.f);
    float t = (tau > 0.f) ? 1.f / (tau + w) : 1.f / (tau - w);
    float sign_t = t > 0.f ? 1.f : -1.f;
    float n = 1.f / sqrtf(t * t + 1.f);
    r.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    r.c = n;
    return r;
}
TDV_DI Rot2 givens(float p, float q) {
    Rot2 r;
    if (q == 0.f) { r.c = p < 0.f ? -1.f : 1.f; r.s = 0.f; }
    else if (p == 0.f) { r.c = 0.f; r.s = q < 0.f ? 1.f : -1.f; }
    else if (fabsf(p) > fabsf(q)) {
        float t = q / p; float u = sqrtf(1.f + t * t); if (p < 0.f) u = -u;
        r.c = 1.f / u; r.s = -t * r.c;
    } else {
        float t = p / q; float u = sqrtf(1.f + t * t); if (q < 0.f) u = -u;
        r.s = -1.f / u; r.c = -t * r.s;
    }
    return r;
}
// rows P,Q of m:  x' = c x + s y ; y' = -s x + c y
template <int P, int Q> TDV_DI void rot_rows(Mat3& m, Rot2 j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float xi = el(m, P, i), yi = el(m, Q, i);
        el(m, P, i) = j.c * xi + j.s * yi;
        el(m, Q, i) = -j.s * xi + j.c * yi;
    }
}
// columns P,Q of m rotated by j^T (i.e. "apply j on the right")
template <int P, int Q> TDV_DI void rot_cols(Mat3& m, Rot2 j) {
    Rot2 jt = rtrans(j);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float xi = el(m, i, P), yi = el(m, i, Q);
        el(m, i, P) = jt.c * xi + jt.s * yi;
        el(m, i, Q) = -jt.s * xi + jt.c * yi;
    }
}

template <int P, int Q>
TDV_DI bool svd_sweep_pair(Mat3& W, Mat3& U, Mat3& V, float& maxDiag) {
    const float precision = 2.f * FLT_EPSILON;
    float threshold = fmaxf(FLT_MIN, precision * maxDiag);
    if (!(fabsf(el(W, P, Q)) > threshold || fabsf(el(W, Q, P)) > threshold)) return false;
    float m00 = el(W, P, P), m01 = el(W, P, Q), m10 = el(W, Q, P), m11 = el(W, Q, Q);
    Rot2 rot1;
    float t = m00 + m11;
    float d = m10 - m01;
    if (fabsf(d) < FLT_MIN) { rot1.s = 0.f; rot1.c = 1.f; }
    else {
        float u = t / d;
        float tmp = sqrtf(1.f + u * u);
        rot1.s = 1.f / tmp;
        rot1.c = u / tmp;
    }
    float a00 = rot1.c * m00 + rot1.s * m10, a01 = rot1.c * m01 + rot1.s * m11;
    float a11 = -rot1.s * m01 + rot1.c * m11;
    Rot2 jr = jacobi_sym(a00, a01, a11);
    Rot2 jrt = rtrans(jr);
    Rot2 jl{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
    rot_rows<P, Q>(W, jl);
    rot_cols<P, Q>(U, rtrans(jl));
    rot_cols<P, Q>(W, jr);
    rot_cols<P, Q>(V, jr);
    maxDiag = fmaxf(maxDiag, fmaxf(fabsf(el(W, P, P)), fabsf(el(W, Q, Q))));
    return true;
}

template <int A, int B> TDV_DI void swap_cols(Mat3& m) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { float t = el(m, r, A); el(m, r, A) = el(m, r, B); el(m, r, B) = t; }
}

// Full SVD A = U diag(s) V^T, s descending.
TDV_DI void svd3(const Mat3& A, Mat3& U, Mat3& V, float& s0, float& s1, float& s2) {
    float scale = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(A.a[i]));
    if (scale == 0.f) scale = 1.f;
    Mat3 W;
#pragma unroll
    for (int i = 0; i < 9; ++i) W.a[i] = A.a[i] / scale;
    U = ident3(); V = ident3();
    float maxDiag = fmaxf(fabsf(el(W, 0, 0)), fmaxf(fabsf(el(W, 1, 1)), fabsf(el(W, 2, 2))));
    bool finished = false;
    int guard = 0;  // the sweep converges in a handful of passes; bound it so no lane can spin
    while (!finished && guard < 64) {
        ++guard;
        bool any = false;
        any |= svd_sweep_pair<1, 0>(W, U, V, maxDiag);
        any |= svd_sweep_pair<2, 0>(W, U, V, maxDiag);
        any |= svd_sweep_pair<2, 1>(W, U, V, maxDiag);
        finished = !any;
    }
    float d0 = el(W, 0, 0), d1 = el(W, 1, 1), d2 = el(W, 2, 2);
    s0 = fabsf(d0); s1 = fabsf(d1); s2 = fabsf(d2);
    if (d0 < 0.f) { el(U, 0, 0) = -el(U, 0, 0); el(U, 1, 0) = -el(U, 1, 0); el(U, 2, 0) = -el(U, 2, 0); }
    if (d1 < 0.f) { el(U, 0, 1) = -el(U, 0, 1); el(U, 1, 1) = -el(U, 1, 1); el(U, 2, 1) = -el(U, 2, 1); }
    if (d2 < 0.f) { el(U, 0, 2) = -el(U, 0, 2); el(U, 1, 2) = -el(U, 1, 2); el(U, 2, 2) = -el(U, 2, 2); }
    s0 *= scale; s1 *= scale; s2 *= scale;
    // selection sort, first maximum wins
    {
        int pos = 0; float best = s0;
        if (s1 > best) { best = s1; pos = 1; }
        if (s2 > best) { best = s2; pos = 2; }
        if (pos == 1) { float t = s0; s0 = s1; s1 = t; swap_cols<0, 1>(U); swap_cols<0, 1>(V); }
        else if (pos == 2) { float t = s0; s0 = s2; s2 = t; swap_cols<0, 2>(U); swap_cols<0, 2>(V); }
    }
    if (s2 > s1) { float t = s1; s1 = s2; s2 = t; swap_cols<1, 2>(U); swap_cols<1, 2>(V); }
}

// R = V U^T with the reflection fix (negate V.col(2) when det < 0).
TDV_DI Mat3 kabsch_rotation(const Mat3& H) {
    Mat3 U, V; float s0, s1, s2;
    svd3(H, U, V, s0, s1, s2);
    Mat3 Ut = transp3(U);
    Mat3 R = mul3(V, Ut);
    if (det3(R) < 0.f) {
        el(V, 0, 2) = -el(V, 0, 2); el(V, 1, 2) = -el(V, 1, 2); el(V, 2, 2) = -el(V, 2, 2);
        R = mul3(V, Ut);
    }
    return R;
}

TDV_DI float hypot_pos(float x, float y) {
    x = fabsf(x); y = fabsf(y);
    float p = fmaxf(x, y);
    if (p == 0.f) return 0.f;
    float qp = fminf(y, x) / p;
    return p * sqrtf(1.f + qp * qp);
}

// columns K,K+1 of Q rotated ("Q = Q * G")
TDV_DI void rot_cols_dyn(Mat3& Q, int k, Rot2 j) {
    if (k == 0) rot_cols<0, 1>(Q, j); else rot_cols<1, 2>(Q, j);
}

// Eigenvector of the smallest eigenvalue of the symmetric 3x3 given by its lower triangle.
// Returns false if the QR iteration did not converge (vector then comes from the unsorted basis,
// as the reference's solver would leave it).
TDV_DI bool smallest_eigvec3(float a00, float a10, float a20, float a11, float a21, float a22,
                             float& vx, float& vy, float& vz) {
    float scale = fmaxf(fmaxf(fabsf(a00), fabsf(a10)), fmaxf(fmaxf(fabsf(a20), fabsf(a11)), fmaxf(fabsf(a21), fabsf(a22))));
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a20 /= scale; a11 /= scale; a21 /= scale; a22 /= scale;
    float d0, d1, d2, e0, e1;
    Mat3 Q = ident3();
    d0 = a00;
    float v1norm2 = a20 * a20;
    if (v1norm2 <= FLT_MIN) {
        d1 = a11; d2 = a22; e0 = a10; e1 = a21;
    } else {
        float beta = sqrtf(a10 * a10 + v1norm2);
        float invBeta = 1.f / beta;
        float m01 = a10 * invBeta;
        float m02 = a20 * invBeta;
        float q = 2.f * m01 * a21 + m02 * (a22 - a11);
        d1 = a11 + m02 * q;
        d2 = a22 - m02 * q;
        e0 = beta;
        e1 = a21 - m01 * q;
        el(Q, 1, 1) = m01; el(Q, 1, 2) = m02; el(Q, 2, 1) = m02; el(Q, 2, 2) = -m01;
    }
    float diag[3] = {d0, d1, d2};
    float sub[2] = {e0, e1};
    const int n = 3, maxIterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const float precision_inv = 1.f / FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            if (fabsf(sub[i]) < FLT_MIN) sub[i] = 0.f;
            else {
                const float scaled = precision_inv * sub[i];
                if (scaled * scaled <= (fabsf(diag[i]) + fabsf(diag[i + 1]))) sub[i] = 0.f;
            }
        }
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        float td = (diag[end - 1] - diag[end]) * 0.5f;
        float e = sub[end - 1];
        float mu = diag[end];
        if (td == 0.f) mu -= fabsf(e);
        else if (e != 0.f) {
            const float e2 = e * e;
            const float h = hypot_pos(td, e);
            if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? h : -h)) / e);
            else           mu -= e2 / (td + (td > 0.f ? h : -h));
        }
        float x = diag[start] - mu;
        float z = sub[start];
        for (int k = start; k < end && z != 0.f; ++k) {
            Rot2 rot = givens(x, z);
            float sdk = rot.s * diag[k] + rot.c * sub[k];
            float dkp1 = rot.s * sub[k] + rot.c * diag[k + 1];
            diag[k] = rot.c * (rot.c * diag[k] - rot.s * sub[k]) - rot.s * (rot.c * sub[k] - rot.s * diag[k + 1]);
            diag[k + 1] = rot.s * sdk + rot.c * dkp1;
            sub[k] = rot.c * sdk - rot.s * dkp1;
            if (k > start) sub[k - 1] = rot.c * sub[k - 1] - rot.s * z;
            x = sub[k];
            if (k < end - 1) { z = -rot.s * sub[k + 1]; sub[k + 1] = rot.c * sub[k + 1]; }
            rot_cols_dyn(Q, k, rot);
        }
    }
    bool ok = iter <= maxIterations * n;
    int col = 0;
    if (ok) {  // index of the first minimum = column that the ascending sort brings to position 0
        float mn = diag[0];
        if (diag[1] < mn) { mn = diag[1]; col = 1; }
        if (diag[2] < mn) { col = 2; }
    }
    vx = col == 0 ? el(Q, 0, 0) : (col == 1 ? el(Q, 0, 1) : el(Q, 0, 2));
    vy = col == 0 ? el(Q, 1, 0) : (col == 1 ? el(Q, 1, 1) : el(Q, 1, 2));
    vz = col == 0 ? el(Q, 2, 0) : (col == 1 ? el(Q, 2, 1) : el(Q, 2, 2));
    return ok;
}

// Solve A x = b for symmetric 6x6 A (row-major, lower triangle read) by pivoted LDL^T (the sequence of operations of
// Eigen::LDLT<Matrix<float,6,6>>::compute + solve; the CPU restatement used by the tests follows the same sequence).
//
// The only run-time index of the algorithm is the pivot `big` of step k.  With the lower triangle in 21 REGISTERS and the
// symmetric interchange of k and big written as predicated moves over the (static) triangle positions, every other access has
// compile-time indices: no scratch memory, no LDS.  (As private arrays indexed at run time the factor lived in scratch memory:
// ~500 dependent round trips per solve, 25-40 us per ICP iteration in round 2's one-lane tail; in LDS ~10 us.)  The values each
// operation sees, and the order of the operations, are unchanged - tests/test_gpu_icp.py and test_gpu_demo_chain.py hold the
// device transform against the CPU restatement bit for bit, the rank-deficient planar case (zero pivots) included.
__device__ __forceinline__ void ldlt6_solve(const float* Ain, const float* b, float* x) {
    constexpr int N = 6;
    float L[N][N];                                      // only j <= i is used
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = Ain[i * 6 + j];
    int transp[N];
    bool stop = false;                                  // k == 0 with a zero pivot: Eigen leaves the matrix alone, identity permutation
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (stop) break;
        int big = k; float bv = fabsf(L[k][k]);
#pragma unroll
        for (int i = k + 1; i < N; ++i) { const float a = fabsf(L[i][i]); if (a > bv) { bv = a; big = i; } }
        transp[k] = big;
        if (big != k) {
            // the four partial interchanges of Eigen's in-place LDLT on the lower triangle, each written over the STATIC rows /
            // columns that `big` can be: a predicated swap per candidate position
            auto cswap = [](bool c, float& a, float& b2) { const float t = a; a = c ? b2 : a; b2 = c ? t : b2; };
#pragma unroll
            for (int r = k + 1; r < N; ++r) {
                const bool is_big = r == big;
#pragma unroll
                for (int j = 0; j < k; ++j) cswap(is_big, L[k][j], L[r][j]);              // row k <-> row big, columns before k
                cswap(is_big, L[k][k], L[r][r]);                                           // the two diagonal entries
#pragma unroll
                for (int i = k + 1; i < r; ++i) cswap(is_big, L[i][k], L[r][i]);          // column k below k <-> row big, between k and big
            }
#pragma unroll
            for (int r = k + 2; r < N; ++r)
#pragma unroll
                for (int c = k + 1; c < r; ++c) cswap(c == big, L[r][k], L[r][c]);        // column k <-> column big, rows below big
        }
        if (k > 0) {
            float temp[N];
#pragma unroll
            for (int j = 0; j < k; ++j) temp[j] = L[j][j] * L[k][j];
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < k; ++j) acc += L[k][j] * temp[j];
            L[k][k] -= acc;
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                float a2 = 0.f;
#pragma unroll
                for (int j = 0; j < k; ++j) a2 += L[i][j] * temp[j];
                L[i][k] -= a2;
            }
        }
        const float akk = L[k][k];
        const bool valid = fabsf(akk) > 0.f;
        if (k == 0 && !valid) {
#pragma unroll
            for (int j = 0; j < N; ++j) transp[j] = j;
            stop = true;
        } else if (valid) {
#pragma unroll
            for (int i = k + 1; i < N; ++i) L[i][k] /= akk;
        }
    }
    float y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = b[i];
    // y = P b: interchanges k <-> transp[k] in ascending k (run-time partner: predicated over the static candidates)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int t = transp[k];
        if (t != k) {
            const float yk = y[k];
            float yt = yk;
#pragma unroll
            for (int q = 0; q < N; ++q) if (q == t) yt = y[q];
            y[k] = yt;
#pragma unroll
            for (int q = 0; q < N; ++q) if (q == t) y[q] = yk;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { float a = y[i];
#pragma unroll
        for (int j = 0; j < i; ++j) a -= L[i][j] * y[j];
        y[i] = a; }
#pragma unroll
    for (int i = 0; i < N; ++i) { if (fabsf(L[i][i]) > FLT_MIN) y[i] /= L[i][i]; else y[i] = 0.f; }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) { float a = y[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j) a -= L[j][i] * y[j];
        y[i] = a; }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        const int t = transp[k];
        if (t != k) {
            const float yk = y[k];
            float yt = yk;
#pragma unroll
            for (int q = 0; q < N; ++q) if (q == t) yt = y[q];
            y[k] = yt;
#pragma unroll
            for (int q = 0; q < N; ++q) if (q == t) y[q] = yk;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = y[i];
}
__device__ __forceinline__ void ldlt6_solve(const float* Ain, const float* b, float* x, float* /* LDS scratch of an earlier version: unused */) {
    ldlt6_solve(Ain, b, x);
}

// Rx(a) * Ry(b) * Rz(g) through quaternions.  The half-angle sines and cosines are glibc's sinf / cosf (libm_f32.hpp), the
// functions the reference's AngleAxisf -> Quaternionf conversion calls on the host: the update rotation, and with it the
// refined transform, can then equal the CPU path's bit for bit.
__device__ inline Mat3 euler_xyz(float a, float b, float g) {
    float qxw = lm::cosf_glibc(0.5f * a), qxx = lm::sinf_glibc(0.5f * a);
    float qyw = lm::cosf_glibc(0.5f * b), qyy = lm::sinf_glibc(0.5f * b);
    float qzw = lm::cosf_glibc(0.5f * g), qzz = lm::sinf_glibc(0.5f * g);
    // q1 = qx * qy  (qx = (w,x,0,0), qy = (w,0,y,0))
    float w1 = qxw * qyw - qxx * 0.f - 0.f * qyy - 0.f * 0.f;
    float x1 = qxw * 0.f + qxx * qyw + 0.f * 0.f - 0.f * qyy;
    float y1 = qxw * qyy + 0.f * qyw + 0.f * 0.f - qxx * 0.f;
    float z1 = qxw * 0.f + 0.f * qyw + qxx * qyy - 0.f * 0.f;
    // q = q1 * qz  (qz = (w,0,0,z))
    float w = w1 * qzw - x1 * 0.f - y1 * 0.f - z1 * qzz;
    float x = w1 * 0.f + x1 * qzw + y1 * qzz - z1 * 0.f;
    float y = w1 * 0.f + y1 * qzw + z1 * 0.f - x1 * qzz;
    float z = w1 * qzz + z1 * qzw + x1 * 0.f - y1 * 0.f;
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Mat3 r;
    el(r, 0, 0) = 1.f - (tyy + tzz); el(r, 0, 1) = txy - twz; el(r, 0, 2) = txz + twy;
    el(r, 1, 0) = txy + twz; el(r, 1, 1) = 1.f - (txx + tzz); el(r, 1, 2) = tyz - twx;
    el(r, 2, 0) = txz - twy; el(r, 2, 1) = tyz + twx; el(r, 2, 2) = 1.f - (txx + tyy);
    return r;
}

// C = A * B for column-major 4x4, k ascending.
__device__ inline void mul44(const float* A, const float* B, float* C) {
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float acc = A[0 * 4 + i] * B[j * 4 + 0];
            for (int k = 1; k < 4; ++k) acc = A[k * 4 + i] * B[j * 4 + k] + acc;
            r[j * 4 + i] = acc;
        }
    for (int i = 0; i < 16; ++i) C[i] = r[i];
}

}  // namespace dl
}  // namespace tdv
