// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under 3dvision_amd/ (the product)
// may include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use the oracle, and only as the checker.
//
// Small dense solvers restating the Eigen routines that
// /root/reference/src/registration.cpp calls.  Eigen is a third-party
// dependency of the reference that is NOT vendored under /root/reference and
// is not installed in this image (CMakeLists.txt:27 asks for "Eigen3 3.3" with
// no pinned version).  What follows restates the published algorithms of
// Eigen 3.4.0 (the version Ubuntu 22.04 / g++ 11 ships), file by file:
//   Eigen/src/Jacobi/Jacobi.h                 makeJacobi, makeGivens, rotations
//   Eigen/src/SVD/JacobiSVD.h                 two-sided Jacobi SVD, real_2x2_jacobi_svd
//   Eigen/src/Eigenvalues/Tridiagonalization.h   3x3 real specialisation
//   Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h   implicit symmetric QR
//   Eigen/src/Cholesky/LDLT.h                 pivoted LDLT, pseudo-inverse of D
//   Eigen/src/Geometry/{AngleAxis,Quaternion}.h
//   Eigen/src/Core/Redux.h                    3-term sums are  c0 + (c1 + c2)
// PARITY UNPINNED: the reference holds no golden vectors and cannot be built
// here, so bitwise agreement with Eigen's internals cannot be checked.  Results
// that depend on these solvers are compared under the tolerances of
// BASELINE.json (1e-4 rotation, 1e-3 mm translation); integer outputs are
// compared exactly against this restatement.
//
// Compile with -ffp-contract=off (x86-64 baseline has no FMA; keep it explicit).
#pragma once
#include <cmath>
#include <cfloat>
#include <algorithm>

namespace orc {

struct V3 { float x, y, z; };

// Eigen redux_novec_unroller<.,.,0,3>: func(c0, func(c1, c2)).
static inline float sum3(float a, float b, float c) { return a + (b + c); }
static inline float dot3(const float* a, const float* b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }

// Column-major 3x3 (Eigen default): m[c*3 + r].
struct M3 {
    float m[9];
    float& operator()(int r, int c) { return m[c * 3 + r]; }
    float operator()(int r, int c) const { return m[c * 3 + r]; }
    static M3 identity() { M3 a; for (int i = 0; i < 9; ++i) a.m[i] = 0.f; a.m[0] = a.m[4] = a.m[8] = 1.f; return a; }
};

// Coefficient-based lazy product (Eigen ProductEvaluators.h, CoeffBasedProductMode):
// coeff(i,j) = (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum()
static inline M3 mul(const M3& a, const M3& b) {
    M3 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            r(i, j) = sum3(a(i, 0) * b(0, j), a(i, 1) * b(1, j), a(i, 2) * b(2, j));
    return r;
}
static inline M3 transpose(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
    return r;
}
static inline void mulv(const M3& a, const float* v, float* out) {
    for (int i = 0; i < 3; ++i) out[i] = sum3(a(i, 0) * v[0], a(i, 1) * v[1], a(i, 2) * v[2]);
}
// Eigen Determinant.h determinant_impl<Derived,3>
static inline float det3(const M3& m) {
    auto h = [&](int a, int b, int c) { return m(0, a) * (m(1, b) * m(2, c) - m(1, c) * m(2, b)); };
    return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

struct Rot { float c, s; };  // Eigen JacobiRotation (real)

// JacobiRotation::makeJacobi(x, y, z) for the 2x2 self-adjoint [[x,y],[y,z]].
static inline Rot make_jacobi(float x, float y, float z) {
    Rot r;
    float deno = 2.f * std::fabs(y);
    if (deno < FLT_MIN) { r.c = 1.f; r.s = 0.f; return r; }
    float tau = (x - z) / deno;
    float w = std::sqrt(tau * tau + 1.f);
    float t;
    if (tau > 0.f) t = 1.f / (tau + w);
    else           t = 1.f / (tau - w);
    float sign_t = t > 0.f ? 1.f : -1.f;
    float n = 1.f / std::sqrt(t * t + 1.f);
    r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    r.c = n;
    return r;
}
// JacobiRotation::makeGivens(p, q) real case.
static inline Rot make_givens(float p, float q) {
    Rot r;
    if (q == 0.f) { r.c = p < 0.f ? -1.f : 1.f; r.s = 0.f; }
    else if (p == 0.f) { r.c = 0.f; r.s = q < 0.f ? 1.f : -1.f; }
    else if (std::fabs(p) > std::fabs(q)) {
        float t = q / p; float u = std::sqrt(1.f + t * t); if (p < 0.f) u = -u;
        r.c = 1.f / u; r.s = -t * r.c;
    } else {
        float t = p / q; float u = std::sqrt(1.f + t * t); if (q < 0.f) u = -u;
        r.s = -1.f / u; r.c = -t * r.s;
    }
    return r;
}
static inline Rot rot_transpose(Rot a) { return Rot{a.c, -a.s}; }
// JacobiRotation::operator* (real): (c1*c2 - s1*s2, c1*s2 + s1*c2)
static inline Rot rot_mul(Rot a, Rot b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
// apply_rotation_in_the_plane(x, y, j): x' = c x + s y ; y' = -s x + c y
template <class M> static inline void apply_left(M& w, int p, int q, Rot j, int n) {
    for (int i = 0; i < n; ++i) {
        float xi = w(p, i), yi = w(q, i);
        w(p, i) = j.c * xi + j.s * yi;
        w(q, i) = -j.s * xi + j.c * yi;
    }
}
// applyOnTheRight(p,q,j) == rotate columns p,q with j.transpose()
template <class M> static inline void apply_right(M& w, int p, int q, Rot j, int n) {
    Rot jt = rot_transpose(j);
    for (int i = 0; i < n; ++i) {
        float xi = w(i, p), yi = w(i, q);
        w(i, p) = jt.c * xi + jt.s * yi;
        w(i, q) = -jt.s * xi + jt.c * yi;
    }
}

// JacobiSVD<Matrix3f>(H, ComputeFullU|ComputeFullV) — registration.cpp:255,388.
struct SVD3 { M3 U, V; float s[3]; };
static inline SVD3 jacobi_svd3(const M3& A) {
    SVD3 out;
    const float precision = 2.f * FLT_EPSILON;
    const float considerAsZero = FLT_MIN;
    float scale = 0.f;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(A.m[i]));
    if (scale == 0.f) scale = 1.f;
    M3 W;
    for (int i = 0; i < 9; ++i) W.m[i] = A.m[i] / scale;
    out.U = M3::identity(); out.V = M3::identity();
    float maxDiag = std::max(std::fabs(W(0, 0)), std::max(std::fabs(W(1, 1)), std::fabs(W(2, 2))));
    bool finished = false;
    while (!finished) {
        finished = true;
        for (int p = 1; p < 3; ++p) {
            for (int q = 0; q < p; ++q) {
                float threshold = std::max(considerAsZero, precision * maxDiag);
                if (std::fabs(W(p, q)) > threshold || std::fabs(W(q, p)) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd(W, p, q, &j_left, &j_right)
                    float m00 = W(p, p), m01 = W(p, q), m10 = W(q, p), m11 = W(q, q);
                    Rot rot1;
                    float t = m00 + m11;
                    float d = m10 - m01;
                    if (std::fabs(d) < FLT_MIN) { rot1.s = 0.f; rot1.c = 1.f; }
                    else {
                        float u = t / d;
                        float tmp = std::sqrt(1.f + u * u);
                        rot1.s = 1.f / tmp;
                        rot1.c = u / tmp;
                    }
                    // m.applyOnTheLeft(0,1,rot1)
                    float a00 = rot1.c * m00 + rot1.s * m10, a01 = rot1.c * m01 + rot1.s * m11;
                    float a11 = -rot1.s * m01 + rot1.c * m11;
                    Rot j_right = make_jacobi(a00, a01, a11);
                    Rot j_left = rot_mul(rot1, rot_transpose(j_right));
                    apply_left(W, p, q, j_left, 3);
                    apply_right(out.U, p, q, rot_transpose(j_left), 3);
                    apply_right(W, p, q, j_right, 3);
                    apply_right(out.V, p, q, j_right, 3);
                    maxDiag = std::max(maxDiag, std::max(std::fabs(W(p, p)), std::fabs(W(q, q))));
                }
            }
        }
    }
    for (int i = 0; i < 3; ++i) {
        float a = W(i, i);
        out.s[i] = std::fabs(a);
        if (a < 0.f) for (int r = 0; r < 3; ++r) out.U(r, i) = -out.U(r, i);
    }
    for (int i = 0; i < 3; ++i) out.s[i] *= scale;
    // sort singular values in descending order, permuting U and V columns
    for (int i = 0; i < 3; ++i) {
        int pos = i; float best = out.s[i];
        for (int k = i + 1; k < 3; ++k) if (out.s[k] > best) { best = out.s[k]; pos = k; }
        if (best == 0.f) break;
        if (pos != i) {
            std::swap(out.s[i], out.s[pos]);
            for (int r = 0; r < 3; ++r) { std::swap(out.U(r, i), out.U(r, pos)); std::swap(out.V(r, i), out.V(r, pos)); }
        }
    }
    return out;
}

// Kabsch step shared by registration.cpp:255-262 and :388-394:
// R = V U^T ; if det(R) < 0 negate V.col(2) and recompute.
static inline M3 kabsch_rotation(const M3& H) {
    SVD3 svd = jacobi_svd3(H);
    M3 Ut = transpose(svd.U);
    M3 R = mul(svd.V, Ut);
    if (det3(R) < 0.f) {
        M3 V = svd.V;
        for (int r = 0; r < 3; ++r) V(r, 2) *= -1.f;
        R = mul(V, Ut);
    }
    return R;
}

// SelfAdjointEigenSolver<Matrix3f>(cov) — registration.cpp:122-123.
// Returns eigenvalues ascending and eigenvectors as columns.
static inline float pos_hypot(float x, float y) {  // Eigen numext::hypot (positive_real_hypot)
    x = std::fabs(x); y = std::fabs(y);
    float p = std::max(x, y);
    if (p == 0.f) return 0.f;
    float qp = std::min(y, x) / p;
    return p * std::sqrt(1.f + qp * qp);
}
struct Eig3 { float w[3]; M3 V; bool ok; };
static inline Eig3 self_adjoint_eig3(const M3& A) {
    Eig3 out;
    // mat = lower triangle of A, scaled into [-1,1]
    float m00 = A(0, 0), m10 = A(1, 0), m20 = A(2, 0), m11 = A(1, 1), m21 = A(2, 1), m22 = A(2, 2);
    float scale = std::max(std::max(std::fabs(m00), std::fabs(m10)), std::max(std::max(std::fabs(m20), std::fabs(m11)), std::max(std::fabs(m21), std::fabs(m22))));
    if (scale == 0.f) scale = 1.f;
    m00 /= scale; m10 /= scale; m20 /= scale; m11 /= scale; m21 /= scale; m22 /= scale;
    float diag[3], sub[2];
    M3 Q;
    // tridiagonalization_inplace_selector<MatrixType,3,false>
    diag[0] = m00;
    float v1norm2 = m20 * m20;
    if (v1norm2 <= FLT_MIN) {
        diag[1] = m11; diag[2] = m22; sub[0] = m10; sub[1] = m21;
        Q = M3::identity();
    } else {
        float beta = std::sqrt(m10 * m10 + v1norm2);
        float invBeta = 1.f / beta;
        float m01 = m10 * invBeta;
        float m02 = m20 * invBeta;
        float q = 2.f * m01 * m21 + m02 * (m22 - m11);
        diag[1] = m11 + m02 * q;
        diag[2] = m22 - m02 * q;
        sub[0] = beta;
        sub[1] = m21 - m01 * q;
        Q = M3::identity();
        Q(1, 1) = m01; Q(1, 2) = m02; Q(2, 1) = m02; Q(2, 2) = -m01;
    }
    // computeFromTridiagonal_impl
    const int n = 3, maxIterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const float considerAsZero = FLT_MIN;
    const float precision_inv = 1.f / FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            if (std::fabs(sub[i]) < considerAsZero) sub[i] = 0.f;
            else {
                const float scaled = precision_inv * sub[i];
                if (scaled * scaled <= (std::fabs(diag[i]) + std::fabs(diag[i + 1]))) sub[i] = 0.f;
            }
        }
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        // tridiagonal_qr_step
        float td = (diag[end - 1] - diag[end]) * 0.5f;
        float e = sub[end - 1];
        float mu = diag[end];
        if (td == 0.f) mu -= std::fabs(e);
        else if (e != 0.f) {
            const float e2 = e * e;
            const float h = pos_hypot(td, e);
            if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? h : -h)) / e);
            else           mu -= e2 / (td + (td > 0.f ? h : -h));
        }
        float x = diag[start] - mu;
        float z = sub[start];
        for (int k = start; k < end && z != 0.f; ++k) {
            Rot rot = make_givens(x, z);
            float sdk = rot.s * diag[k] + rot.c * sub[k];
            float dkp1 = rot.s * sub[k] + rot.c * diag[k + 1];
            diag[k] = rot.c * (rot.c * diag[k] - rot.s * sub[k]) - rot.s * (rot.c * sub[k] - rot.s * diag[k + 1]);
            diag[k + 1] = rot.s * sdk + rot.c * dkp1;
            sub[k] = rot.c * sdk - rot.s * dkp1;
            if (k > start) sub[k - 1] = rot.c * sub[k - 1] - rot.s * z;
            x = sub[k];
            if (k < end - 1) { z = -rot.s * sub[k + 1]; sub[k + 1] = rot.c * sub[k + 1]; }
            apply_right(Q, k, k + 1, rot, 3);
        }
    }
    out.ok = iter <= maxIterations * n;
    if (out.ok) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0; float mn = diag[i];
            for (int j = 1; j < n - i; ++j) if (diag[i + j] < mn) { mn = diag[i + j]; k = j; }
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r) std::swap(Q(r, i), Q(r, k + i));
            }
        }
    }
    for (int i = 0; i < 3; ++i) out.w[i] = diag[i] * scale;
    out.V = Q;
    return out;
}

// Matrix<float,6,6>::ldlt().solve(b) — registration.cpp:366.
// A is row-major 6x6 (symmetric; only the lower triangle is read, as Eigen does).
static inline void ldlt6_solve(const float* Ain, const float* b, float* x) {
    const int N = 6;
    float mat[6][6];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) mat[i][j] = Ain[i * 6 + j];
    int transp[6];
    float temp[6];
    bool all_zero_diag = false;
    for (int k = 0; k < N; ++k) {
        int big = k; float bv = std::fabs(mat[k][k]);
        for (int i = k + 1; i < N; ++i) if (std::fabs(mat[i][i]) > bv) { bv = std::fabs(mat[i][i]); big = i; }
        transp[k] = big;
        if (k != big) {
            int s = N - big - 1;
            for (int j = 0; j < k; ++j) std::swap(mat[k][j], mat[big][j]);
            for (int i = 0; i < s; ++i) std::swap(mat[big + 1 + i][k], mat[big + 1 + i][big]);
            std::swap(mat[k][k], mat[big][big]);
            for (int i = k + 1; i < big; ++i) { float tmp = mat[i][k]; mat[i][k] = mat[big][i]; mat[big][i] = tmp; }
        }
        int rs = N - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; ++j) temp[j] = mat[j][j] * mat[k][j];
            float acc = 0.f;
            for (int j = 0; j < k; ++j) acc += mat[k][j] * temp[j];
            mat[k][k] -= acc;
            for (int i = 0; i < rs; ++i) {
                float a2 = 0.f;
                for (int j = 0; j < k; ++j) a2 += mat[k + 1 + i][j] * temp[j];
                mat[k + 1 + i][k] -= a2;
            }
        }
        float realAkk = mat[k][k];
        bool pivot_is_valid = std::fabs(realAkk) > 0.f;
        if (k == 0 && !pivot_is_valid) {
            for (int j = 0; j < N; ++j) transp[j] = j;
            all_zero_diag = true;
            break;
        }
        if (rs > 0 && pivot_is_valid) for (int i = 0; i < rs; ++i) mat[k + 1 + i][k] /= realAkk;
    }
    (void)all_zero_diag;
    float y[6];
    for (int i = 0; i < N; ++i) y[i] = b[i];
    for (int k = 0; k < N; ++k) if (transp[k] != k) std::swap(y[k], y[transp[k]]);         // P b
    for (int i = 0; i < N; ++i) { float a = y[i]; for (int j = 0; j < i; ++j) a -= mat[i][j] * y[j]; y[i] = a; }  // L^-1
    for (int i = 0; i < N; ++i) { if (std::fabs(mat[i][i]) > FLT_MIN) y[i] /= mat[i][i]; else y[i] = 0.f; }       // D^+
    for (int i = N - 1; i >= 0; --i) { float a = y[i]; for (int j = i + 1; j < N; ++j) a -= mat[j][i] * y[j]; y[i] = a; }  // L^-T
    for (int k = N - 1; k >= 0; --k) if (transp[k] != k) std::swap(y[k], y[transp[k]]);    // P^T
    for (int i = 0; i < N; ++i) x[i] = y[i];
}

// (AngleAxisf(a,X) * AngleAxisf(b,Y) * AngleAxisf(g,Z)).matrix() — registration.cpp:369-371.
struct Quat { float w, x, y, z; };
static inline Quat qmul(const Quat& a, const Quat& b) {
    return Quat{
        a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
        a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
        a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
        a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
static inline M3 euler_xyz_matrix(float a, float b, float g) {
    Quat qx{std::cos(0.5f * a), std::sin(0.5f * a), 0.f, 0.f};
    Quat qy{std::cos(0.5f * b), 0.f, std::sin(0.5f * b), 0.f};
    Quat qz{std::cos(0.5f * g), 0.f, 0.f, std::sin(0.5f * g)};
    Quat q = qmul(qmul(qx, qy), qz);
    const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r(0, 0) = 1.f - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1.f - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1.f - (txx + tyy);
    return r;
}

// 4x4 column-major product T = A * B, k ascending (Eigen packet path: pmul then pmadd k=1..3).
static inline void mul44(const float* A, const float* B, float* C) {
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float acc = A[0 * 4 + i] * B[j * 4 + 0];
            for (int k = 1; k < 4; ++k) acc = A[k * 4 + i] * B[j * 4 + k] + acc;
            r[j * 4 + i] = acc;
        }
    for (int i = 0; i < 16; ++i) C[i] = r[i];
}

}  // namespace orc
