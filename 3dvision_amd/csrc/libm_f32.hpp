// sinf / cosf / atanf / atan2f as glibc 2.35 computes them on x86-64.
//
// sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h,
// s_sincosf_data.c: the Arm "optimized routines" algorithm — a degree-7 / degree-8 polynomial in DOUBLE on [-pi/4, pi/4],
// rounded once to float; below 2^-12 the argument itself resp. 1).
//
// Why the product restates a libm function: the reference builds ICP's update rotation from AngleAxisf(x[k], axis)
// (/root/reference/src/registration.cpp:369-371), i.e. from the platform's sinf / cosf of the half angles, and glibc's are
// not correctly rounded (0.56 ULP): a correctly rounded sine differs from glibc's on ~1 % of the arguments near 0.05 rad,
// which is enough to break a bit-for-bit comparison of the refined transform with the reference's CPU path.  The values of
// a double polynomial rounded to float do not depend on how its double operations are scheduled except when the double
// result sits within ~1e-16 (relative) of a float rounding boundary, so what has to be reproduced is the polynomial, not
// the machine code; the fused multiply-adds below are nevertheless placed where GCC places them in glibc's FMA build of
// these files (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c: the same source compiled with -mfma -mavx2), the variant
// glibc's ifunc selects on every x86-64 CPU with FMA.
//
// Pinned by measurement, not by reading: tests/test_libm_restatement.py compiles this header on the host and compares
// it with the running glibc's sinf / cosf on EVERY float of [0, 120) and their negatives (2.2e9 arguments: 0 differences on
// glibc 2.35; the non-FMA scheduling differs on 34 of them, all beyond |x| = 17).  From |x| >= 120 on glibc switches to a
// table-driven reduction that is not restated: the double-precision function rounded once is returned there (an ICP
// increment of 120 rad does not occur).
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#ifdef __HIPCC__
#define TDV_LIBM_HD __host__ __device__ __forceinline__
#else
#define TDV_LIBM_HD static inline
#endif

namespace tdv {
namespace lm {

struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };

TDV_LIBM_HD uint32_t f32_bits(float f) {
#ifdef __HIP_DEVICE_COMPILE__
    return __float_as_uint(f);
#else
    uint32_t u; std::memcpy(&u, &f, 4); return u;
#endif
}
TDV_LIBM_HD uint32_t abstop12(float x) { return (f32_bits(x) >> 20) & 0x7ff; }

// the polynomial of quadrant n (even: sine, odd: cosine); negate = the table of the quadrants with n & 2 set
TDV_LIBM_HD float sincos_poly(double x, double x2, int n, bool negate) {
    const double c0 = negate ? -0x1p0 : 0x1p0, c1 = negate ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = negate ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = negate ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = negate ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = __builtin_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, s1, x);
        return (float)__builtin_fma(x7, t1, s);
    }
    const double x4 = x2 * x2;
    const double t2 = __builtin_fma(x2, c4, c3);
    const double t1 = __builtin_fma(x2, c1, c0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, c2, t1);
    return (float)__builtin_fma(x6, t2, c);
}

// x - n * (pi/2), n = round(x * 2/pi) through a scaled float-to-int conversion (glibc's reduce_fast without TOINT_INTRINSICS)
TDV_LIBM_HD double reduce_fast(double x, int* np) {
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
}

TDV_LIBM_HD float sinf_glibc(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {          // |y| < pi/4
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sincos_poly(x, x * x, 0, false);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        int n;
        x = reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;     // sign of the sine in quadrant n
        return sincos_poly(x * s, x * x, n, (n & 2) != 0);
    }
    return (float)sin((double)y);
}

TDV_LIBM_HD float cosf_glibc(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sincos_poly(x, x * x, 1, false);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        int n;
        x = reduce_fast(x, &n);
        const int m = n + 1;
        const double s = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
        return sincos_poly(x * s, x * x, n ^ 1, (m & 2) != 0);
    }
    return (float)cos((double)y);
}

// ---- atanf / atan2f: glibc's sysdeps/ieee754/flt-32/s_atanf.c and e_atan2f.c (the fdlibm float algorithms: argument reduction
// to [0, 7/16) around atan(0.5), atan(1), atan(1.5), atan(inf) kept as hi + lo floats, an odd / even split degree-11 polynomial; atan2f =
// atanf(|y / x|) moved to the quadrant with pi - (z - pi_lo)).  FLOAT arithmetic throughout - glibc 2.35 has no FMA variant of
// these two on x86-64 - so every operation below must stay an IEEE single-precision operation in this order: the file is compiled
// with -ffp-contract=off and correctly rounded division.  Why the product restates them: the reference's SPFH bins
// theta = atan2(w . n_j, u . n_j) (/root/reference/src/registration.cpp:154-160) and a last-bit difference in theta can move a pair
// into the neighbouring histogram bin - integer work.  Checked against the running libm by tests/test_libm_restatement.py:
// atanf on all 2^32 floats and atan2f on 3e8 pairs (uniform bit patterns, uniform values, near-unit vectors): 0 differences on glibc 2.35.
TDV_LIBM_HD float f32_from_bits(uint32_t u) {
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; std::memcpy(&f, &u, 4); return f;
#endif
}

TDV_LIBM_HD float atanf_glibc(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
                aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
                aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)f32_bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;                   // NaN
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                   // |x| < 0.4375
        if (ix < 0x31000000) return x;                       // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                               // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }     // 7/16 <= |x| < 11/16
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                            // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }     // |x| < 2.4375
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
    const float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

TDV_LIBM_HD float atan2f_glibc(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f32_bits(x), ix = hx & 0x7fffffff, hy = (int32_t)f32_bits(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   // NaN
    if (hx == 0x3f800000) return atanf_glibc(y);            // x = 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);      // 2 * sign(x) + sign(y)
    if (iy == 0) { if (m < 2) return y; return m == 2 ? pi + tiny : -pi - tiny; }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                  // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                   // |y| / x < -2^60
    else z = atanf_glibc(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return f32_from_bits(f32_bits(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

}  // namespace lm
}  // namespace tdv
