// sinf / cosf as glibc 2.35 computes them on x86-64 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h,
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

}  // namespace lm
}  // namespace tdv
