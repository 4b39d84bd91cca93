// Test infrastructure: compares the product's restatement of glibc's float functions (3dvision_amd/csrc/libm_f32.hpp, compiled
// here for the host) with the running libm.
//   check_libm_f32 sincos lo hi stride   every float of [lo, hi) taken with the stride, both signs: sinf, cosf
//   check_libm_f32 atan stride           every stride-th of the 2^32 bit patterns: atanf
//   check_libm_f32 atan2 count           count pseudo-random pairs (uniform bit patterns, uniform values in (-1, 1), small exponents,
//                                        x near +-1 - the shape of SPFH's arguments): atan2f
// prints "tested N bad A [bad B]"
#include "libm_f32.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using tdv::lm::f32_bits;
static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static bool same(float a, float b) { return f32_bits(a) == f32_bits(b) || (a != a && b != b); }
int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "sincos";
    unsigned long long tested = 0, bad = 0, bad2 = 0;
    if (!std::strcmp(mode, "sincos")) {
        const float lo = argc > 2 ? strtof(argv[2], nullptr) : 0.f, hi = argc > 3 ? strtof(argv[3], nullptr) : 120.f;
        const unsigned stride = argc > 4 ? (unsigned)strtoul(argv[4], nullptr, 10) : 1u;
        for (uint32_t u = f32_bits(lo); u < f32_bits(hi); u += stride) {
            const float f = from_bits(u);
            for (int sg = 0; sg < 2; ++sg) {
                const float y = sg ? -f : f;
                ++tested;
                if (!same(sinf(y), tdv::lm::sinf_glibc(y))) { if (bad < 4) printf("sin %a: libm %a restated %a\n", y, sinf(y), tdv::lm::sinf_glibc(y)); ++bad; }
                if (!same(cosf(y), tdv::lm::cosf_glibc(y))) { if (bad2 < 4) printf("cos %a: libm %a restated %a\n", y, cosf(y), tdv::lm::cosf_glibc(y)); ++bad2; }
            }
        }
        printf("tested %llu bad_sin %llu bad_cos %llu\n", tested, bad, bad2);
    } else if (!std::strcmp(mode, "atan")) {
        const unsigned long long stride = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1ull;
        for (unsigned long long u = 0; u <= 0xffffffffull; u += stride) {
            const float x = from_bits((uint32_t)u);
            ++tested;
            if (!same(atanf(x), tdv::lm::atanf_glibc(x))) { if (bad < 4) printf("atanf %a: libm %a restated %a\n", x, atanf(x), tdv::lm::atanf_glibc(x)); ++bad; }
        }
        printf("tested %llu bad_atanf %llu\n", tested, bad);
    } else {
        const long count = argc > 2 ? strtol(argv[2], nullptr, 10) : 1000000L;
        unsigned long long st = 88172645463325252ull;
        for (long i = 0; i < count; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const uint32_t a = (uint32_t)st, b = (uint32_t)(st >> 32);
            float y, x;
            switch (i & 3) {
                case 0: y = from_bits(a); x = from_bits(b); break;
                case 1: y = (float)((int32_t)a) * 4.6566e-10f; x = (float)((int32_t)b) * 4.6566e-10f; break;
                case 2: y = from_bits((a & 0x807fffffu) | (0x3e000000u + (((a >> 23) & 15u) << 23))); x = from_bits((b & 0x807fffffu) | (0x3e000000u + (((b >> 23) & 15u) << 23))); break;
                default: y = (float)((int32_t)a) * 4.6566e-10f; x = from_bits((b & 0x80000000u) | (0x3f800000u - (b & 0xfffffu))); break;
            }
            ++tested;
            if (!same(atan2f(y, x), tdv::lm::atan2f_glibc(y, x))) { if (bad < 4) printf("atan2f(%a, %a): libm %a restated %a\n", y, x, atan2f(y, x), tdv::lm::atan2f_glibc(y, x)); ++bad; }
        }
        printf("tested %llu bad_atan2f %llu\n", tested, bad);
    }
    return 0;
}
