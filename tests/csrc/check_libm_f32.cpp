// Test infrastructure: compares the product's restatement of glibc's sinf / cosf (3dvision_amd/csrc/libm_f32.hpp, compiled
// here for the host) with the running libm on every float of [lo, hi) taken with the given stride, both signs.
// usage: check_libm_f32 lo hi stride   -> prints "tested N bad_sin A bad_cos B"
#include "libm_f32.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    const float lo = argc > 1 ? strtof(argv[1], nullptr) : 0.f, hi = argc > 2 ? strtof(argv[2], nullptr) : 120.f;
    const unsigned stride = argc > 3 ? (unsigned)strtoul(argv[3], nullptr, 10) : 1u;
    unsigned long long tested = 0, bad_s = 0, bad_c = 0;
    for (uint32_t u = tdv::lm::f32_bits(lo); u < tdv::lm::f32_bits(hi); u += stride) {
        float f; std::memcpy(&f, &u, 4);
        for (int sg = 0; sg < 2; ++sg) {
            const float y = sg ? -f : f;
            ++tested;
            if (tdv::lm::f32_bits(sinf(y)) != tdv::lm::f32_bits(tdv::lm::sinf_glibc(y))) { if (bad_s < 4) printf("sin %a: libm %a restated %a\n", y, sinf(y), tdv::lm::sinf_glibc(y)); ++bad_s; }
            if (tdv::lm::f32_bits(cosf(y)) != tdv::lm::f32_bits(tdv::lm::cosf_glibc(y))) { if (bad_c < 4) printf("cos %a: libm %a restated %a\n", y, cosf(y), tdv::lm::cosf_glibc(y)); ++bad_c; }
        }
    }
    printf("tested %llu bad_sin %llu bad_cos %llu\n", tested, bad_s, bad_c);
    return 0;
}
