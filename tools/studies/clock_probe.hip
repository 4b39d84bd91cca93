// Study: the shader clock under a dense packed-FMA load (what the lane-op "peak" of DESIGN.md is priced with is the guide's nominal
// clock).  Every wave runs a dependent-free stream of v_pk_fma_f32 for a few milliseconds and reads clock64() (shader cycles) and
// wall_clock64() (constant rate, hipDeviceAttributeWallClockRate) before and after.
//   hipcc --offload-arch=gfx950 -O3 tools/studies/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_load(int iters, unsigned long long* out, float* sink) {
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.0f};
    const v2f m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64;
        out[2 * w] = c1 - c0; out[2 * w + 1] = w1 - w0;
    }
}
int main() {
    int wall_khz = 0, clk_khz = 0, cus = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 8, waves = blocks * 4;      // 8 waves per SIMD
    unsigned long long* d; float* sink;
    hipMalloc(&d, (size_t)waves * 16); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h((size_t)waves * 2);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = rep == 0 ? 2000 : 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_load<<<blocks, 256>>>(iters, d, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, wal = 0;
        for (int w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; wal += (double)h[2 * w + 1]; }
        cyc /= waves; wal /= waves;
        const double secs = wal / (wall_khz * 1e3);
        const double pk = (double)iters * 64 * waves;                 // packed instructions issued
        printf("rep %d: kernel %.3f ms; per wave %.0f shader cycles in %.0f wall ticks (%d kHz) = %.3f ms -> shader clock %.0f MHz (attribute %d MHz); "
               "%.1f T lane-ops/s (2 per lane and packed instruction), %d CUs\n",
               rep, ms, cyc, wal, wall_khz, secs * 1e3, cyc / secs * 1e-6, clk_khz / 1000, pk * 64 * 2 / (ms * 1e-3) * 1e-12, cus);
    }
    return 0;
}
