// What does code size cost a short kernel?  The same 4096 dependent FMAs as a loop (a few instructions of code) and as
// straight-line code (32 KB), one workgroup and 1024 workgroups, each launch timed on its own between events.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/icache_probe tools/studies/icache_probe.hip && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R4096(x) R16(R256(x))
__global__ void k_loop(float* out, float a, float b, int n) {
    float x = out[threadIdx.x];
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, a, b);
    out[threadIdx.x] = x;
}
__global__ void k_flat(float* out, float a, float b) {
    float x = out[threadIdx.x];
    R4096(x = __builtin_fmaf(x, a, b); asm volatile("" : "+v"(x));)
    out[threadIdx.x] = x;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    for (int wgs : {1, 1024}) {
        float ms_loop = 0, ms_flat = 0, ms;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0, s); k_loop<<<wgs, 64, 0, s>>>(d, 0.5f, 1.f, 4096); hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) ms_loop += ms / 4;
            hipEventRecord(e0, s); k_flat<<<wgs, 64, 0, s>>>(d, 0.5f, 1.f); hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) ms_flat += ms / 4;
        }
        printf("{\"workgroups\": %d, \"loop_us\": %.2f, \"straight_line_32KB_us\": %.2f}\n", wgs, ms_loop * 1e3, ms_flat * 1e3);
    }
    return 0;
}
