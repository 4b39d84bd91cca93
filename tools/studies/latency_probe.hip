// Dependent-load latency on this part: one lane (resp. many waves) chases pointers through arrays of several sizes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe tools/studies/latency_probe.hip && /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void k_chase(const unsigned* __restrict__ next, int steps, unsigned stride_elems, unsigned* out) {
    unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * stride_elems;
    for (int s = 0; s < steps; ++s) p = next[p];
    if (p == 0xffffffffu) out[0] = p;
}
__global__ void k_empty(unsigned* out) { if (out == nullptr) out[0] = 1; }

int main() {
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned* d_out; hipMalloc(&d_out, 4);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, s); k_empty<<<1, 64, 0, s>>>(d_out); hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    printf("{\"empty_kernel_between_events_us\": %.2f}\n", ms * 1e3);
    for (size_t bytes : {size_t(256) << 10, size_t(2) << 20, size_t(16) << 20, size_t(512) << 20}) {
        const size_t n = bytes / 4;
        std::vector<unsigned> perm(n), nxt(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
        for (size_t i = 0; i < n; ++i) nxt[perm[i]] = perm[(i + 1) % n];     // one random cycle over the whole array
        unsigned* d; hipMalloc(&d, bytes); hipMemcpy(d, nxt.data(), bytes, hipMemcpyHostToDevice);
        for (int waves : {1, 256 * 4, 256 * 4 * 8}) {
            const int steps = 2000;
            const int threads = waves == 1 ? 1 : 64;
            k_chase<<<waves == 1 ? 1 : waves, threads, 0, s>>>(d, 100, (unsigned)(n / (size_t)(waves * 64 + 1)), d_out);   // warm
            hipEventRecord(a, s); k_chase<<<waves == 1 ? 1 : waves, threads, 0, s>>>(d, steps, (unsigned)(n / (size_t)(waves * 64 + 1)), d_out);
            hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("{\"array_MB\": %.2f, \"waves\": %d, \"lanes\": %d, \"ns_per_dependent_load\": %.1f}\n", bytes / 1048576.0, waves, threads, ms * 1e6 / steps);
        }
        hipFree(d);
    }
    return 0;
}
