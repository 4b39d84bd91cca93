// Do matrix-core and vector instructions of one SIMD overlap?  Waves with an MFMA-only loop, waves with a packed-FMA-only loop,
// and both kinds side by side (even / odd workgroups of 8 waves: two waves of each kind per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_overlap tools/studies/mfma_valu_overlap.hip && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: f32 MFMA 32x32x2, 1: bf16 MFMA 32x32x16
__device__ __forceinline__ void mfma_loop(float* out, int iters) {
    v16f acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const v4f fa = {a, b, a, b};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
        } else {
            const v8bf x = __builtin_bit_cast(v8bf, fa);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc3, 0, 0, 0);
        }
    }
    out[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
template <int VK> __device__ __forceinline__ void valu_loop_k(float* out, int iters) {   // 1: unpacked v_fma_f32, 2: v_add_u32 / v_xor
    float x0 = 1.f, x1 = 2.f, x2 = 3.f, x3 = 4.f;
    unsigned u0 = threadIdx.x, u1 = 2, u2 = 3, u3 = 4;
    const float a = 0.999f + threadIdx.x * 1e-6f, b = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (VK == 1) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
            else { u0 = (u0 + u1) ^ 0x9e37u; u1 = (u1 + u2) ^ 0x79b9u; u2 = (u2 + u3) ^ 0x7f4au; u3 = (u3 + u0) ^ 0x7c15u; }
        }
    }
    out[threadIdx.x] = x0 + x1 + x2 + x3 + (float)(u0 ^ u1 ^ u2 ^ u3);
}
__device__ __forceinline__ void valu_loop(float* out, int iters) {
    v2f x0 = {1.f, 2.f}, x1 = {3.f, 4.f}, x2 = {5.f, 6.f}, x3 = {7.f, 8.f};
    const v2f a = {0.999f + threadIdx.x * 1e-6f, 0.998f}, b = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x0 = __builtin_elementwise_fma(x0, a, b); x1 = __builtin_elementwise_fma(x1, a, b);
            x2 = __builtin_elementwise_fma(x2, a, b); x3 = __builtin_elementwise_fma(x3, a, b);
        }
    }
    out[threadIdx.x] = x0.x + x1.y + x2.x + x3.y;
}
template <int KIND, int VK = 0>
__global__ __launch_bounds__(512) void k_mix(float* out, int mode, int mfma_iters, int valu_iters) {   // mode 0: MFMA waves only do work, 1: VALU waves only, 2: both
    // a workgroup is of one kind (the first 256: MFMA, the next 256: vector): its 8 waves spread over the 4 SIMDs of its CU, and
    // with workgroups dealt round-robin over the XCDs and their CUs every CU holds one workgroup of each kind, every SIMD 2 waves
    // of each kind.  (Kinds by workgroup parity put each kind on half the XCDs; kinds by wave parity left the placement unknown.)
    float* o = out + (size_t)blockIdx.x * 512;
    if (((blockIdx.x >> 8) & 1) == 0) { if (mode != 1) mfma_loop<KIND>(o, mfma_iters); }
    else { if (mode != 0) { if (VK == 0) valu_loop(o, valu_iters); else valu_loop_k<VK>(o, valu_iters); } }
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 512;   // 2 workgroups of 8 waves per CU: per SIMD 2 MFMA waves + 2 VALU waves
    float* d; hipMalloc(&d, (size_t)wgs * 512 * 4);
    // 4 MFMAs of 64 (f32) resp. 32 (bf16) cycles per iteration; 32 packed FMAs of 4 cycles per iteration
    for (int kind = 0; kind < 2; ++kind) {
        const int mi = 20000, vi = kind == 0 ? 40000 : 20000;   // nominal cycles per wave: f32 20000*256 = 5.1 M, bf16 2.6 M; vector 40000*128 = 5.1 M resp. 2.6 M
        float t[3];
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, s);
                if (kind == 0) k_mix<0><<<wgs, 512, 0, s>>>(d, mode, mi, vi); else k_mix<1><<<wgs, 512, 0, s>>>(d, mode, mi, vi);
                hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&t[mode], e0, e1);
            }
        }
        printf("{\"mfma\": \"%s\", \"mfma_waves_only_ms\": %.3f, \"vector_waves_only_ms\": %.3f, \"both_ms\": %.3f, \"sum_ms\": %.3f}\n",
               kind == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16", t[0], t[1], t[2], t[0] + t[1]);
    }
    for (int vk = 1; vk <= 2; ++vk) {   // bf16 MFMA beside unpacked FMAs resp. integer operations
        const int mi = 20000, vi = 20000;
        float t[3];
        for (int mode = 0; mode < 3; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, s);
                if (vk == 1) k_mix<1, 1><<<wgs, 512, 0, s>>>(d, mode, mi, vi); else k_mix<1, 2><<<wgs, 512, 0, s>>>(d, mode, mi, vi);
                hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&t[mode], e0, e1);
            }
        printf("{\"mfma\": \"v_mfma_f32_32x32x16_bf16\", \"vector_work\": \"%s\", \"mfma_waves_only_ms\": %.3f, \"vector_waves_only_ms\": %.3f, \"both_ms\": %.3f, \"sum_ms\": %.3f}\n",
               vk == 1 ? "v_fma_f32" : "v_add_u32 + v_xor_b32", t[0], t[1], t[2], t[0] + t[1]);
    }
    return 0;
}