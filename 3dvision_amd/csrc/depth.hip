// Depth scale + mask and depth -> point-cloud unprojection on gfx950 (HBM-bound scans).
//
// Replaces GPUDepth::preprocess (/root/reference/src/gpu_impl.cpp:28-66, kernel
// cuda/depth_processing.cu:10-30) and GPUPointCloud::generate (gpu_impl.cpp:69-128, kernel
// cuda/pointcloud.cu:11-51); results follow the CPU branches src/pipeline.cpp:46-54 and :61-84:
//   depth  = float(raw) * float(1.0/scale), zero where the mask rejects the pixel
//   keep 0 < z <= zmax ; x = (u - cx) * z / fx ; y = (v - cy) * z / fy ; colour = BGR->RGB / 255
// and — unlike the reference's CUDA kernel, whose global atomicAdd makes the order
// nondeterministic — points are emitted in the CPU's row-major scan order:
//   pass 1 (k_valid_count)  : 1024 pixels per workgroup, ballot/popcount -> one count per block
//   pass 2 (k_block_scan)   : exclusive scan of the block counts (one workgroup)
//   pass 3 (k_emit)         : recompute validity, wave-ballot prefix + block offset -> slot
// Algorithmic HBM bytes per frame (fused u16+mask+bgr path): 2x(2+1) in + 3 in + 24n out.
#include "tdv_internal.hpp"
#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstring>
#include <vector>

namespace tdv {

constexpr int DP_BLOCK = 256;
constexpr int DP_PX_PER_THREAD = 4;
constexpr int DP_PX_PER_BLOCK = DP_BLOCK * DP_PX_PER_THREAD;

// mask rule: reference CPU (> 10, src/pipeline.cpp:51), reference CUDA (!= 0, cuda/depth_processing.cu:22), or
// label image (pixel value == instance label; SURVEY.md 8f N2: one u8 image instead of B full-frame masks)
__device__ __forceinline__ bool mask_keeps(uint8_t m, int mask_mode) {
    if (mask_mode == TDV_MASK_THRESHOLD10) return m > 10;
    if (mask_mode == TDV_MASK_NONZERO) return m != 0;
    return (int)m == mask_mode - TDV_MASK_LABEL_BASE;
}

__device__ __forceinline__ float scaled_depth(const uint16_t* __restrict__ raw, const uint8_t* __restrict__ mask,
                                              size_t i, float inv_scale, int mask_mode) {
    float v = (float)raw[i] * inv_scale;
    if (mask && !mask_keeps(mask[i], mask_mode)) v = 0.f;
    return v;
}

__global__ __launch_bounds__(DP_BLOCK)
void k_depth_preprocess(const uint16_t* __restrict__ raw, const uint8_t* __restrict__ mask, size_t n,
                        float inv_scale, int mask_mode, float* __restrict__ out) {
    // 4 consecutive pixels per lane: 8 B of depth, 4 B of mask in, 16 B out
    size_t i4 = ((size_t)blockIdx.x * DP_BLOCK + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        ushort4 r = *reinterpret_cast<const ushort4*>(raw + i4);
        float4 o = make_float4((float)r.x * inv_scale, (float)r.y * inv_scale, (float)r.z * inv_scale, (float)r.w * inv_scale);
        if (mask) {
            uchar4 m = *reinterpret_cast<const uchar4*>(mask + i4);
            if (!mask_keeps(m.x, mask_mode)) o.x = 0.f;
            if (!mask_keeps(m.y, mask_mode)) o.y = 0.f;
            if (!mask_keeps(m.z, mask_mode)) o.z = 0.f;
            if (!mask_keeps(m.w, mask_mode)) o.w = 0.f;
        }
        *reinterpret_cast<float4*>(out + i4) = o;
    } else {
        for (size_t i = i4; i < n; ++i) out[i] = scaled_depth(raw, mask, i, inv_scale, mask_mode);
    }
}

// depth of pixel i from either the raw u16 (+mask) image or an f32 depth image
template <bool RAW>
__device__ __forceinline__ float pixel_depth(const uint16_t* __restrict__ raw, const float* __restrict__ depth,
                                             const uint8_t* __restrict__ mask, size_t i, float inv_scale, int mask_mode) {
    if (RAW) return scaled_depth(raw, mask, i, inv_scale, mask_mode);
    return depth[i];
}

template <bool RAW>
__global__ __launch_bounds__(DP_BLOCK)
void k_valid_count(const uint16_t* __restrict__ raw, const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                   size_t n, float inv_scale, int mask_mode, float zmax, int* __restrict__ block_counts) {
    // pixel p of the block = threadIdx.x + k*256 (coalesced), k = 0..3
    const size_t base = (size_t)blockIdx.x * DP_PX_PER_BLOCK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        size_t i = base + k * DP_BLOCK + threadIdx.x;
        if (i < n) {
            float z = pixel_depth<RAW>(raw, depth, mask, i, inv_scale, mask_mode);
            c += !(z <= 0.f || z > zmax);
        }
    }
    __shared__ int red[DP_BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// exclusive scan of nblocks counts by ONE workgroup of 1024 threads; total -> offsets[nblocks]
__global__ __launch_bounds__(1024)
void k_block_scan(const int* __restrict__ counts, int nblocks, int* __restrict__ offsets) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        int i = b0 + threadIdx.x;
        int v = i < nblocks ? counts[i] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        int carry = carry_s;
        if (i < nblocks) offsets[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[nblocks] = carry_s;
}

template <bool RAW>
__global__ __launch_bounds__(DP_BLOCK)
void k_emit(const uint16_t* __restrict__ raw, const float* __restrict__ depth, const uint8_t* __restrict__ mask,
            const uint8_t* __restrict__ bgr, int width, size_t n, float inv_scale, int mask_mode,
            float fx, float fy, float cx, float cy, float zmax,
            const int* __restrict__ offsets, int capacity, float* __restrict__ xyz, float* __restrict__ rgb) {
    const size_t base = (size_t)blockIdx.x * DP_PX_PER_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wcnt[DP_PX_PER_THREAD][DP_BLOCK / 64];
    float zs[DP_PX_PER_THREAD]; bool ok[DP_PX_PER_THREAD]; int rank[DP_PX_PER_THREAD];
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        size_t i = base + k * DP_BLOCK + threadIdx.x;
        float z = 0.f;
        if (i < n) z = pixel_depth<RAW>(raw, depth, mask, i, inv_scale, mask_mode);
        ok[k] = (i < n) && !(z <= 0.f || z > zmax);
        zs[k] = z;
        unsigned long long b = __ballot(ok[k]);
        rank[k] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[k][wave] = __popcll(b);
    }
    __syncthreads();
    int run = offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        int before = 0;
#pragma unroll
        for (int w = 0; w < DP_BLOCK / 64; ++w) before += (w < wave) ? wcnt[k][w] : 0;
        int slot = run + before + rank[k];
        if (ok[k] && slot < capacity) {
            size_t i = base + k * DP_BLOCK + threadIdx.x;
            int v = (int)(i / width), u = (int)(i - (size_t)v * width);
            float z = zs[k];
            float x = ((float)u - cx) * z / fx;   // pipeline.cpp:73
            float y = ((float)v - cy) * z / fy;   // pipeline.cpp:74
            xyz[3 * (size_t)slot] = x; xyz[3 * (size_t)slot + 1] = y; xyz[3 * (size_t)slot + 2] = z;
            if (rgb && bgr) {
                const uint8_t* p = bgr + i * 3;
                rgb[3 * (size_t)slot] = (float)p[2] / 255.0f;
                rgb[3 * (size_t)slot + 1] = (float)p[1] / 255.0f;
                rgb[3 * (size_t)slot + 2] = (float)p[0] / 255.0f;
            }
        }
        run += (wcnt[k][0] + wcnt[k][1]) + (wcnt[k][2] + wcnt[k][3]);
    }
}

// ------------------------------------------------------------------ bilateral depth filter
// Semantics of the reference's (never-called) filter, cuda/depth_processing.cu:62-155: window radius
// int(2 sigma_s + 0.5) clamped to 5; weight expf(d2 * (-0.5/sigma_s^2) + dr^2 * (-0.5/sigma_r^2)); zero depths are
// skipped and stay zero; taps outside the image count as zero; sums run row-major over the window (so results equal the
// CPU restatement's within expf's last bit).
// Design for gfx950 — no LDS tile, no barriers: one wave owns 64 consecutive pixels of ONE image row and streams the
// 2r + 1 rows of its window through registers.  A row of the window is two coalesced loads (the 64 pixels shifted by
// -r, and the 2r pixels that follow); the tap at offset dx is then a lane rotation of that register pair
// (ds_bpermute through the LDS crossbar, no LDS storage).  The spatial term of a tap is wave-uniform (a scalar
// operand); only the range term and the exponential are per lane.
constexpr int BF_MAXR = 5;

__global__ __launch_bounds__(256)
void k_bilateral_rows(const float* __restrict__ depth, float* __restrict__ filtered, int width, int height, int radius,
                      float spatial_scale, float range_scale) {
    const int lane = threadIdx.x & 63;
    const int segments = (width + 63) / 64;
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);             // wave-uniform: (row, 64-pixel segment)
    if (seg >= segments * height) return;
    const int row = seg / segments, col0 = (seg - row * segments) * 64;
    const int col = col0 + lane;
    const float mine = col < width ? depth[(size_t)row * width + col] : 0.f;
    float weight_sum = 0.f, value_sum = 0.f;
    for (int oy = -radius; oy <= radius; ++oy) {
        const int r = row + oy;
        const bool row_inside = r >= 0 && r < height;                 // wave-uniform
        // window of this row: pixels col0 - radius ... col0 + 63 + radius as (lo: first 64, hi: the rest)
        const int c_lo = col0 - radius + lane, c_hi = c_lo + 64;
        const float lo = (row_inside && c_lo >= 0 && c_lo < width) ? depth[(size_t)r * width + c_lo] : 0.f;
        const float hi = (row_inside && lane < 2 * radius && c_hi < width) ? depth[(size_t)r * width + c_hi] : 0.f;
        for (int ox = -radius; ox <= radius; ++ox) {
            const int k = lane + ox + radius;                          // position of this lane's tap inside the window
            const float a = __shfl(lo, k & 63, 64), b = __shfl(hi, k & 63, 64);
            const float tap = k < 64 ? a : b;
            if (tap <= 0.f) continue;                                  // missing depth (or outside the image)
            const float dr = tap - mine;
            const float w = expf((float)(ox * ox + oy * oy) * spatial_scale + dr * dr * range_scale);
            weight_sum += w;
            value_sum += w * tap;
        }
    }
    if (col < width) filtered[(size_t)row * width + col] = mine <= 0.f ? 0.f : (weight_sum > 0.f ? value_sum / weight_sum : mine);
}

int bilateral_filter_dev(tdv_ctx* ctx, const float* d_in, float* d_out, int w, int h, float sigma_spatial, float sigma_range) {
    if (!ctx || !d_in || !d_out || w < 0 || h < 0 || !(sigma_spatial > 0.f) || !(sigma_range > 0.f)) return TDV_ERR_BAD_ARG;
    if ((size_t)w * h == 0) return TDV_OK;
    const int radius = std::min(BF_MAXR, static_cast<int>(2.0f * sigma_spatial + 0.5f));   // depth_processing.cu:131-136
    const float spatial_scale = -0.5f / (sigma_spatial * sigma_spatial);
    const float range_scale = -0.5f / (sigma_range * sigma_range);
    const long long waves = (long long)((w + 63) / 64) * h;
    ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
    k_bilateral_rows<<<(unsigned)((waves + 3) / 4), 256, 0, ctx->stream>>>(d_in, d_out, w, h, radius, spatial_scale, range_scale);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

// ------------------------------------------------------------------ all instances of a frame in one pass
// B masks (stacked u8 images, or one label image with label = b + 1) of ONE depth/colour frame -> B clouds stored
// back to back, each in row-major pixel order.  Grid = (pixel blocks, instances); the depth image is re-read from
// L2 / Infinity Cache by every instance, the masks and the output stream through HBM once.
__global__ __launch_bounds__(DP_BLOCK)
void k_valid_count_batch(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ masks, size_t n, int stacked,
                         float inv_scale, int mask_mode, float zmax, int* __restrict__ block_counts) {
    const int b = blockIdx.y;
    const size_t frame = frame_of ? (size_t)frame_of[b] : 0;
    const uint16_t* __restrict__ raw = raw0 + frame * n;
    const uint8_t* __restrict__ mask = stacked ? masks + (size_t)b * n : masks;
    const int mode = stacked ? mask_mode : TDV_MASK_LABEL_BASE + b + 1;
    const size_t base = (size_t)blockIdx.x * DP_PX_PER_BLOCK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        size_t i = base + k * DP_BLOCK + threadIdx.x;
        if (i < n) {
            float z = scaled_depth(raw, mask, i, inv_scale, mode);
            c += !(z <= 0.f || z > zmax);
        }
    }
    __shared__ int red[DP_BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(DP_BLOCK)
void k_emit_batch(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ masks, const uint8_t* __restrict__ bgr0,
                  int width, size_t n, int stacked, float inv_scale, int mask_mode,
                  float fx, float fy, float cx, float cy, float zmax,
                  const int* __restrict__ offsets, float* __restrict__ xyz, float* __restrict__ rgb) {
    const int b = blockIdx.y;
    const size_t frame = frame_of ? (size_t)frame_of[b] : 0;
    const uint16_t* __restrict__ raw = raw0 + frame * n;
    const uint8_t* __restrict__ bgr = bgr0 ? bgr0 + frame * n * 3 : nullptr;
    const uint8_t* __restrict__ mask = stacked ? masks + (size_t)b * n : masks;
    const int mode = stacked ? mask_mode : TDV_MASK_LABEL_BASE + b + 1;
    const size_t base = (size_t)blockIdx.x * DP_PX_PER_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wcnt[DP_PX_PER_THREAD][DP_BLOCK / 64];
    float zs[DP_PX_PER_THREAD]; bool ok[DP_PX_PER_THREAD]; int rank[DP_PX_PER_THREAD];
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        size_t i = base + k * DP_BLOCK + threadIdx.x;
        float z = 0.f;
        if (i < n) z = scaled_depth(raw, mask, i, inv_scale, mode);
        ok[k] = (i < n) && !(z <= 0.f || z > zmax);
        zs[k] = z;
        unsigned long long bal = __ballot(ok[k]);
        rank[k] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[k][wave] = __popcll(bal);
    }
    __syncthreads();
    size_t run = (size_t)offsets[(size_t)b * gridDim.x + blockIdx.x];
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        int before = 0;
#pragma unroll
        for (int w = 0; w < DP_BLOCK / 64; ++w) before += (w < wave) ? wcnt[k][w] : 0;
        if (ok[k]) {
            const size_t slot = run + before + rank[k];
            const size_t i = base + k * DP_BLOCK + threadIdx.x;
            const int v = (int)(i / width), u = (int)(i - (size_t)v * width);
            const float z = zs[k];
            xyz[3 * slot] = ((float)u - cx) * z / fx;
            xyz[3 * slot + 1] = ((float)v - cy) * z / fy;
            xyz[3 * slot + 2] = z;
            if (rgb && bgr) {
                const uint8_t* p = bgr + i * 3;
                rgb[3 * slot] = (float)p[2] / 255.0f; rgb[3 * slot + 1] = (float)p[1] / 255.0f; rgb[3 * slot + 2] = (float)p[0] / 255.0f;
            }
        }
        run += (wcnt[k][0] + wcnt[k][1]) + (wcnt[k][2] + wcnt[k][3]);
    }
}

// ---- vectorised path (frame size a multiple of 16 pixels, 16-B aligned pointers): the masks cross HBM ONCE ----------
// Pass 1 (k_depth_bits): a wave owns 1,024 consecutive pixels (16 per lane).  It loads their depth once per frame, then
// walks DB_GROUP instances: one 16-B mask load per lane and instance -> 16 validity bits per lane, stored as a bitmap
// (2 B per lane, 1/8 of the mask's size) and counted per (instance, tile).  All DB_GROUP mask loads are issued before
// the first is used.  Pass 2 (k_depth_emit_bits) reads the bitmap instead of the masks, skips tiles without a valid
// pixel before touching anything else, recomputes depth only there, compacts through LDS and writes the cloud with
// coalesced stores.  HBM bytes for B stacked masks of one W x H frame with n valid pixels in total:
//   B W H (masks, once) + 2 B W H / 8 (bitmap out and in) + 12 n (points)        [round 1: 2 B W H + 12 n]
constexpr int DB_PX = 16;
constexpr int DB_TILE = 64 * DB_PX;       // pixels per wave
#ifndef DB_GROUP_VALUE
#define DB_GROUP_VALUE 8
#endif
constexpr int DB_GROUP = DB_GROUP_VALUE;  // instances per wave in pass 1 (8: the LDS transpose of the bitmap words assumes it)
static_assert(DB_GROUP == 8, "k_depth_bits stores 8 instances x 64 words as 64 x 16 B");

__device__ __forceinline__ unsigned keep_bits16(const uint4 mv, int mode) {
    const unsigned mw[4] = {mv.x, mv.y, mv.z, mv.w};
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < DB_PX; ++k) keep |= mask_keeps((uint8_t)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu), mode) ? (1u << k) : 0u;
    return keep;
}
__device__ __forceinline__ void depth16(const uint16_t* __restrict__ raw, size_t i0, float inv_scale, float (&z)[DB_PX]) {
    const uint4 r0 = *reinterpret_cast<const uint4*>(raw + i0);
    const uint4 r1 = *reinterpret_cast<const uint4*>(raw + i0 + 8);
    const unsigned rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int k = 0; k < DB_PX; ++k) z[k] = (float)((rw[k >> 1] >> (16 * (k & 1))) & 0xffffu) * inv_scale;   // pipeline.cpp:47
}
__device__ __forceinline__ unsigned depth_valid16(const float (&z)[DB_PX], float zmax) {
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < DB_PX; ++k) v |= (!(z[k] <= 0.f || z[k] > zmax)) ? (1u << k) : 0u;                    // pipeline.cpp:71
    return v;
}

__global__ __launch_bounds__(DP_BLOCK)
void k_depth_bits(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ masks, size_t n, int n_inst,
                  int stacked, float inv_scale, int mask_mode, float zmax, int ntiles, uint16_t* __restrict__ bits, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (DP_BLOCK / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const size_t i0 = (size_t)tile * DB_TILE + (size_t)lane * DB_PX;
    const bool inside = i0 < n;                       // n is a multiple of 16: a lane is in or out as a whole
    const int b0 = blockIdx.y * DB_GROUP, b1 = min(n_inst, b0 + DB_GROUP);
    uint4 mv[DB_GROUP];
#pragma unroll
    for (int u = 0; u < DB_GROUP; ++u) {              // all mask loads of the group in flight at once
        const int b = b0 + u;
        mv[u] = make_uint4(0u, 0u, 0u, 0u);
        if (inside && b < b1 && (stacked || u == 0)) mv[u] = *reinterpret_cast<const uint4*>((stacked ? masks + (size_t)b * n : masks) + i0);
    }
    // The 16 validity bits of a lane are 2 bytes: stored lane by lane they would leave as 2-byte stores, the slowest shape
    // there is (per byte an order of magnitude above 16-B stores).  The wave therefore transposes its DB_GROUP x 64 words
    // through 1 KB of LDS and every lane stores 16 B: eight lanes cover one instance's 128-B row segment.
    __shared__ __attribute__((aligned(16))) uint16_t xpose[DP_BLOCK / 64][DB_GROUP][64];
    const int wave = threadIdx.x >> 6;
    int cur_frame = -1;
    unsigned dvalid = 0;
#pragma unroll
    for (int u = 0; u < DB_GROUP; ++u) {
        const int b = b0 + u;
        unsigned v = 0u;
        if (b < b1) {                                      // wave-uniform
            const int frame = frame_of ? frame_of[b] : 0;
            if (frame != cur_frame) {                      // wave-uniform
                cur_frame = frame;
                dvalid = 0;
                if (inside) { float z[DB_PX]; depth16(raw0 + (size_t)frame * n, i0, inv_scale, z); dvalid = depth_valid16(z, zmax); }
            }
            const int mode = stacked ? mask_mode : TDV_MASK_LABEL_BASE + b + 1;
            v = inside ? (keep_bits16(stacked ? mv[u] : mv[0], mode) & dvalid) : 0u;
            const int c = wave_sum_i32(__popc(v));
            if (lane == 0) counts[(size_t)b * ntiles + tile] = c;
        }
        xpose[wave][u][lane] = (uint16_t)v;
    }
    __builtin_amdgcn_wave_barrier();                       // a wave reads back only what it wrote itself
    const size_t words = n / DB_PX;                        // bitmap words per instance
    const int u2 = lane >> 3, seg = lane & 7;              // this lane stores words [seg*8, seg*8+8) of instance b0 + u2
    const size_t w0 = (size_t)tile * 64 + (size_t)seg * 8;
    if (b0 + u2 < b1 && w0 < words) {
        uint16_t* dst = bits + (size_t)(b0 + u2) * words + w0;
        if (w0 + 8 <= words && (words % 8) == 0) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&xpose[wave][u2][seg * 8]);
        else for (int k = 0; k < 8 && w0 + k < words; ++k) dst[k] = xpose[wave][u2][seg * 8 + k];
    }
}

#ifndef DE_BLOCK_VALUE
#define DE_BLOCK_VALUE 256
#endif
constexpr int DE_BLOCK = DE_BLOCK_VALUE;
// Pass 2.  A wave owns the same 1,024 pixels as in pass 1 but walks them in 16 rounds of 64 CONSECUTIVE pixels (lane =
// pixel): the valid pixels of a round are consecutive points of the output, so a lane's slot is the round's base plus the
// number of valid lanes below it (one s_bcnt / v_mbcnt on the round's 64-bit mask) — no per-lane prefix sums, no
// compaction buffer, stores of neighbouring lanes are neighbours in memory.  (The first version let every lane own 16
// consecutive pixels and compacted through LDS: its 48 ds_write_b32 per lane hit 4 of the 64 banks — lane stride 48 dwords
// — and the pass ran at 3.8 TB/s.)  The round masks come from the bitmap words the lanes hold (4 v_readlane per round);
// the tile's raw depths are staged once in 2 KB of LDS so that a round reads them lane = pixel.
__global__ __launch_bounds__(DE_BLOCK)
void k_depth_emit_bits(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ bgr0, const uint16_t* __restrict__ bits,
                       int width, size_t n, float inv_scale, float fx, float fy, float cx, float cy, int ntiles,
                       const int* __restrict__ offsets, float* __restrict__ xyz, float* __restrict__ rgb) {
    __shared__ __attribute__((aligned(16))) uint16_t zraw[DE_BLOCK / 64][DB_TILE];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * (DE_BLOCK / 64) + wave;
    if (tile >= ntiles) return;
    const size_t t0 = (size_t)tile * DB_TILE;                         // first pixel of the tile
    const size_t i0 = t0 + (size_t)lane * DB_PX;
    const unsigned word = i0 < n ? (unsigned)bits[(size_t)b * (n / DB_PX) + i0 / DB_PX] : 0u;   // validity of pixels i0 .. i0 + 15
    if (!__any(word != 0u)) return;                                   // most tiles of an instance: nothing else is read
    const size_t frame = frame_of ? (size_t)frame_of[b] : 0;
    const uint16_t* __restrict__ raw = raw0 + frame * n;
    if (i0 < n) {                                                     // n is a multiple of 16: 32 B per lane, whole or not at all
        *reinterpret_cast<uint4*>(&zraw[wave][lane * DB_PX]) = *reinterpret_cast<const uint4*>(raw + i0);
        *reinterpret_cast<uint4*>(&zraw[wave][lane * DB_PX + 8]) = *reinterpret_cast<const uint4*>(raw + i0 + 8);
    }
    __builtin_amdgcn_wave_barrier();                                  // a wave reads back only what it staged itself
    size_t slot0 = (size_t)offsets[(size_t)b * ntiles + tile];        // output slot of the tile's first valid pixel
    const uint8_t* __restrict__ bgr = (rgb && bgr0) ? bgr0 + frame * n * 3 : nullptr;
#pragma unroll 1
    for (int j = 0; j < DB_PX; ++j) {                                 // round j: pixels t0 + 64 j .. + 63
        const unsigned long long m = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)word, 4 * j)
                                   | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)word, 4 * j + 1) << 16)
                                   | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)word, 4 * j + 2) << 32)
                                   | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)word, 4 * j + 3) << 48);
        if (!m) continue;                                             // wave-uniform
        if ((m >> lane) & 1ull) {
            const size_t i = t0 + 64 * (size_t)j + lane;
            const float z = (float)zraw[wave][64 * j + lane] * inv_scale;   // pipeline.cpp:47
            const int v = (int)(i / width), u = (int)(i - (size_t)v * width);
            const size_t slot = slot0 + __popcll(m & ((1ull << lane) - 1ull));
            xyz[3 * slot] = ((float)u - cx) * z / fx;                  // pipeline.cpp:73
            xyz[3 * slot + 1] = ((float)v - cy) * z / fy;              // pipeline.cpp:74
            xyz[3 * slot + 2] = z;
            if (bgr) {
                const uint8_t* p = bgr + i * 3;
                rgb[3 * slot] = (float)p[2] / 255.0f; rgb[3 * slot + 1] = (float)p[1] / 255.0f; rgb[3 * slot + 2] = (float)p[0] / 255.0f;
            }
        }
        slot0 += __popcll(m);
    }
}

__global__ void k_gather_instance_offsets(const int* __restrict__ offsets, const int* __restrict__ total, int blocks, int n_inst, int* __restrict__ out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_inst) out[b] = offsets[(size_t)b * blocks];
    if (b == n_inst) out[b] = *total;
}

static bool batch_vectorisable(const uint16_t* d_raw, const uint8_t* d_masks, size_t n) {
    return n % 16 == 0 && ((uintptr_t)d_raw % 16 == 0) && ((uintptr_t)d_masks % 16 == 0);   // stacked masks and frames sit at multiples of n
}

// Device copy of the instance -> frame map (nullptr when every instance reads frame 0).
int frame_map_dev(tdv_ctx* ctx, int n_inst, int n_frames, const int* h_frame_of, const int** d_frame_of) {
    *d_frame_of = nullptr;
    if (n_frames <= 1 || n_inst <= 0) return TDV_OK;
    std::vector<int> f((size_t)n_inst);
    for (int b = 0; b < n_inst; ++b) {
        f[b] = h_frame_of ? h_frame_of[b] : (int)((long long)b * n_frames / n_inst);
        if (f[b] < 0 || f[b] >= n_frames) return TDV_ERR_BAD_ARG;
    }
    int* d;
    TDV_TRY(ws_alloc(ctx, (size_t)n_inst, &d));
    TDV_HIP(ctx, hipMemcpyAsync(d, f.data(), (size_t)n_inst * 4, hipMemcpyHostToDevice, ctx->stream));
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));   // f is a host temporary
    *d_frame_of = d;
    return TDV_OK;
}

// pass 1 (bitmap + counts + scan): returns the per-instance start offsets (host, n_inst + 1 entries) and keeps the device
// scan and the bitmap (ctx workspace) for pass 2
int depth_to_cloud_batch_count(tdv_ctx* ctx, const uint16_t* d_raw, const int* d_frame_of, const uint8_t* d_masks, int n_inst, int stacked, int w, int h,
                               float scale, int mask_mode, float zmax, int** d_offsets_out, int* h_offsets) {
    const size_t n = (size_t)w * h;
    const float inv_scale = (float)(1.0 / (double)scale);
    const bool vec = batch_vectorisable(d_raw, d_masks, n);
    const int blocks = vec ? (int)((n + DB_TILE - 1) / DB_TILE) : (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);   // tiles resp. workgroups per instance
    int *counts, *offsets, *d_total, *d_inst;
    TDV_TRY(ws_alloc(ctx, (size_t)blocks * n_inst, &counts));
    TDV_TRY(ws_alloc(ctx, (size_t)blocks * n_inst, &offsets));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(ws_alloc(ctx, (size_t)n_inst + 1, &d_inst));
    ctx->depth_bits = nullptr;
    if (vec) TDV_TRY(ws_alloc(ctx, (size_t)n_inst * (n / DB_PX), &ctx->depth_bits));
    hipStream_t s = ctx->stream;
    {
        ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
        if (vec) k_depth_bits<<<dim3((blocks + DP_BLOCK / 64 - 1) / (DP_BLOCK / 64), (n_inst + DB_GROUP - 1) / DB_GROUP), DP_BLOCK, 0, s>>>(
                     d_raw, d_frame_of, d_masks, n, n_inst, stacked, inv_scale, mask_mode, zmax, blocks, ctx->depth_bits, counts);
        else k_valid_count_batch<<<dim3(blocks, n_inst), DP_BLOCK, 0, s>>>(d_raw, d_frame_of, d_masks, n, stacked, inv_scale, mask_mode, zmax, counts);
    }
    TDV_CHECK_LAUNCH(ctx);
    {
        ScopedTimer tm(ctx, TDV_TIMER_DEPTH);   // the scan belongs to the operator's kernel time
        TDV_TRY(exclusive_scan_dev(ctx, counts, blocks * n_inst, offsets, d_total));
        k_gather_instance_offsets<<<(n_inst + 256) / 256, 256, 0, s>>>(offsets, d_total, blocks, n_inst, d_inst);
    }
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, ((size_t)n_inst + 1) * 4));
    TDV_HIP(ctx, hipMemcpyAsync(ctx->pin, d_inst, ((size_t)n_inst + 1) * 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    std::memcpy(h_offsets, ctx->pin, ((size_t)n_inst + 1) * 4);
    *d_offsets_out = offsets;
    return TDV_OK;
}

// pass 2 (emit) into buffers sized from pass 1
int depth_to_cloud_batch_emit(tdv_ctx* ctx, const uint16_t* d_raw, const int* d_frame_of, const uint8_t* d_masks, const uint8_t* d_bgr, int n_inst, int stacked,
                              int w, int h, float scale, int mask_mode, float fx, float fy, float cx, float cy, float zmax,
                              const int* d_offsets, float* d_xyz, float* d_rgb) {
    const size_t n = (size_t)w * h;
    const float inv_scale = (float)(1.0 / (double)scale);
    const bool vec = batch_vectorisable(d_raw, d_masks, n);
    ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
    if (vec) {
        if (!ctx->depth_bits) return TDV_ERR_INTERNAL;
        const int ntiles = (int)((n + DB_TILE - 1) / DB_TILE);
        k_depth_emit_bits<<<dim3((ntiles + DE_BLOCK / 64 - 1) / (DE_BLOCK / 64), n_inst), DE_BLOCK, 0, ctx->stream>>>(
            d_raw, d_frame_of, d_bgr, ctx->depth_bits, w, n, inv_scale, fx, fy, cx, cy, ntiles, d_offsets, d_xyz, d_rgb);
    } else {
        const int blocks = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
        k_emit_batch<<<dim3(blocks, n_inst), DP_BLOCK, 0, ctx->stream>>>(d_raw, d_frame_of, d_masks, d_bgr, w, n, stacked, inv_scale, mask_mode,
                                                                          fx, fy, cx, cy, zmax, d_offsets, d_xyz, d_rgb);
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

int depth_preprocess_dev(tdv_ctx* ctx, const uint16_t* d_raw, const uint8_t* d_mask, int w, int h, float scale,
                         int mask_mode, float* d_out) {
    if (!ctx || !d_raw || !d_out || w < 0 || h < 0) return TDV_ERR_BAD_ARG;
    const size_t n = (size_t)w * h;
    if (n == 0) return TDV_OK;
    const float inv_scale = (float)(1.0 / (double)scale);  // cv::Mat::convertTo(CV_32F, 1.0/scale)
    const int blocks = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
    ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
    k_depth_preprocess<<<blocks, DP_BLOCK, 0, ctx->stream>>>(d_raw, d_mask, n, inv_scale, mask_mode, d_out);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

int depth_to_cloud_dev(tdv_ctx* ctx, const uint16_t* d_raw, const float* d_depth, const uint8_t* d_mask,
                       const uint8_t* d_bgr, int w, int h, float scale, int mask_mode,
                       float fx, float fy, float cx, float cy, float zmax,
                       float* d_xyz, float* d_rgb, int capacity, int* n_out) {
    if (!ctx || (!d_raw && !d_depth) || !n_out || w < 0 || h < 0 || capacity < 0) return TDV_ERR_BAD_ARG;
    *n_out = 0;
    const size_t n = (size_t)w * h;
    if (n == 0) return TDV_OK;
    const float inv_scale = (float)(1.0 / (double)scale);
    const int blocks = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
    int *counts, *offsets;
    TDV_TRY(ws_alloc(ctx, (size_t)blocks, &counts));
    TDV_TRY(ws_alloc(ctx, (size_t)blocks + 1, &offsets));
    TDV_TRY(pin_reserve(ctx, 64));
    hipStream_t s = ctx->stream;
    {
        ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
        if (d_raw) k_valid_count<true><<<blocks, DP_BLOCK, 0, s>>>(d_raw, nullptr, d_mask, n, inv_scale, mask_mode, zmax, counts);
        else k_valid_count<false><<<blocks, DP_BLOCK, 0, s>>>(nullptr, d_depth, nullptr, n, 0.f, 0, zmax, counts);
        k_block_scan<<<1, 1024, 0, s>>>(counts, blocks, offsets);
        if (d_xyz && capacity > 0) {
            if (d_raw) k_emit<true><<<blocks, DP_BLOCK, 0, s>>>(d_raw, nullptr, d_mask, d_bgr, w, n, inv_scale, mask_mode, fx, fy, cx, cy, zmax, offsets, capacity, d_xyz, d_rgb);
            else k_emit<false><<<blocks, DP_BLOCK, 0, s>>>(nullptr, d_depth, nullptr, d_bgr, w, n, 0.f, 0, fx, fy, cx, cy, zmax, offsets, capacity, d_xyz, d_rgb);
        }
    }
    TDV_CHECK_LAUNCH(ctx);
    int* h_total = reinterpret_cast<int*>(ctx->pin);
    TDV_HIP(ctx, hipMemcpyAsync(h_total, offsets + blocks, 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    *n_out = *h_total;
    if (*h_total > capacity) return TDV_ERR_BAD_ARG;  // caller's buffers too small; *n_out = needed
    return TDV_OK;
}

}  // namespace tdv
