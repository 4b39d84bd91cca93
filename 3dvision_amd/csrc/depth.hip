// Depth scale + mask and depth -> point-cloud unprojection on gfx950 (HBM-bound scans).
//
// Replaces GPUDepth::preprocess (/root/reference/src/gpu_impl.cpp:28-66, kernel
// cuda/depth_processing.cu:10-30) and GPUPointCloud::generate (gpu_impl.cpp:69-128, kernel
// cuda/pointcloud.cu:11-51); results follow the CPU branches src/pipeline.cpp:46-54 and :61-84:
//   depth  = float(raw) * float(1.0/scale), zero where the mask rejects the pixel
//   keep 0 < z <= zmax ; x = (u - cx) * z / fx ; y = (v - cy) * z / fy ; colour = BGR->RGB / 255
// and — unlike the reference's CUDA kernel, whose global atomicAdd makes the order
// nondeterministic — points are emitted in the CPU's row-major scan order.  One frame is ONE launch (round 4, k_depth_cloud_chain):
// a workgroup takes the next tile of 1,024 pixels (a ticket), counts its valid pixels, learns how many precede it from the tiles before it
// (a chained scan: every tile publishes its count, then its inclusive prefix, in one 64-bit word; a wave looks back over 64 predecessors
// at a time) and writes its points.  Rounds 1-3 ran three launches (count per block, scan of the blocks, emit), whose two launch gaps
// were half of the 13 us a frame took; they are kept in the study build (TDV_DEPTH_THREE_PASS=1).
// Algorithmic HBM bytes per frame (fused u16+mask+bgr path): 2 + 1 in (+ 3 per valid pixel) + 12n (24n) out.
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

#ifdef TDV_STUDY
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
void k_block_scan(const int* __restrict__ counts, int nblocks, int* __restrict__ offsets, int* __restrict__ host_total = nullptr) {
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
    if (threadIdx.x == 0) {
        offsets[nblocks] = carry_s;
        if (host_total) { *host_total = carry_s; __threadfence_system(); }   // straight into pinned host memory: a 4-byte D2H copy is a 9-us blit kernel
    }
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

#endif  // TDV_STUDY

// One frame in one launch: tile = ticket, chained scan over the tiles' counts, emit.
// status[tile] = epoch << 34 | state << 32 | value; state 1: value = the tile's own count, 2: value = the inclusive prefix up to and with
// the tile.  The epoch (one per call) makes words of earlier calls read as "not there yet" - the array is never cleared.  Tickets are
// handed out in the order workgroups START, so every tile a workgroup waits for is already running and publishes its count without
// waiting for anything: the chain cannot deadlock.  A wave that waited CHAIN_SPIN_LIMIT rounds gives up and reports it (fail word) instead of
// hanging the GPU - that would be a bug here, not a state of the machine.
// Tiles are 4,096 pixels (1,024 threads): a 1280 x 720 frame is 225 tiles - one round of workgroups, all running at once - and the wave that
// looks back reads 256 predecessors per round trip (four loads per lane in flight), so such a frame's whole scan is one publish and one
// read.  (First version: 1,024-pixel tiles and 64 predecessors per round; the last of 900 tiles walked 15 dependent device-scope round
// trips, 22.6 us per frame against the three launches' 13.1.)
constexpr int CHAIN_SPIN_LIMIT = 1 << 22;
constexpr int CH_BLOCK = 1024, CH_WAVES = CH_BLOCK / 64, CH_TILE = CH_BLOCK * DP_PX_PER_THREAD, CH_LOOK = 4;
template <bool RAW, bool VEC>
__global__ __launch_bounds__(CH_BLOCK)
void k_depth_cloud_chain(const uint16_t* __restrict__ raw, const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                         const uint8_t* __restrict__ bgr, int width, size_t n, float inv_scale, int mask_mode,
                         float fx, float fy, float cx, float cy, float zmax,
                         unsigned* __restrict__ ticket, unsigned ticket_base, unsigned long long* __restrict__ status, unsigned epoch, int ntiles,
                         int capacity, float* __restrict__ xyz, float* __restrict__ rgb, int* __restrict__ host_total /* pinned: [0] total, [1] fail */) {
    __shared__ int s_tile, s_excl;
    __shared__ int wcnt[CH_WAVES];
    if (threadIdx.x == 0) s_tile = (int)(atomicAdd(ticket, 1u) - ticket_base);
    __syncthreads();
    const int tile = s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // four CONSECUTIVE pixels per lane: one 8-byte depth load, one 4-byte mask load
    const size_t i0 = (size_t)tile * CH_TILE + (size_t)threadIdx.x * DP_PX_PER_THREAD;
    float zs[DP_PX_PER_THREAD]; bool ok[DP_PX_PER_THREAD];
    if (VEC && i0 + DP_PX_PER_THREAD <= n) {
        if (RAW) {
            const ushort4 r = *reinterpret_cast<const ushort4*>(raw + i0);
            zs[0] = (float)r.x * inv_scale; zs[1] = (float)r.y * inv_scale; zs[2] = (float)r.z * inv_scale; zs[3] = (float)r.w * inv_scale;
            if (mask) {
                const uchar4 m = *reinterpret_cast<const uchar4*>(mask + i0);
                if (!mask_keeps(m.x, mask_mode)) zs[0] = 0.f;
                if (!mask_keeps(m.y, mask_mode)) zs[1] = 0.f;
                if (!mask_keeps(m.z, mask_mode)) zs[2] = 0.f;
                if (!mask_keeps(m.w, mask_mode)) zs[3] = 0.f;
            }
        } else {
            const float4 d = *reinterpret_cast<const float4*>(depth + i0);
            zs[0] = d.x; zs[1] = d.y; zs[2] = d.z; zs[3] = d.w;
        }
#pragma unroll
        for (int k = 0; k < DP_PX_PER_THREAD; ++k) ok[k] = !(zs[k] <= 0.f || zs[k] > zmax);
    } else {
#pragma unroll
        for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
            const size_t i = i0 + k;
            zs[k] = i < n ? pixel_depth<RAW>(raw, depth, mask, i, inv_scale, mask_mode) : 0.f;
            ok[k] = i < n && !(zs[k] <= 0.f || zs[k] > zmax);
        }
    }
    // rank of pixel (lane, k) inside its wave: pixels of lower lanes, then the lane's own earlier pixels
    int before = 0, wave_total = 0, own = 0;
    int rank[DP_PX_PER_THREAD];
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        const unsigned long long b = __ballot(ok[k]);
        before += __popcll(b & ((1ull << lane) - 1ull));
        wave_total += __popcll(b);
        rank[k] = own;
        own += ok[k] ? 1 : 0;
    }
    if (lane == 0) wcnt[wave] = wave_total;
    __syncthreads();
    if (wave == 0) {
        int total = lane < CH_WAVES ? wcnt[lane] : 0;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
        total = __shfl(total, 0, 64);
        const unsigned long long tag = (unsigned long long)epoch << 34;
        if (lane == 0) __hip_atomic_store(&status[tile], tag | ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int excl = 0; bool failed = false;
        for (int look = tile - 1; look >= 0; look -= 64 * CH_LOOK) {
            unsigned long long sv[CH_LOOK];      // predecessor look - (64 j + lane): near to far
            int spins = 0;
            while (true) {
                bool ready = true;
#pragma unroll
                for (int j = 0; j < CH_LOOK; ++j) {
                    const int idx = look - (64 * j + lane);
                    sv[j] = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | (2ull << 32));   // before the first tile: an inclusive prefix of 0
                }
#pragma unroll
                for (int j = 0; j < CH_LOOK; ++j) ready = ready && (unsigned)(sv[j] >> 34) == epoch && ((sv[j] >> 32) & 3ull) != 0ull;
                if (__all(ready)) break;
                if (++spins > CHAIN_SPIN_LIMIT) { failed = true; break; }      // (wave-uniform: every lane counts the same rounds)
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed) break;
            int v = 0; bool cut = false;                                         // nothing beyond the nearest inclusive prefix
#pragma unroll
            for (int j = 0; j < CH_LOOK; ++j) {
                const unsigned long long m = __ballot(((sv[j] >> 32) & 3ull) == 2ull);
                if (!cut && (!m || lane <= __ffsll((long long)m) - 1)) v += (int)(unsigned)sv[j];
                cut = cut || m != 0ull;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            excl += v;
            if (cut) break;
        }
        if (lane == 0) {
            __hip_atomic_store(&status[tile], tag | (2ull << 32) | (unsigned)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
            if (failed) { host_total[1] = 1; __threadfence_system(); }
            if (tile == ntiles - 1) { host_total[0] = excl + total; __threadfence_system(); }   // straight into pinned host memory: a 4-byte D2H copy is a 9-us blit kernel
        }
    }
    __syncthreads();
    if (!xyz) return;
    int slot0 = s_excl + before;
#pragma unroll
    for (int w = 0; w < CH_WAVES; ++w) slot0 += (w < wave) ? wcnt[w] : 0;
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        const int slot = slot0 + rank[k];
        if (ok[k] && slot < capacity) {
            const size_t i = i0 + k;
            const int v = (int)(i / width), u = (int)(i - (size_t)v * width);
            const float z = zs[k];
            const float x = ((float)u - cx) * z / fx;   // pipeline.cpp:73
            const float y = ((float)v - cy) * z / fy;   // pipeline.cpp:74
            xyz[3 * (size_t)slot] = x; xyz[3 * (size_t)slot + 1] = y; xyz[3 * (size_t)slot + 2] = z;
            if (rgb && bgr) {
                const uint8_t* p = bgr + i * 3;
                rgb[3 * (size_t)slot] = (float)p[2] / 255.0f;
                rgb[3 * (size_t)slot + 1] = (float)p[1] / 255.0f;
                rgb[3 * (size_t)slot + 2] = (float)p[0] / 255.0f;
            }
        }
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
// `stacked` carries the mask layout: 1 = one u8 mask per instance (mask_mode applies), 0 = ONE u8 label image (instance b
// keeps the pixels equal to b + 1), 2 = ONE u16 label image (the same rule, for more than 255 instances).
__device__ __forceinline__ bool batch_mask_keeps(const uint8_t* __restrict__ masks, size_t n, int b, int layout, size_t i, int mask_mode) {
    if (layout == 1) return mask_keeps(masks[(size_t)b * n + i], mask_mode);
    if (layout == 0) return (int)masks[i] == b + 1;
    return (int)reinterpret_cast<const uint16_t*>(masks)[i] == b + 1;
}
__device__ __forceinline__ float batch_pixel_depth(const uint16_t* __restrict__ raw, const uint8_t* __restrict__ masks, size_t n, int b, int layout,
                                                   size_t i, float inv_scale, int mask_mode) {
    float v = (float)raw[i] * inv_scale;
    if (!batch_mask_keeps(masks, n, b, layout, i, mask_mode)) v = 0.f;
    return v;
}

__global__ __launch_bounds__(DP_BLOCK)
void k_valid_count_batch(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ masks, size_t n, int stacked,
                         float inv_scale, int mask_mode, float zmax, int* __restrict__ block_counts) {
    const int b = blockIdx.y;
    const size_t frame = frame_of ? (size_t)frame_of[b] : 0;
    const uint16_t* __restrict__ raw = raw0 + frame * n;
    const size_t base = (size_t)blockIdx.x * DP_PX_PER_BLOCK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        size_t i = base + k * DP_BLOCK + threadIdx.x;
        if (i < n) {
            float z = batch_pixel_depth(raw, masks, n, b, stacked, i, inv_scale, mask_mode);
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
    const size_t base = (size_t)blockIdx.x * DP_PX_PER_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wcnt[DP_PX_PER_THREAD][DP_BLOCK / 64];
    float zs[DP_PX_PER_THREAD]; bool ok[DP_PX_PER_THREAD]; int rank[DP_PX_PER_THREAD];
#pragma unroll
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        size_t i = base + k * DP_BLOCK + threadIdx.x;
        float z = 0.f;
        if (i < n) z = batch_pixel_depth(raw, masks, n, b, stacked, i, inv_scale, mask_mode);
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

// The mask rule on four mask bytes at once (one dword, no per-byte extraction: the byte-by-byte form made pass 1
// VALU-bound at twice its memory time).  y = w, or w ^ label in every byte for the label rule; low 7 bits + `add` carry
// into bit 7 exactly when the byte exceeds the threshold (thresholds < 128; bytes >= 128 have bit 7 set themselves);
// the label rule is the exact zero-byte test of y, i.e. the complement.  Then the four bit-7 flags move to bits 0..3.
struct MaskRule { unsigned xor4, add4, inv, and4; };
__device__ __forceinline__ MaskRule mask_rule(int mode) {
    if (mode == TDV_MASK_THRESHOLD10) return {0u, (127u - 10u) * 0x01010101u, 0u, 0x80808080u};   // m > 10   (pipeline.cpp:51)
    if (mode == TDV_MASK_NONZERO) return {0u, 127u * 0x01010101u, 0u, 0x80808080u};             // m != 0   (depth_processing.cu:22)
    const int label = mode - TDV_MASK_LABEL_BASE;                                        // m == label
    if (label < 0 || label > 255) return {0u, 0u, 0u, 0u};                              // no byte can equal it: keeps nothing
    return {(unsigned)label * 0x01010101u, 127u * 0x01010101u, 0xffffffffu, 0x80808080u};
}
__device__ __forceinline__ unsigned keep_nibble(unsigned w, const MaskRule r) {
    const unsigned y = w ^ r.xor4;
    const unsigned f = (((((y & 0x7f7f7f7fu) + r.add4) | y) ^ r.inv) & r.and4) >> 7;   // flags at bits 0, 8, 16, 24
    return (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xfu;
}
__device__ __forceinline__ unsigned keep_bits16(const uint4 mv, const MaskRule r) {
    return keep_nibble(mv.x, r) | (keep_nibble(mv.y, r) << 4) | (keep_nibble(mv.z, r) << 8) | (keep_nibble(mv.w, r) << 12);
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

// `0 < z <= zmax` with z = (float)raw * inv_scale (pipeline.cpp:47,71) is a test on the raw value itself: the product is
// monotone in raw, so the raw values that pass form one range [raw_lo, raw_hi], found on the host with the same float
// expression (depth_valid_range).  Sixteen pixels: the sign of (raw - lo) | (hi - raw) is shifted into the result, last
// pixel first.
__device__ __forceinline__ unsigned range_bits16(const uint4 r0, const uint4 r1, int lo, int hi) {
    const unsigned rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    unsigned inv = 0u;
#pragma unroll
    for (int k = DB_PX - 1; k >= 0; --k) {
        const int a = (int)((rw[k >> 1] >> (16 * (k & 1))) & 0xffffu);
        inv = __builtin_amdgcn_alignbit(inv, (unsigned)((a - lo) | (hi - a)), 31);   // inv = inv << 1 | sign
    }
    return ~inv & 0xffffu;
}

// STACKED: one u8 mask per instance (else one label image, label = instance + 1).  PER_FRAME: instances read their own
// depth frame (frame_of), else all read frame 0 and its validity bits are computed once per wave.  The control flow is
// uniform: a ragged last group repeats its last instance and only skips the stores.
template <bool STACKED, bool PER_FRAME>
__global__ __launch_bounds__(DP_BLOCK)
void k_depth_bits(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ masks, size_t n, int n_inst,
                  int raw_lo, int raw_hi, int mask_mode, int ntiles, uint16_t* __restrict__ bits, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (DP_BLOCK / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const size_t i0 = (size_t)tile * DB_TILE + (size_t)lane * DB_PX;
    const bool inside = i0 < n;                       // n is a multiple of 16: a lane is in or out as a whole
    const size_t il = inside ? i0 : 0;                // lanes past the end load pixel 0 and drop the result
    const int b0 = blockIdx.y * DB_GROUP, nb = min(DB_GROUP, n_inst - b0);
    uint4 mv[STACKED ? DB_GROUP : 1];
    if (STACKED) {
#pragma unroll
        for (int u = 0; u < DB_GROUP; ++u)            // all mask loads of the group in flight at once
            mv[u] = *reinterpret_cast<const uint4*>(masks + (size_t)(b0 + min(u, nb - 1)) * n + il);
    } else {
        mv[0] = *reinterpret_cast<const uint4*>(masks + il);
    }
    unsigned dvalid = 0u;
    if (!PER_FRAME) {
        const uint4 r0 = *reinterpret_cast<const uint4*>(raw0 + il), r1 = *reinterpret_cast<const uint4*>(raw0 + il + 8);
        dvalid = inside ? range_bits16(r0, r1, raw_lo, raw_hi) : 0u;
    }
    // The 16 validity bits of a lane are 2 bytes: stored lane by lane they would leave as 2-byte stores, the slowest shape
    // there is (per byte an order of magnitude above 16-B stores).  The wave therefore transposes its DB_GROUP x 64 words
    // through 1 KB of LDS and every lane stores 16 B: eight lanes cover one instance's 128-B row segment.
    __shared__ __attribute__((aligned(16))) uint16_t xpose[DP_BLOCK / 64][DB_GROUP][64];
    const int wave = threadIdx.x >> 6;
    const MaskRule stacked_rule = mask_rule(mask_mode);
#pragma unroll
    for (int u = 0; u < DB_GROUP; ++u) {
        const int b = b0 + min(u, nb - 1);
        if (PER_FRAME) {
            const uint16_t* __restrict__ raw = raw0 + (size_t)frame_of[b] * n + il;
            const uint4 r0 = *reinterpret_cast<const uint4*>(raw), r1 = *reinterpret_cast<const uint4*>(raw + 8);
            dvalid = inside ? range_bits16(r0, r1, raw_lo, raw_hi) : 0u;
        }
        // An instance's mask is zero over most of the frame, and a zero byte passes none of the rules (labels start at 1):
        // a wave whose 1 KB of mask is all zero - most waves - skips the bit arithmetic, which otherwise costs this
        // pass more than its memory traffic.
        const uint4 m = mv[STACKED ? u : 0];
        unsigned v = 0u; int c = 0;
        if (__any((m.x | m.y | m.z | m.w) != 0u)) {       // wave-uniform
            const MaskRule r = STACKED ? stacked_rule : mask_rule(TDV_MASK_LABEL_BASE + b + 1);
            v = keep_bits16(m, r) & dvalid;
            c = wave_sum_i32(__popc(v));
        }
        if (lane == 0 && u < nb) counts[(size_t)b * ntiles + tile] = c;
        xpose[wave][u][lane] = (uint16_t)v;
    }
    __builtin_amdgcn_wave_barrier();                       // a wave reads back only what it wrote itself
    const size_t words = n / DB_PX;                        // bitmap words per instance
    const int u2 = lane >> 3, seg = lane & 7;              // this lane stores words [seg*8, seg*8+8) of instance b0 + u2
    const size_t w0 = (size_t)tile * 64 + (size_t)seg * 8;
    if (u2 < nb && w0 < words) {
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
// a / d for a wave-uniform d, with the instructions the compiler emits for `a / d` (v_div_scale, v_rcp, two Newton steps on
// the reciprocal, three on the quotient, v_div_fmas, v_div_fixup) minus what does not depend on a - the reciprocal and its
// refinement, hoisted out of the loop - and minus the scaling, which is the identity (scale 1, VCC = 0) whenever d is
// normal, 1/d is normal, |a| >= 2^-102, the exponents of a and d differ by less than 96 and a / d is normal
// (V_DIV_SCALE_F32 in the CDNA ISA).  emit_fast_div_ok checks on the host that the call's camera parameters keep every
// pixel inside those conditions; otherwise the kernel divides the plain way.  Same quotient bit for bit: 6 instructions
// instead of 11 per division, and the pass is bound by instruction issue.
__device__ __forceinline__ float refined_rcp(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_by_uniform(float a, float d, float r) {
    const float q0 = a * r;
    const float q1 = fmaf(fmaf(-d, q0, a), r, q0);
    const float q2 = fmaf(fmaf(-d, q1, a), r, q1);
    return __builtin_amdgcn_div_fixupf(q2, d, a);
}
#ifndef DE_TPW_VALUE
#define DE_TPW_VALUE 4
#endif
constexpr int DE_TPW = DE_TPW_VALUE;   // tiles per wave
// Four of five tiles of an instance hold no valid pixel.  With one tile per wave the pass was bound by the rate at which
// workgroups can be launched (151k waves, most of them gone after one load; 19 % of the wave slots occupied — SQ counters
// in profiles/r2): a wave therefore takes DE_TPW tiles, strided by the number of waves of its instance so that every wave
// gets its share of the contiguous band of occupied tiles, and loads all their bitmap words before it looks at the first.
// ROWS (pixel count a multiple of 1,024: whole tiles only): the 64-bit validity mask of round j IS the j-th 8-byte word of
// the tile's 128-byte row of the bitmap, at a wave-uniform address — the wave reads the 16 masks of a tile with two
// scalar loads instead of assembling each from four v_readlane (the SQ counters showed 755 scalar next to 840 vector
// instructions per wave, most of them this assembling).
template <bool FAST_DIV, bool ROWS>
__global__ __launch_bounds__(DE_BLOCK)
void k_depth_emit_bits(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ bgr0, const uint16_t* __restrict__ bits,
                       int width, size_t n, float inv_scale, float fx, float fy, float cx, float cy, int ntiles,
                       const int* __restrict__ offsets, float* __restrict__ xyz, float* __restrict__ rgb) {
    __shared__ __attribute__((aligned(16))) uint16_t zraw[DE_BLOCK / 64][DB_TILE];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = __builtin_amdgcn_readfirstlane(blockIdx.x * (DE_BLOCK / 64) + wave), stride = gridDim.x * (DE_BLOCK / 64);   // wave-uniform, and known as such
    const size_t frame = frame_of ? (size_t)frame_of[b] : 0;
    const uint16_t* __restrict__ raw = raw0 + frame * n;
    const uint8_t* __restrict__ bgr = (rgb && bgr0) ? bgr0 + frame * n * 3 : nullptr;
    const unsigned bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane < 32 ? 0u : 1u << (lane - 32);
    const float rfx = refined_rcp(fx), rfy = refined_rcp(fy);
    unsigned words[DE_TPW];                                           // validity of pixels i0 .. i0 + 15 of each of the wave's tiles
#pragma unroll
    for (int k = 0; k < DE_TPW; ++k) {
        const int tile = first + k * stride;
        const size_t i0 = (size_t)tile * DB_TILE + (size_t)lane * DB_PX;
        words[k] = (tile < ntiles && i0 < n) ? (unsigned)bits[(size_t)b * (n / DB_PX) + i0 / DB_PX] : 0u;
    }
#pragma unroll
    for (int k = 0; k < DE_TPW; ++k) {
        const unsigned word = words[k];
        if (!__any(word != 0u)) continue;                             // most tiles of an instance: nothing else is read
        const int tile = first + k * stride;
        const size_t t0 = (size_t)tile * DB_TILE;                     // first pixel of the tile
        const size_t i0 = t0 + (size_t)lane * DB_PX;
        __builtin_amdgcn_wave_barrier();                              // the previous tile's rounds are done with zraw
        if (i0 < n) {                                                 // n is a multiple of 16: 32 B per lane, whole or not at all
            *reinterpret_cast<uint4*>(&zraw[wave][lane * DB_PX]) = *reinterpret_cast<const uint4*>(raw + i0);
            *reinterpret_cast<uint4*>(&zraw[wave][lane * DB_PX + 8]) = *reinterpret_cast<const uint4*>(raw + i0 + 8);
        }
        __builtin_amdgcn_wave_barrier();                              // a wave reads back only what it staged itself
        const size_t slot0 = (size_t)offsets[(size_t)b * ntiles + tile];    // output slot of the tile's first valid pixel
        float* __restrict__ tile_xyz = xyz + 3 * slot0;
        float* __restrict__ tile_rgb = bgr ? rgb + 3 * slot0 : nullptr;
        unsigned run = 0u;                                            // valid pixels of the tile's earlier rounds (32-bit offsets from here on)
        // (column, row) of this lane's pixel, advanced by 64 pixels per round: a division per pixel would be a third of
        // the loop's instructions
        unsigned row = (unsigned)(t0 / (size_t)width);
        unsigned col = (unsigned)(t0 - (size_t)row * (size_t)width) + (unsigned)lane;
        while (col >= (unsigned)width) { col -= (unsigned)width; ++row; }
        unsigned long long mrow[ROWS ? DB_PX : 1];
        if (ROWS) {
            const unsigned long long* __restrict__ mp = reinterpret_cast<const unsigned long long*>(bits + (size_t)b * (n / DB_PX)) + (size_t)tile * DB_PX;   // wave-uniform: scalar loads
#pragma unroll
            for (int j = 0; j < DB_PX; ++j) mrow[j] = mp[j];
        }
#pragma unroll ROWS ? DB_PX : 1
        for (int j = 0; j < DB_PX; ++j) {                             // round j: pixels t0 + 64 j .. + 63
            unsigned m_lo, m_hi;
            if (ROWS) { m_lo = (unsigned)mrow[ROWS ? j : 0]; m_hi = (unsigned)(mrow[ROWS ? j : 0] >> 32); }
            else {
                m_lo = (unsigned)__builtin_amdgcn_readlane((int)word, 4 * j) | ((unsigned)__builtin_amdgcn_readlane((int)word, 4 * j + 1) << 16);
                m_hi = (unsigned)__builtin_amdgcn_readlane((int)word, 4 * j + 2) | ((unsigned)__builtin_amdgcn_readlane((int)word, 4 * j + 3) << 16);
            }
            if (m_lo | m_hi) {                                        // wave-uniform
                if ((m_lo & bit_lo) | (m_hi & bit_hi)) {
                    const float z = (float)zraw[wave][64 * j + lane] * inv_scale;   // pipeline.cpp:47
                    const unsigned slot = 3u * (run + __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u)));   // + valid lanes below this one
                    float* __restrict__ o = tile_xyz + slot;
                    const float ax = ((float)col - cx) * z, ay = ((float)row - cy) * z;
                    o[0] = FAST_DIV ? div_by_uniform(ax, fx, rfx) : ax / fx;   // pipeline.cpp:73
                    o[1] = FAST_DIV ? div_by_uniform(ay, fy, rfy) : ay / fy;   // pipeline.cpp:74
                    o[2] = z;
                    if (bgr) {
                        const uint8_t* p = bgr + (t0 + 64 * (size_t)j + lane) * 3;
                        float* __restrict__ c = tile_rgb + slot;
                        c[0] = (float)p[2] / 255.0f; c[1] = (float)p[1] / 255.0f; c[2] = (float)p[0] / 255.0f;   // pipeline.cpp:78-80
                    }
                }
                run += __popc(m_lo) + __popc(m_hi);
            }
            col += 64u;
            while (col >= (unsigned)width) { col -= (unsigned)width; ++row; }
        }
    }
}

// ---- label images: every instance's cloud from ONE pass over the frame (round 3) ---------------------------------------------
// A label image assigns each pixel to at most one instance (label = instance + 1, u8 or u16), so the per-instance kernels above
// - a grid of (pixel blocks x instances), every instance re-reading the whole image: 1.7 ms for C5's 1,024 instances - are the
// wrong shape.  Here a wave owns a tile of 1,024 consecutive pixels and walks it in 16 rounds of 64 (lane = pixel).  The valid
// pixels of a round are grouped by label (one pass per distinct label: an object's pixels are neighbours, a round holds one to
// three labels): pass 1 adds the group's size to counts[instance][tile]; after the usual scan, pass 2 takes the group's base
// from the scanned entry with an atomic add of its size - only this wave ever touches (instance, tile), and it walks its rounds
// in order, so the bases it gets are the row-major ones - and every pixel writes at base + its rank among the group's lanes.
// Frame bytes are read twice (4 B per pixel and pass), the clouds written once.
template <bool L16>
__device__ __forceinline__ int label_at(const uint8_t* __restrict__ labels, size_t i) {
    return L16 ? (int)reinterpret_cast<const uint16_t*>(labels)[i] : (int)labels[i];
}
template <bool L16>
__global__ __launch_bounds__(256)
void k_label_count(const uint16_t* __restrict__ raw, const uint8_t* __restrict__ labels, size_t n, int n_inst, float inv_scale, float zmax, int ntiles,
                   int* __restrict__ counts) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    for (int j = 0; j < 16; ++j) {
        const size_t i = (size_t)tile * 1024 + 64 * j + lane;
        int L = 0;
        if (i < n) {
            const float z = (float)raw[i] * inv_scale;                         // pipeline.cpp:47
            const int l = label_at<L16>(labels, i);
            if (!(z <= 0.f || z > zmax) && l >= 1 && l <= n_inst) L = l;       // pipeline.cpp:71
        }
        unsigned long long todo = __ballot(L != 0);
        while (todo) {                                                          // wave-uniform
            const int src = __builtin_ctzll(todo);
            const int Ls = __builtin_amdgcn_readlane(L, src);
            const unsigned long long m = __ballot(L == Ls);
            if (lane == src) atomicAdd(&counts[(size_t)(Ls - 1) * ntiles + tile], __popcll(m));
            todo &= ~m;
        }
    }
}
template <bool L16>
__global__ __launch_bounds__(256)
void k_label_emit(const uint16_t* __restrict__ raw, const uint8_t* __restrict__ labels, const uint8_t* __restrict__ bgr, int width, size_t n, int n_inst,
                  float inv_scale, float fx, float fy, float cx, float cy, float zmax, int ntiles, int* __restrict__ offsets,
                  float* __restrict__ xyz, float* __restrict__ rgb) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    for (int j = 0; j < 16; ++j) {
        const size_t i = (size_t)tile * 1024 + 64 * j + lane;
        int L = 0; float z = 0.f;
        if (i < n) {
            z = (float)raw[i] * inv_scale;
            const int l = label_at<L16>(labels, i);
            if (!(z <= 0.f || z > zmax) && l >= 1 && l <= n_inst) L = l;
        }
        unsigned long long todo = __ballot(L != 0);
        int slot = 0;
        while (todo) {
            const int src = __builtin_ctzll(todo);
            const int Ls = __builtin_amdgcn_readlane(L, src);
            const unsigned long long m = __ballot(L == Ls);
            int base = 0;
            if (lane == src) base = atomicAdd(&offsets[(size_t)(Ls - 1) * ntiles + tile], __popcll(m));
            base = __builtin_amdgcn_readlane(base, src);
            if (L == Ls) slot = base + __popcll(m & ((1ull << lane) - 1ull));
            todo &= ~m;
        }
        if (L != 0) {
            const int v = (int)(i / width), u = (int)(i - (size_t)v * width);
            const size_t o = 3 * (size_t)slot;
            xyz[o] = ((float)u - cx) * z / fx;                                    // pipeline.cpp:73
            xyz[o + 1] = ((float)v - cy) * z / fy;                                // pipeline.cpp:74
            xyz[o + 2] = z;
            if (rgb && bgr) {
                const uint8_t* p = bgr + i * 3;
                rgb[o] = (float)p[2] / 255.0f; rgb[o + 1] = (float)p[1] / 255.0f; rgb[o + 2] = (float)p[0] / 255.0f;   // pipeline.cpp:78-80
            }
        }
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

// The raw values r with `z > 0 && z <= zmax`, z = (float)r * inv_scale, evaluated exactly as the kernels do (pipeline.cpp:47,71:
// `!(z <= 0 || z > zmax)`).  For a finite inv_scale >= 0 the product is monotone in r and never NaN, so the values form one
// range, found by two bisections; anything else (an infinite scale factor makes 0 * inf = NaN pass the test) is left to the
// per-pixel kernels.  An empty range comes back as [1, 0].
static bool depth_valid_range(float inv_scale, float zmax, int* lo, int* hi) {
    if (!(inv_scale >= 0.f) || !(inv_scale <= FLT_MAX)) return false;
    auto z = [inv_scale](int r) { volatile float v = (float)r * inv_scale; return (float)v; };
    int a = 0, b = 65536;                      // first r with z(r) > 0 (65536: none)
    while (a < b) { const int m = (a + b) / 2; if (z(m) > 0.f) b = m; else a = m + 1; }
    const int first = a;
    a = 0; b = 65536;                          // first r with z(r) > zmax (65536: none; NaN zmax: none, as in the test itself)
    while (a < b) { const int m = (a + b) / 2; if (z(m) > zmax) b = m; else a = m + 1; }
    *lo = first; *hi = a - 1;
    if (*lo > *hi) { *lo = 1; *hi = 0; }
    return true;
}

// Camera parameters for which div_by_uniform is `/` bit for bit on every pixel (see there).  a = (col - c) * z with
// integer col, row < 2^16: col - c is 0 or at least half an ulp of c in size, hence >= 2^-34 for |c| >= 2^-10, and at
// most 2^21; z = raw * inv_scale in [2^-30, 2^26]; so a is 0 (both forms return a signed zero through v_div_fixup) or
// 2^-64 <= |a| <= 2^47, and with 2^-10 <= |f| <= 2^30 the quotient stays within [2^-94, 2^57].
static bool emit_fast_div_ok(float fx, float fy, float cx, float cy, float inv_scale, int w, int h) {
    auto focal = [](float f) { const float a = std::fabs(f); return a >= 0x1p-10f && a <= 0x1p30f; };
    auto centre = [](float c) { const float a = std::fabs(c); return a == 0.f || (a >= 0x1p-10f && a <= 0x1p20f); };
    return focal(fx) && focal(fy) && centre(cx) && centre(cy) && inv_scale >= 0x1p-30f && inv_scale <= 0x1p10f && w <= 65536 && h <= 65536;
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
    int raw_lo = 1, raw_hi = 0;
    const bool label_image = stacked != 1;                      // one u8 (0) or u16 (2) label image: the one-pass kernels
    const bool vec = !label_image && batch_vectorisable(d_raw, d_masks, n) && depth_valid_range(inv_scale, zmax, &raw_lo, &raw_hi);
    const int blocks = vec ? (int)((n + DB_TILE - 1) / DB_TILE) : (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);   // tiles resp. workgroups per instance (1,024 pixels either way)
    int *counts, *offsets, *d_total, *d_inst;
    TDV_TRY(ws_alloc(ctx, (size_t)blocks * n_inst, &counts));
    TDV_TRY(ws_alloc(ctx, (size_t)blocks * n_inst, &offsets));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(ws_alloc(ctx, (size_t)n_inst + 1, &d_inst));
    ctx->depth_bits = nullptr;
    if (vec) TDV_TRY(ws_alloc(ctx, (size_t)n_inst * (n / DB_PX), &ctx->depth_bits));
    hipStream_t s = ctx->stream;
    if (label_image && d_frame_of) return TDV_ERR_BAD_ARG;      // a label image belongs to one frame
    if (label_image) TDV_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)blocks * n_inst * 4, s));
    {
        ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
        if (label_image) {
            if (stacked == 2) k_label_count<true><<<(blocks + 3) / 4, 256, 0, s>>>(d_raw, d_masks, n, n_inst, inv_scale, zmax, blocks, counts);
            else k_label_count<false><<<(blocks + 3) / 4, 256, 0, s>>>(d_raw, d_masks, n, n_inst, inv_scale, zmax, blocks, counts);
        } else if (vec) {
            const dim3 grid((blocks + DP_BLOCK / 64 - 1) / (DP_BLOCK / 64), (n_inst + DB_GROUP - 1) / DB_GROUP);
#define TDV_DEPTH_BITS(ST, PF) k_depth_bits<ST, PF><<<grid, DP_BLOCK, 0, s>>>(d_raw, d_frame_of, d_masks, n, n_inst, raw_lo, raw_hi, mask_mode, blocks, ctx->depth_bits, counts)
            if (d_frame_of) TDV_DEPTH_BITS(true, true); else TDV_DEPTH_BITS(true, false);
#undef TDV_DEPTH_BITS
        } else k_valid_count_batch<<<dim3(blocks, n_inst), DP_BLOCK, 0, s>>>(d_raw, d_frame_of, d_masks, n, stacked, inv_scale, mask_mode, zmax, counts);
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
    const bool vec = stacked == 1 && batch_vectorisable(d_raw, d_masks, n) && ctx->depth_bits;   // pass 1 left a bitmap: it took the tiled path
    ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
    if (stacked != 1) {                                          // label image: the one-pass kernels (the scanned offsets serve as cursors)
        const int ntiles = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
        int* cursors = const_cast<int*>(d_offsets);
        if (stacked == 2) k_label_emit<true><<<(ntiles + 3) / 4, 256, 0, ctx->stream>>>(d_raw, d_masks, d_bgr, w, n, n_inst, inv_scale, fx, fy, cx, cy, zmax, ntiles, cursors, d_xyz, d_rgb);
        else k_label_emit<false><<<(ntiles + 3) / 4, 256, 0, ctx->stream>>>(d_raw, d_masks, d_bgr, w, n, n_inst, inv_scale, fx, fy, cx, cy, zmax, ntiles, cursors, d_xyz, d_rgb);
    } else if (vec) {
        const int ntiles = (int)((n + DB_TILE - 1) / DB_TILE);
        const int tiles_per_block = (DE_BLOCK / 64) * DE_TPW;
        const dim3 grid((ntiles + tiles_per_block - 1) / tiles_per_block, n_inst);
        const bool fd = emit_fast_div_ok(fx, fy, cx, cy, inv_scale, w, h), rows = n % DB_TILE == 0;
#define TDV_EMIT(FD, RW) k_depth_emit_bits<FD, RW><<<grid, DE_BLOCK, 0, ctx->stream>>>(d_raw, d_frame_of, d_bgr, ctx->depth_bits, w, n, inv_scale, fx, fy, cx, cy, ntiles, d_offsets, d_xyz, d_rgb)
        if (fd) { if (rows) TDV_EMIT(true, true); else TDV_EMIT(true, false); }
        else { if (rows) TDV_EMIT(false, true); else TDV_EMIT(false, false); }
#undef TDV_EMIT
    } else {
        const int blocks = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
        k_emit_batch<<<dim3(blocks, n_inst), DP_BLOCK, 0, ctx->stream>>>(d_raw, d_frame_of, d_masks, d_bgr, w, n, stacked, inv_scale, mask_mode,
                                                                          fx, fy, cx, cy, zmax, d_offsets, d_xyz, d_rgb);
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

// cv::resize(mask, resized, depth.size(), 0, 0, cv::INTER_NEAREST) of src/pipeline.cpp:38-41 for B stacked masks: a gather
// through the two index tables of OpenCV's resizeNN (imgproc/src/resize.cpp: x_ofs[x] = min(cvFloor(x * ifx), sw - 1) with
// ifx = 1. / ((double)dw / sw), likewise for rows), which the host computes in double exactly as OpenCV does.
__global__ __launch_bounds__(256)
void k_mask_resize_nn(const uint8_t* __restrict__ src, int sw, int sh, const int* __restrict__ x_ofs, const int* __restrict__ y_ofs,
                      int dw, int dh, uint8_t* __restrict__ dst) {
    const size_t b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)dw * dh) return;
    const int y = (int)(i / dw), x = (int)(i - (size_t)y * dw);
    dst[b * (size_t)dw * dh + i] = src[b * (size_t)sw * sh + (size_t)y_ofs[y] * sw + x_ofs[x]];
}

void resize_nn_tables(int sw, int sh, int dw, int dh, int* x_ofs, int* y_ofs) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double ifx = 1. / inv_scale_x, ify = 1. / inv_scale_y;
    for (int x = 0; x < dw; ++x) x_ofs[x] = std::min((int)std::floor(x * ifx), sw - 1);
    for (int y = 0; y < dh; ++y) y_ofs[y] = std::min((int)std::floor(y * ify), sh - 1);
}

int mask_resize_nearest_dev(tdv_ctx* ctx, const uint8_t* d_src, int n_masks, int sw, int sh, int dw, int dh, uint8_t* d_dst) {
    if (!ctx || n_masks < 0 || sw <= 0 || sh <= 0 || dw < 0 || dh < 0) return TDV_ERR_BAD_ARG;
    if (n_masks == 0 || (size_t)dw * dh == 0) return TDV_OK;
    if (!d_src || !d_dst) return TDV_ERR_BAD_ARG;
    std::vector<int> tab((size_t)dw + dh);
    resize_nn_tables(sw, sh, dw, dh, tab.data(), tab.data() + dw);
    int* d_tab;
    TDV_TRY(ws_alloc(ctx, tab.size(), &d_tab));
    TDV_HIP(ctx, hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    const size_t npx = (size_t)dw * dh;
    k_mask_resize_nn<<<dim3((unsigned)((npx + 255) / 256), n_masks), 256, 0, ctx->stream>>>(d_src, sw, sh, d_tab, d_tab + dw, dw, dh, d_dst);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));   // tab is a host temporary
    return TDV_OK;
}

// Why an instance came out of the batched count pass with no point: does its masked depth image hold ANY non-zero value
// (cv::countNonZero(scaled_depth) of src/pipeline.cpp:57)?  If not the reference stops at :57-60 ("empty depth after
// masking"), else at :86-89 ("empty point cloud": every masked pixel lies beyond the z clip).  Run for the empty instances
// only - rare - so the count pass itself stays as it is.
__global__ __launch_bounds__(DP_BLOCK)
void k_masked_nonzero_any(const uint16_t* __restrict__ raw0, const int* __restrict__ frame_of, const uint8_t* __restrict__ masks, size_t n, int layout,
                          float inv_scale, int mask_mode, const int* __restrict__ inst, int* __restrict__ flags) {
    const int b = inst[blockIdx.y];
    const uint16_t* __restrict__ raw = raw0 + (frame_of ? (size_t)frame_of[b] : 0) * n;
    bool any = false;
    for (int k = 0; k < DP_PX_PER_THREAD; ++k) {
        const size_t i = (size_t)blockIdx.x * DP_PX_PER_BLOCK + k * DP_BLOCK + threadIdx.x;
        if (i < n) any |= batch_pixel_depth(raw, masks, n, b, layout, i, inv_scale, mask_mode) != 0.f;
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(&flags[blockIdx.y], 1);
}

int depth_batch_nonzero_any(tdv_ctx* ctx, const uint16_t* d_raw, const int* d_frame_of, const uint8_t* d_masks, int layout, int w, int h, float scale,
                            int mask_mode, const int* h_inst, int n_list, int* h_flags) {
    if (n_list <= 0) return TDV_OK;
    const size_t n = (size_t)w * h;
    const float inv_scale = (float)(1.0 / (double)scale);
    int *d_inst, *d_flags;
    TDV_TRY(ws_alloc(ctx, (size_t)n_list, &d_inst));
    TDV_TRY(ws_alloc(ctx, (size_t)n_list, &d_flags));
    TDV_HIP(ctx, hipMemcpyAsync(d_inst, h_inst, (size_t)n_list * 4, hipMemcpyHostToDevice, ctx->stream));
    TDV_HIP(ctx, hipMemsetAsync(d_flags, 0, (size_t)n_list * 4, ctx->stream));
    const int blocks = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
    k_masked_nonzero_any<<<dim3(blocks, n_list), DP_BLOCK, 0, ctx->stream>>>(d_raw, d_frame_of, d_masks, n, layout, inv_scale, mask_mode, d_inst, d_flags);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipMemcpyAsync(h_flags, d_flags, (size_t)n_list * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    TDV_TRY(pin_reserve(ctx, 64));
    hipStream_t s = ctx->stream;
    int* h_total = reinterpret_cast<int*>(ctx->pin);
    const bool three_pass = study_env("TDV_DEPTH_THREE_PASS") && atoi(study_env("TDV_DEPTH_THREE_PASS")) == 1;     // (study build: rounds 1-3)
    if (!three_pass) {
        const int tiles = (int)((n + CH_TILE - 1) / CH_TILE);
        // the chain's persistent state: status words (never cleared: the epoch tells this call's words from older ones) and the ticket
        if ((size_t)tiles > ctx->chain_cap || ctx->chain_epoch >= (1u << 30) - 2u) {
            TDV_HIP(ctx, hipStreamSynchronize(s));
            if (ctx->chain_status) (void)hipFree(ctx->chain_status);
            ctx->chain_status = nullptr; ctx->chain_cap = 0;
            const size_t cap = std::max<size_t>((size_t)tiles, 4096);
            TDV_HIP(ctx, hipMalloc((void**)&ctx->chain_status, cap * sizeof(unsigned long long)));
            TDV_HIP(ctx, hipMemsetAsync(ctx->chain_status, 0, cap * sizeof(unsigned long long), s));     // on the ctx's stream: ordered before the kernel (the stream is non-blocking - a null-stream memset would not be)
            ctx->chain_cap = cap; ctx->chain_epoch = 0;
        }
        const unsigned epoch = ++ctx->chain_epoch;
        unsigned* ticket = ctx->scan_ticket + 8;
        h_total[0] = -1; h_total[1] = 0;
        float* xyz = (d_xyz && capacity > 0) ? d_xyz : nullptr;
        const bool vec = d_raw ? (((uintptr_t)d_raw & 7) == 0 && (!d_mask || ((uintptr_t)d_mask & 3) == 0)) : (((uintptr_t)d_depth & 15) == 0);
        {
            ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
#define TDV_CHAIN(RW, VC) k_depth_cloud_chain<RW, VC><<<tiles, CH_BLOCK, 0, s>>>(d_raw, d_depth, d_raw ? d_mask : nullptr, d_bgr, w, n, d_raw ? inv_scale : 0.f, \
            d_raw ? mask_mode : 0, fx, fy, cx, cy, zmax, ticket, ctx->chain_ticket_base, ctx->chain_status, epoch, tiles, capacity, xyz, d_rgb, h_total)
            if (d_raw) { if (vec) TDV_CHAIN(true, true); else TDV_CHAIN(true, false); }
            else { if (vec) TDV_CHAIN(false, true); else TDV_CHAIN(false, false); }
#undef TDV_CHAIN
        }
        ctx->chain_ticket_base += (unsigned)tiles;
        const hipError_t le = hipGetLastError();
        const hipError_t se = le == hipSuccess ? hipStreamSynchronize(s) : le;
        if (se != hipSuccess || h_total[1] != 0 || h_total[0] < 0) {
            // the ticket word and the host's base may no longer agree (a launch that did not run): start over
            (void)hipMemsetAsync(ctx->scan_ticket + 8, 0, 4, s); (void)hipStreamSynchronize(s); ctx->chain_ticket_base = 0;
            if (se != hipSuccess) return set_err(ctx, se, "k_depth_cloud_chain", __LINE__);
            snprintf(ctx->err, sizeof(ctx->err), "%s", h_total[1] ? "depth_to_cloud: the chained scan gave up waiting for a tile" : "depth_to_cloud: the count did not reach the host");
            return TDV_ERR_INTERNAL;
        }
        *n_out = h_total[0];
        if (h_total[0] > capacity) return TDV_ERR_BAD_ARG;  // caller's buffers too small; *n_out = needed
        return TDV_OK;
    }
#ifdef TDV_STUDY
    const int blocks = (int)((n + DP_PX_PER_BLOCK - 1) / DP_PX_PER_BLOCK);
    int *counts, *offsets;
    TDV_TRY(ws_alloc(ctx, (size_t)blocks, &counts));
    TDV_TRY(ws_alloc(ctx, (size_t)blocks + 1, &offsets));
    {
        ScopedTimer tm(ctx, TDV_TIMER_DEPTH);
        if (d_raw) k_valid_count<true>
<<<blocks, DP_BLOCK, 0, s>>>(d_raw, nullptr, d_mask, n, inv_scale, mask_mode, zmax, counts);
        else k_valid_count<false><<<blocks, DP_BLOCK, 0, s>>>(nullptr, d_depth, nullptr, n, 0.f, 0, zmax, counts);
        int* h_total0 = reinterpret_cast<int*>(ctx->pin);
        *h_total0 = -1;
        k_block_scan<<<1, 1024, 0, s>>>(counts, blocks, offsets, h_total0);
        if (d_xyz && capacity > 0) {
            if (d_raw) k_emit<true><<<blocks, DP_BLOCK, 0, s>>>(d_raw, nullptr, d_mask, d_bgr, w, n, inv_scale, mask_mode, fx, fy, cx, cy, zmax, offsets, capacity, d_xyz, d_rgb);
            else k_emit<false><<<blocks, DP_BLOCK, 0, s>>>(nullptr, d_depth, nullptr, d_bgr, w, n, 0.f, 0, fx, fy, cx, cy, zmax, offsets, capacity, d_xyz, d_rgb);
        }
    }
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipStreamSynchronize(s));
    if (*h_total < 0) { snprintf(ctx->err, sizeof(ctx->err), "depth_to_cloud: the count did not reach the host"); return TDV_ERR_INTERNAL; }
    *n_out = *h_total;
    if (*h_total > capacity) return TDV_ERR_BAD_ARG;  // caller's buffers too small; *n_out = needed
    return TDV_OK;
#else
    return TDV_ERR_INTERNAL;
#endif
}

}  // namespace tdv
