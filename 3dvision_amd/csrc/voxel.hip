// Voxel-grid downsample on gfx950 (HBM-bound: 24 B in + 24 B out per point, plus the sort).
//
// Replaces Registration::voxelDownsample (/root/reference/src/registration.cpp:29-60; key/hash
// :15-27), which has no GPU entry point in the reference (src/pipeline.cpp:92 calls the CPU static).
// Semantics kept: key = (int)floor(p * (1/voxel)) per axis; per-voxel mean = f32 sum of the member
// points IN ASCENDING INPUT INDEX divided by float(count); colours likewise; normals dropped.
//
// The reference groups with std::unordered_map and emits voxels in that container's iteration
// order.  Here grouping is by hashing + tiny per-bucket sorts (below), with a full sort as the fallback:
//   1. k_voxel_records   : record (kx, ky, kz, idx) per point
//   2. bitonic sort      : ascending by (kx, ky, kz, idx) as unsigned words — members of a voxel
//                          become one run, in ascending input index (LDS-tiled local passes,
//                          global passes only for strides >= the 2048-record tile)
//   3. k_voxel_heads     : run heads; a run's first record carries the voxel's smallest index
//   4. exclusive scan    : rank of every leader index = voxel position in FIRST-OCCURRENCE order
//   5. k_voxel_means     : one lane per run, sequential sum in run order -> mean -> out[rank]
// TDV_VOXEL_ORDER_FIRST stops here.  TDV_VOXEL_ORDER_REFERENCE additionally replays the
// reference's container on the host (same key, same hash, the node-list manipulation of this libstdc++'s
// std::unordered_map) to obtain its iteration order, and permutes the means on the device.  Only the first
// point of every voxel changes a container, so the host receives 16 B per VOXEL (cell + input index) from the
// device and nothing else: the cloud stays in HBM, which is what lets the batched chain run in the reference's order.
#include "tdv_internal.hpp"
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <algorithm>

namespace tdv {

constexpr int BT_TILE = 2048;     // records per LDS tile (32 KB)
constexpr int BT_THREADS = 1024;  // one compare-exchange per thread per pass

__device__ __forceinline__ bool rec_less(const uint4& a, const uint4& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.z != b.z) return a.z < b.z;
    return a.w < b.w;
}

__global__ void k_voxel_records(const float* __restrict__ xyz, int n, int n_pow2, float inv, uint4* __restrict__ rec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    uint4 r;
    if (i < n) {
        r.x = (unsigned)(int)floorf(xyz[3 * i] * inv);
        r.y = (unsigned)(int)floorf(xyz[3 * i + 1] * inv);
        r.z = (unsigned)(int)floorf(xyz[3 * i + 2] * inv);
        r.w = (unsigned)i;
    } else {
        r = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);  // padding sorts last
    }
    rec[i] = r;
}

__device__ __forceinline__ void cmpxchg(uint4& a, uint4& b, bool ascending) {
    bool sw = ascending ? rec_less(b, a) : rec_less(a, b);
    if (sw) { uint4 t = a; a = b; b = t; }
}

// full bitonic sort of each 2048-record tile (all k <= BT_TILE)
__global__ __launch_bounds__(BT_THREADS)
void k_bitonic_local_sort(uint4* __restrict__ rec) {
    __shared__ uint4 s[BT_TILE];
    const size_t base = (size_t)blockIdx.x * BT_TILE;
    const int t = threadIdx.x;
    s[t] = rec[base + t]; s[t + BT_THREADS] = rec[base + t + BT_THREADS];
    __syncthreads();
    for (int k = 2; k <= BT_TILE; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            int i = ((t / j) * (j << 1)) + (t % j);
            bool asc = (((base + i) & (size_t)k) == 0);
            cmpxchg(s[i], s[i + j], asc);
            __syncthreads();
        }
    }
    rec[base + t] = s[t]; rec[base + t + BT_THREADS] = s[t + BT_THREADS];
}

// one global compare-exchange pass (stride j >= BT_TILE)
__global__ __launch_bounds__(256)
void k_bitonic_global_step(uint4* __restrict__ rec, size_t n_pairs, unsigned k, unsigned j) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_pairs) return;
    size_t i = ((t / j) * ((size_t)j << 1)) + (t % j);
    uint4 a = rec[i], b = rec[i + j];
    bool asc = ((i & (size_t)k) == 0);
    bool sw = asc ? rec_less(b, a) : rec_less(a, b);
    if (sw) { rec[i] = b; rec[i + j] = a; }
}

// two global passes in one launch (strides j and j / 2, both >= BT_TILE): a thread holds the four records its two
// compare-exchanges per pass touch.  The sort of 131k-262k records is a chain of ~30 tiny launches; this removes a third.
__global__ __launch_bounds__(256)
void k_bitonic_global_step2(uint4* __restrict__ rec, size_t n_quads, unsigned k, unsigned j) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_quads) return;
    const size_t h = j >> 1;                                              // the second stride
    const size_t i = ((t / h) * ((size_t)j << 1)) + (t % h);              // bits j and h of i are clear
    uint4 a = rec[i], b = rec[i + h], c = rec[i + j], d = rec[i + j + h];
    const bool asc = ((i & (size_t)k) == 0);                              // k > j: the same for all four
    cmpxchg(a, c, asc); cmpxchg(b, d, asc);                               // stride j
    cmpxchg(a, b, asc); cmpxchg(c, d, asc);                               // stride j / 2
    rec[i] = a; rec[i + h] = b; rec[i + j] = c; rec[i + j + h] = d;
}

// all passes with stride < BT_TILE of merge stage k, inside LDS
__global__ __launch_bounds__(BT_THREADS)
void k_bitonic_local_merge(uint4* __restrict__ rec, unsigned k) {
    __shared__ uint4 s[BT_TILE];
    const size_t base = (size_t)blockIdx.x * BT_TILE;
    const int t = threadIdx.x;
    s[t] = rec[base + t]; s[t + BT_THREADS] = rec[base + t + BT_THREADS];
    __syncthreads();
    for (int j = BT_TILE >> 1; j > 0; j >>= 1) {
        int i = ((t / j) * (j << 1)) + (t % j);
        bool asc = (((base + i) & (size_t)k) == 0);
        cmpxchg(s[i], s[i + j], asc);
        __syncthreads();
    }
    rec[base + t] = s[t]; rec[base + t + BT_THREADS] = s[t + BT_THREADS];
}

// ascending bitonic sort of n_pow2 (power of two, >= BT_TILE) uint4 records by (x, y, z, w); also used by knn.hip
int sort_records_dev(tdv_ctx* ctx, uint4* rec, size_t n_pow2) {
    hipStream_t s = ctx->stream;
    const unsigned tiles = (unsigned)(n_pow2 / BT_TILE);
    k_bitonic_local_sort<<<tiles, BT_THREADS, 0, s>>>(rec);
    for (size_t k = (size_t)BT_TILE << 1; k <= n_pow2; k <<= 1) {
        for (size_t j = k >> 1; j >= BT_TILE; j >>= 1) {
            if ((j >> 1) >= BT_TILE) {
                k_bitonic_global_step2<<<(unsigned)((n_pow2 / 4 + 255) / 256), 256, 0, s>>>(rec, n_pow2 / 4, (unsigned)k, (unsigned)j);
                j >>= 1;
            } else {
                k_bitonic_global_step<<<(unsigned)((n_pow2 / 2 + 255) / 256), 256, 0, s>>>(rec, n_pow2 / 2, (unsigned)k, (unsigned)j);
            }
        }
        k_bitonic_local_merge<<<tiles, BT_THREADS, 0, s>>>(rec, (unsigned)k);
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

// Every segment [seg_start[c], seg_start[c + 1]) of rec sorted on its own, one workgroup and one launch for all of them:
// for segments of at most BT_TILE records (the caller checks), e.g. the columns of the descriptor index, whose records
// only have to be ordered inside their column.
__global__ __launch_bounds__(BT_THREADS)
void k_bitonic_segment_sort(uint4* __restrict__ rec, const int* __restrict__ seg_start) {
    __shared__ uint4 s[BT_TILE];
    const int c0 = seg_start[blockIdx.x], m = seg_start[blockIdx.x + 1] - c0;
    if (m <= 1) return;
    const int t = threadIdx.x;
    const uint4 pad = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    s[t] = t < m ? rec[(size_t)c0 + t] : pad;
    s[t + BT_THREADS] = t + BT_THREADS < m ? rec[(size_t)c0 + t + BT_THREADS] : pad;
    __syncthreads();
    int span = 2;
    while (span < m) span <<= 1;                      // the padded power of two that holds the segment
    for (int k = 2; k <= span; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((t / j) * (j << 1)) + (t % j);
            if (i + j < span) cmpxchg(s[i], s[i + j], (i & k) == 0);
            __syncthreads();
        }
    }
    if (t < m) rec[(size_t)c0 + t] = s[t];
    if (t + BT_THREADS < m) rec[(size_t)c0 + t + BT_THREADS] = s[t + BT_THREADS];
}
int segment_sort_records_dev(tdv_ctx* ctx, uint4* rec, const int* d_seg_start, int nseg) {
    if (nseg <= 0) return TDV_OK;
    k_bitonic_segment_sort<<<nseg, BT_THREADS, 0, ctx->stream>>>(rec, d_seg_start);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}
int segment_sort_max_len() { return BT_TILE; }

// ---- grouping by hashing (the default; the bitonic sort above is the fallback) -------------------------------------
// Members of a voxel only have to become one run in ascending input index; the order of the runs is irrelevant
// (positions come from the leader scan).  So: bucket = hash(cell) into >= 2n buckets (counting sort, integer atomics),
// then every bucket — a handful of records, possibly of several colliding cells — is insertion-sorted by
// (cell, index) by one lane, which also emits the bucket's voxels (k_voxel_bucket_emit).  A bucket larger than VX_MAX_BUCKET (a very coarse grid) raises a flag and the call is
// redone with the full sort.
constexpr int VX_MAX_BUCKET = 96;
__device__ __forceinline__ unsigned voxel_hash(unsigned x, unsigned y, unsigned z) {
    unsigned h = x * 73856093u ^ y * 19349663u ^ z * 83492791u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    return h;
}
__global__ void k_voxel_hist(const float* __restrict__ xyz, int n, float inv, unsigned mask, uint4* __restrict__ rec_in, int* __restrict__ hist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 r;
    r.x = (unsigned)(int)floorf(xyz[3 * i] * inv);
    r.y = (unsigned)(int)floorf(xyz[3 * i + 1] * inv);
    r.z = (unsigned)(int)floorf(xyz[3 * i + 2] * inv);
    r.w = (unsigned)i;
    rec_in[i] = r;
    atomicAdd(&hist[voxel_hash(r.x, r.y, r.z) & mask], 1);
}
__global__ void k_voxel_scatter(const uint4* __restrict__ rec_in, int n, unsigned mask, const int* __restrict__ start, int* __restrict__ cursor,
                                uint4* __restrict__ rec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = rec_in[i];
    const unsigned b = voxel_hash(r.x, r.y, r.z) & mask;
    rec[start[b] + atomicAdd(&cursor[b], 1)] = r;
}
// The hashed path's third kernel: every lane owns one bucket.  It sorts the bucket's few records by (cell, index); the
// members of a voxel then form a run INSIDE the bucket (a cell hashes to one bucket), so the same lane marks the run's
// leader (its smallest input index), sums the members in ascending index order and parks the mean at the leader's index.
// One launch instead of bucket sort + heads + means; the slot of a mean (its first-occurrence rank) comes from the scan of
// the leader flags, after which k_voxel_compact moves it there.
// The bucket's records sorted by (cell, index), its voxels emitted.  `a` points at the bucket's records - in LDS (the normal case:
// the workgroup's 256 buckets are one contiguous stretch of the record array, staged with coalesced loads) or in global memory.
__device__ __forceinline__ void voxel_emit_bucket(uint4* a, int m, const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                  int* __restrict__ leader, float* __restrict__ mean_xyz, float* __restrict__ mean_rgb) {
    for (int e = 1; e < m; ++e) {   // insertion sort by (x, y, z, index)
        const uint4 key = a[e];
        int f = e - 1;
        while (f >= 0 && rec_less(key, a[f])) { a[f + 1] = a[f]; --f; }
        a[f + 1] = key;
    }
    for (int e = 0; e < m;) {
        const uint4 r = a[e];
        float ax = 0.f, ay = 0.f, az = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        int cnt = 0;
        for (; e < m; ++e) {
            const uint4 q = a[e];
            if (!(q.x == r.x && q.y == r.y && q.z == r.z)) break;
            const size_t idx = q.w;
            ax += xyz[3 * idx]; ay += xyz[3 * idx + 1]; az += xyz[3 * idx + 2];        // registration.cpp:47-50, ascending input index
            if (rgb) { cr += rgb[3 * idx]; cg += rgb[3 * idx + 1]; cb += rgb[3 * idx + 2]; }
            ++cnt;
        }
        const float fn = (float)cnt;
        const size_t l = r.w;
        leader[l] = 1;
        mean_xyz[3 * l] = ax / fn; mean_xyz[3 * l + 1] = ay / fn; mean_xyz[3 * l + 2] = az / fn;   // :52-53
        if (rgb && mean_rgb) { mean_rgb[3 * l] = cr / fn; mean_rgb[3 * l + 1] = cg / fn; mean_rgb[3 * l + 2] = cb / fn; }
    }
}
constexpr int VX_LDS_REC = 1024;   // records of a workgroup's 256 buckets staged in LDS (16 KB); a denser stretch sorts in global memory
__global__ __launch_bounds__(256)
void k_voxel_bucket_emit(uint4* __restrict__ rec, const int* __restrict__ start, const int* __restrict__ hist, int nbuckets, int n_rec,
                         int* __restrict__ too_big, const float* __restrict__ xyz, const float* __restrict__ rgb,
                         int* __restrict__ leader, float* __restrict__ mean_xyz, float* __restrict__ mean_rgb) {
    // The insertion sort is a chain of dependent loads and stores: through global memory every step is a round trip of about a
    // microsecond and the kernel's time was that of its fullest bucket (42 us at 200k points with lanes alive for 3 us on
    // average); through LDS a step costs a hundred cycles: 25 us.  (Staging the records' points in LDS as well, so that the sums
    // run over LDS too, changed nothing; nor did a 12-comparator sorting network over registers for buckets of up to 6 records.)
    __shared__ uint4 s_rec[VX_LDS_REC];
    const int b0 = blockIdx.x * 256, b = b0 + threadIdx.x;
    const int base = start[b0];                                           // nbuckets is a multiple of 256
    const int end = b0 + 256 < nbuckets ? start[b0 + 256] : n_rec;
    const bool staged = end - base <= VX_LDS_REC;
    if (staged) for (int i = threadIdx.x; i < end - base; i += 256) s_rec[i] = rec[base + i];
    __syncthreads();
    const int m = hist[b];
    if (m == 0) return;
    if (m > VX_MAX_BUCKET) { *too_big = 1; return; }
    if (staged) voxel_emit_bucket(s_rec + (start[b] - base), m, xyz, rgb, leader, mean_xyz, mean_rgb);
    else voxel_emit_bucket(rec + start[b], m, xyz, rgb, leader, mean_xyz, mean_rgb);
}
// out[rank[i]] = mean parked at leader i
__global__ void k_voxel_compact(const int* __restrict__ leader, const int* __restrict__ rank, int n, const float* __restrict__ mean_xyz,
                                const float* __restrict__ mean_rgb, int capacity, float* __restrict__ out_xyz, float* __restrict__ out_rgb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !leader[i]) return;
    const int slot = rank[i];
    if (slot >= capacity) return;
    out_xyz[3 * (size_t)slot] = mean_xyz[3 * (size_t)i]; out_xyz[3 * (size_t)slot + 1] = mean_xyz[3 * (size_t)i + 1]; out_xyz[3 * (size_t)slot + 2] = mean_xyz[3 * (size_t)i + 2];
    if (mean_rgb && out_rgb) { out_rgb[3 * (size_t)slot] = mean_rgb[3 * (size_t)i]; out_rgb[3 * (size_t)slot + 1] = mean_rgb[3 * (size_t)i + 1]; out_rgb[3 * (size_t)slot + 2] = mean_rgb[3 * (size_t)i + 2]; }
}

// leader[idx] = 1 for the smallest input index of each voxel
__global__ void k_voxel_heads(const uint4* __restrict__ rec, int n, int* __restrict__ leader) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint4 r = rec[p];
    bool head = true;
    if (p > 0) { uint4 q = rec[p - 1]; head = !(q.x == r.x && q.y == r.y && q.z == r.z); }
    if (head) leader[r.w] = 1;
}

// For every leader i (the first point of each voxel), at its first-occurrence rank: its input index and its integer
// cell — all the host needs to replay the reference's container (the cloud itself never leaves the device).
__global__ void k_voxel_leader_list(const int* __restrict__ leader, const int* __restrict__ rank, const float* __restrict__ xyz, float inv,
                                    int n, int4* __restrict__ leaders) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && leader[i])
        leaders[rank[i]] = make_int4((int)floorf(xyz[3 * (size_t)i] * inv), (int)floorf(xyz[3 * (size_t)i + 1] * inv),
                                     (int)floorf(xyz[3 * (size_t)i + 2] * inv), i);
}

// exclusive scan of n ints in TWO launches: per-block (4096 items: 1024 threads x 4) reduce whose LAST workgroup (atomic
// ticket) scans the block sums, then the local scan.  (Round 1 used three launches; the scan sits on every grouping path —
// voxel x2, Morton order, descriptor buckets, batched depth — and each launch costs ~5 us on a path that is launch-bound
// anyway.  Four items per thread: a quarter of the workgroups, tickets and barriers for the 524k-bucket tables of the voxel
// hash, whose scan took 50 us with one item per thread.)
constexpr int SCAN_IPT = 4;
constexpr int SCAN_BLOCK_ITEMS = 1024 * SCAN_IPT;
__device__ __forceinline__ void scan_load4(const int* __restrict__ in, int n, int i0, int (&x)[SCAN_IPT]) {
    if (i0 + SCAN_IPT <= n && (((size_t)in & 15) == 0)) {
        const int4 q = *reinterpret_cast<const int4*>(in + i0);
        x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < SCAN_IPT; ++e) x[e] = i0 + e < n ? in[i0 + e] : 0;
    }
}
__global__ __launch_bounds__(1024)
void k_scan_reduce(const int* __restrict__ in, int n, int* __restrict__ sums, int* __restrict__ total, unsigned* __restrict__ ticket) {
    __shared__ int w[16];
    __shared__ int carry_s;
    __shared__ bool is_last;
    const int i0 = (blockIdx.x * 1024 + threadIdx.x) * SCAN_IPT;
    int x[SCAN_IPT];
    scan_load4(in, n, i0, x);
    int v = (x[0] + x[1]) + (x[2] + x[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < 16; ++k) s += w[k];
        sums[blockIdx.x] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the sum is visible device-wide before the ticket moves (release only: __threadfence() would also invalidate the XCD's L2 under the workgroups still loading)
        is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        carry_s = 0;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // acquire: the other workgroups' sums
    const int nblocks = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile int* vs = sums;                                    // written by other workgroups of this launch
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        int k = b0 + threadIdx.x;
        int xs = k < nblocks ? vs[k] : 0;
        int incl = xs;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        if (lane == 63) w[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int q = 0; q < wave; ++q) wbase += w[q];
        int carry = carry_s;
        if (k < nblocks) vs[k] = carry + wbase + incl - xs;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *total = carry_s; *ticket = 0u; }   // ready for the next launch (stream order)
}
__global__ __launch_bounds__(1024)
void k_scan_local(const int* __restrict__ in, int n, const int* __restrict__ sums, int* __restrict__ out) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 1024 + threadIdx.x) * SCAN_IPT;
    int x[SCAN_IPT];
    scan_load4(in, n, i0, x);
    const int v = (x[0] + x[1]) + (x[2] + x[3]);
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < wave; ++q) wbase += wsum[q];
    int run = sums[blockIdx.x] + wbase + incl - v;          // exclusive prefix of this thread's first item
#pragma unroll
    for (int e = 0; e < SCAN_IPT; ++e) { if (i0 + e < n) out[i0 + e] = run; run += x[e]; }
}

// one lane per run head: sequential sum over the run (ascending input index), mean -> out[rank]
__global__ __launch_bounds__(256)
void k_voxel_means(const uint4* __restrict__ rec, int n, const float* __restrict__ xyz, const float* __restrict__ rgb,
                   const int* __restrict__ rank, int capacity, float* __restrict__ out_xyz, float* __restrict__ out_rgb) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    uint4 r = rec[p];
    if (p > 0) { uint4 q = rec[p - 1]; if (q.x == r.x && q.y == r.y && q.z == r.z) return; }
    const int slot = rank[r.w];
    if (slot >= capacity) return;
    float ax = 0.f, ay = 0.f, az = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    int cnt = 0;
    for (int e = p; e < n; ++e) {
        uint4 m = rec[e];
        if (!(m.x == r.x && m.y == r.y && m.z == r.z)) break;
        const unsigned idx = m.w;
        ax += xyz[3 * (size_t)idx]; ay += xyz[3 * (size_t)idx + 1]; az += xyz[3 * (size_t)idx + 2];
        if (rgb) { cr += rgb[3 * (size_t)idx]; cg += rgb[3 * (size_t)idx + 1]; cb += rgb[3 * (size_t)idx + 2]; }
        ++cnt;
    }
    const float fn = (float)cnt;
    out_xyz[3 * (size_t)slot] = ax / fn; out_xyz[3 * (size_t)slot + 1] = ay / fn; out_xyz[3 * (size_t)slot + 2] = az / fn;
    if (rgb && out_rgb) { out_rgb[3 * (size_t)slot] = cr / fn; out_rgb[3 * (size_t)slot + 1] = cg / fn; out_rgb[3 * (size_t)slot + 2] = cb / fn; }
}

// out[p] = in[rank[order_first[p]]]
__global__ void k_voxel_permute(const float* __restrict__ in_xyz, const float* __restrict__ in_rgb, const int* __restrict__ rank, int rank_base,
                                const int* __restrict__ order_first, int v, float* __restrict__ out_xyz, float* __restrict__ out_rgb,
                                int* __restrict__ ref2first, int* __restrict__ first2ref) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= v) return;
    int s = rank[order_first[p]] - rank_base;     // (rank_base: the cloud's first voxel when ranks count through several clouds)
    if (ref2first) { ref2first[p] = s; first2ref[s] = p; }
    out_xyz[3 * (size_t)p] = in_xyz[3 * (size_t)s]; out_xyz[3 * (size_t)p + 1] = in_xyz[3 * (size_t)s + 1]; out_xyz[3 * (size_t)p + 2] = in_xyz[3 * (size_t)s + 2];
    if (in_rgb && out_rgb) { out_rgb[3 * (size_t)p] = in_rgb[3 * (size_t)s]; out_rgb[3 * (size_t)p + 1] = in_rgb[3 * (size_t)s + 1]; out_rgb[3 * (size_t)p + 2] = in_rgb[3 * (size_t)s + 2]; }
}

// ---- grouping through a hash table: memset + 2 kernels for any number of clouds (round 3; the default) -------------------
// The counting-sort path above is nine dependent launches whose waves wait >= 90 % of their cycles (profiles/r2: 0.10-0.12 ms
// for a cloud that HBM moves in a microsecond).  This path is three, and takes B clouds stored back to back at once:
//   memset            one fill (0x7f bytes) of [claim table | per-voxel count | scan descriptors | ticket | overflow flag]
//   k_vh_insert       one lane per point: find-or-claim the cell's slot of an open-addressing table (linear probing, 32-bit
//                     atomics).  The slot permanently holds the CLAIMER - the first point that got there; its index names the
//                     voxel: vcnt[claimer] counts the members down from 0x7f7f7f7f, members[claimer][arrival rank] lists them
//                     (VH_K per voxel; a fuller voxel raises a flag and the call is redone on the counting-sort path).  Lanes of
//                     a wave that share a cell go to the table once, through their lowest lane.  Which point claims a slot and
//                     the arrival order depend on the race; nothing that leaves the kernel pair does.
//   k_vh_finalize     one lane per point, 1,024 points per workgroup: every lane reads its voxel's member row; the smallest
//                     index in it is the voxel's leader (an atomic min per point in k_vh_insert did the same for 5 us more).
//                     Leader flags, workgroup scan, DECOUPLED LOOK-BACK over the workgroups' aggregates (single pass: no scan
//                     launches) -> the leader's first-occurrence rank; the leader sums its voxel's points in ascending index
//                     order (registration.cpp:47-50), divides and writes the mean at its rank.  Voxels of cloud b occupy
//                     [voff[b], voff[b + 1]).  Count and overflow flag go straight into pinned host memory (no copy kernel).
// Algorithmic bytes: 12 N in + 12 V out; the table adds 16 B per point of atomics and 4-68 B per voxel of member lists.
constexpr int VH_K = 16;                       // member slots per voxel (one 64-B row)
constexpr int VH_EMPTY = 0x7f7f7f7f;           // what the memset leaves; larger than any point index
constexpr int VH_BLOCK = 1024, VH_IPT = 1, VH_ITEMS = VH_BLOCK * VH_IPT;   // one point per lane: four per lane ran their gather chains one after the other
constexpr unsigned long long VH_ST_MASK = 3ull << 62, VH_ST_WAIT = 1ull << 62, VH_ST_AGG = 2ull << 62, VH_ST_PREFIX = 3ull << 62;   // 0x7f.. has status 01
static_assert((0x7f7f7f7f7f7f7f7full & VH_ST_MASK) == VH_ST_WAIT, "the memset pattern must read as 'not ready'");

__device__ __forceinline__ int vh_segment(const int* __restrict__ seg_off, int nseg, int g) {   // largest b with seg_off[b] <= g (b < nseg)
    int lo = 0, hi = nseg;
    while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (seg_off[m] <= g) lo = m; else hi = m; }
    return lo;
}
__device__ __forceinline__ void vh_cell(const float* __restrict__ xyz, size_t g, float inv, int& cx, int& cy, int& cz) {
    cx = (int)floorf(xyz[3 * g] * inv); cy = (int)floorf(xyz[3 * g + 1] * inv); cz = (int)floorf(xyz[3 * g + 2] * inv);   // registration.cpp:33-36
}

__global__ __launch_bounds__(256)
void k_vh_insert(const float* __restrict__ xyz, int total, const int* __restrict__ seg_off, int nseg, float inv, unsigned mask,
                 int* __restrict__ claim, int* __restrict__ vcnt, int* __restrict__ members,
                 int* __restrict__ voxel_of, int* __restrict__ overflow) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool active = g < total;
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = total, b = 0;
    int cx = 0, cy = 0, cz = 0;
    if (active) {
        if (nseg > 1) { b = vh_segment(seg_off, nseg, g); lo = seg_off[b]; hi = seg_off[b + 1]; }
        vh_cell(xyz, (size_t)g, inv, cx, cy, cz);
    }
    // Device-scope atomics are resolved beyond the XCDs' L2s and cost microseconds each; neighbouring points (one wave holds 64
    // consecutive ones, in image order neighbouring pixels) mostly share their voxels.  The wave therefore groups its lanes by
    // (cloud, cell) first - one pass per distinct cell, the cell of the lowest pending lane broadcast with readlane - and only the
    // lowest lane of a group goes to the table, for the whole group: one find-or-claim, one atomicMin, one atomicSub by the group's
    // size.  Lanes are in index order, so the group's lowest lane holds its smallest index.
    // (a cloud in random order has nothing to group: when no lane shares its cell with the lane before it, every lane goes alone)
    const bool dup = lane > 0 && cx == __shfl_up(cx, 1, 64) && cy == __shfl_up(cy, 1, 64) && cz == __shfl_up(cz, 1, 64) && b == __shfl_up(b, 1, 64);
    unsigned long long todo = __ballot(active), group = 1ull << lane;
    if (!__any(active && dup)) todo = 0ull;
    while (todo) {                                        // wave-uniform
        const int src = __builtin_ctzll(todo);
        const int rx = __builtin_amdgcn_readlane(cx, src), ry = __builtin_amdgcn_readlane(cy, src), rz = __builtin_amdgcn_readlane(cz, src),
                  rb = __builtin_amdgcn_readlane(b, src);
        const bool same = active && cx == rx && cy == ry && cz == rz && b == rb;
        const unsigned long long m = __ballot(same);
        if (same) group = m;
        todo &= ~m;
    }
    if (!active) return;
    const int head = __builtin_ctzll(group);
    const int rank = __popcll(group & ((1ull << lane) - 1ull)), size = __popcll(group);
    int j = 0, pos0 = 0;
    if (lane == head) {
        // (One table for all clouds, the cloud mixed into the hash.  Giving every cloud its OWN stretch of the table - two slots per
        //  point at [2 lo, 2 hi), so that the workgroups in flight touch a few MB instead of 512 - was measured in round 4 and LOST:
        //  k_vh_insert 2.2 -> 3.15 ms for the 256-cloud batch, 20 -> 39 us for one cloud.  The device-scope atomics of the ~2,000
        //  workgroups in flight then land on a few memory channels instead of all of them; spreading them is worth more than locality.)
        unsigned h = voxel_hash((unsigned)cx, (unsigned)cy, (unsigned)cz) + (unsigned)b * 0x9e3779b9u;
        for (;; ++h) {
            int* slot = claim + (h & mask);
            j = *slot;                                                  // (a plain load first: claiming with the CAS straight away - one round trip for the four
            if (j == VH_EMPTY) { const int prev = atomicCAS(slot, VH_EMPTY, g); j = prev == VH_EMPTY ? g : prev; }   // of five groups that find their slot empty - cost a batch 30 % more: a device-scope atomic is dearer than a load + an atomic that mostly succeeds)
            if (j == g) break;
            if (j >= lo && j < hi) {                        // the same cloud: same cell?
                int qx, qy, qz;
                vh_cell(xyz, (size_t)j, inv, qx, qy, qz);
                if (qx == cx && qy == cy && qz == cz) break;
            }
        }
        // The claimer is a member by being the claimer: it neither counts itself nor writes itself into the row (round 4).  Four of five
        // voxels of a depth-camera cloud hold one point: for them the insertion is now the load + the claiming CAS and nothing else
        // (it was: + one more device-scope atomic and a scattered 4-byte store into a 64-byte row nobody reads).
        const int others = j == g ? size - 1 : size;
        if (others > 0) pos0 = VH_EMPTY - atomicSub(&vcnt[j], others);       // arrival ranks of the non-claimers pos0 .. pos0 + others - 1
    }
    const int g_head = __shfl(g, head, 64);
    j = __shfl(j, head, 64); pos0 = __shfl(pos0, head, 64);
    voxel_of[g] = j;
    if (j == g) return;                                  // the claimer itself
    const int pos = pos0 + (j == g_head ? rank - 1 : rank);
    if (pos < VH_K - 1) members[(size_t)j * VH_K + pos] = g;
    else *overflow = 1;                                 // (any value but the fill pattern)
}

// ---- grouping WITHOUT the table, for clouds that come from a pinhole depth image (round 4; the batch's clouds) ---------------
// k_vh_insert is bound by the chip's rate of device-scope atomics (20 G/s: profiles/r4/history/voxel_batch.md).  A cloud
// unprojected from a depth image needs none: its points are in row-major pixel order, and all members of a voxel lie within a
// few pixels of each other - two points of one voxel differ by less than the voxel size s in x, y and z, hence by
//     |du| <= fx * (s / z) * (1 + |x / z|)        (u - cx = fx * x / z;  d(x / z) <= dx / z + |x / z| * dz / z)
// pixels in u, the same with fy in v.  So one workgroup takes a TILE of consecutive points of one cloud plus a HALO of whole
// rows before and after it, recovers every point's pixel (u = rint(cx + fx * x / z): the rounding error of the unprojection is
// 4e-4 pixel), enters the points into a pixel map in LDS, and every point looks its voxel's members up in the (2 win_v + 1) x
// (2 win_u + 1) window around its own pixel - 3 x 3 at the pipeline's 1.2-pixel voxels.  The window walk is in row-major order =
// ascending input index: the smallest member index is the voxel's LEADER, and the leader can add its voxel's points up in the
// reference's order right there.  The kernel writes voxel_of[g] = the leader (what k_vh_insert writes, with the leader as claimer)
// and the voxel's mean at the leader's index; k_vh_finalize then only ranks the leaders and moves the means to their ranks - no
// member rows, no counts, no limit on the members of a voxel, and nothing depends on a race.
// Anything the argument does not cover makes the tile raise *fail (the caller redoes the call through the table): a window
// above VS_WIN_MAX (coarse voxels), rows too long for the halo, a pixel map or a key range that does not fit, points that are not
// in row-major pixel order (the cloud is not what the caller said it was), non-positive depth.
struct VoxelPinhole { float fx, fy, cx, cy; };
#ifndef VS_TILE_VALUE
#define VS_TILE_VALUE 4096
#endif
#ifndef VS_HALO_VALUE
#define VS_HALO_VALUE 2048
#endif
#ifndef VS_MAP_VALUE
#define VS_MAP_VALUE 20480
#endif
constexpr int VS_TILE = VS_TILE_VALUE, VS_HALO = VS_HALO_VALUE, VS_LOAD = VS_TILE + 2 * VS_HALO, VS_BLOCK = 1024, VS_PER = VS_LOAD / VS_BLOCK;
constexpr int VS_MAP = VS_MAP_VALUE;          // pixel map entries (u16: local index + 1)
constexpr int VS_WIN_MAX = 3;
constexpr int VS_FAILED = 2;           // value of the overflow word when a tile could not be grouped this way (written with atomicMin; the word starts as VH_EMPTY)

// LDS per loaded point: its cell relative to the tile's first point, packed into 32 bits (10 + 8 + 14: a tile spans a few hundred cells in
// x, a few dozen in y; a range that does not fit raises *fail), and its map entry: 73 KB per workgroup, two workgroups per CU.  (The
// first version kept three 16-bit cells and the pixels - 129 KB, one workgroup per CU: 1.39 ms for the 256-cloud batch; 16-bit tags
// with the neighbour's cell recomputed from memory on a match: 1.08 ms - the dependent loads of the checks, one after the other.)
constexpr int VS_KX = 10, VS_KY = 8, VS_KZ = 14;

__global__ __launch_bounds__(VS_BLOCK, 2)
void k_vs_group(const float* __restrict__ xyz, const int* __restrict__ seg_off, const int2* __restrict__ tiles /* (first owned point, cloud) */,
                float inv, float voxel, VoxelPinhole cam, float* __restrict__ mean_at /* [point][3]: a leader's voxel mean, at the leader's index */,
                int* __restrict__ voxel_of, int* __restrict__ overflow) {
    __shared__ unsigned key[VS_LOAD];
    __shared__ unsigned short map[VS_MAP];
    __shared__ unsigned edge[VS_PER][VS_BLOCK / 64];                      // pixel of every wave's last lane, per round (the order check across waves)
    __shared__ int s_umin, s_umax, s_vlast, s_vend, s_bad;
    __shared__ unsigned s_zmin, s_amax, s_bmax;
    const int2 tl = tiles[blockIdx.x];
    const int lo = seg_off[tl.y], hi = seg_off[tl.y + 1];
    const int g0 = tl.x, g1 = min(g0 + VS_TILE, hi);
    const int l0 = max(lo, g0 - VS_HALO), l1 = min(hi, g1 + VS_HALO), nl = l1 - l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_umin = INT_MAX; s_umax = INT_MIN; s_vlast = 0; s_vend = 0; s_bad = 0; s_zmin = 0x7f800000u; s_amax = 0u; s_bmax = 0u; }
    for (int e = threadIdx.x; e < VS_MAP / 2; e += VS_BLOCK) reinterpret_cast<unsigned*>(map)[e] = 0u;
    __syncthreads();
    // pass 1: pixel and tag of every loaded point (thread t holds points t, t + 1024, ...: the owned ones among them are its own in pass 3)
    unsigned pvu[VS_PER], pk[VS_PER];                                     // v << 16 | u; the cell relative to the first loaded point's, biased to the middle of its bit field
    int ox, oy, oz;
    vh_cell(xyz, (size_t)l0, inv, ox, oy, oz);
    ox -= 1 << (VS_KX - 1); oy -= 1 << (VS_KY - 1); oz -= 1 << (VS_KZ - 1);
    int umin = INT_MAX, umax = INT_MIN;
    float zmin = INFINITY, amax = 0.f, bmax = 0.f; bool bad = false;
#pragma unroll
    for (int q = 0; q < VS_PER; ++q) {
        const int i = (int)threadIdx.x + q * VS_BLOCK;
        pvu[q] = 0u; pk[q] = 0u;
        if (i < nl) {
            const size_t g = (size_t)(l0 + i);
            const float x = xyz[3 * g], y = xyz[3 * g + 1], z = xyz[3 * g + 2];
            int cx, cy, cz;
            vh_cell(xyz, g, inv, cx, cy, cz);
            const long long rx = (long long)cx - ox, ry = (long long)cy - oy, rq = (long long)cz - oz;
            if (rx < 0 || rx >= (1 << VS_KX) || ry < 0 || ry >= (1 << VS_KY) || rq < 0 || rq >= (1 << VS_KZ)) bad = true;     // a tile wider than the packed cell: the table
            pk[q] = (unsigned)rx | ((unsigned)ry << VS_KX) | ((unsigned)rq << (VS_KX + VS_KY));
            key[i] = pk[q];
            const float rz = __frcp_rn(z);                               // (a reciprocal is plenty: the pixel is recovered to 1e-3, rint needs 0.5)
            const float a = x * rz, b = y * rz;
            const float fu = rintf(cam.cx + cam.fx * a), fv = rintf(cam.cy + cam.fy * b);
            if (!(z > 0.f) || !(fu >= 0.f && fu < 65535.f && fv >= 0.f && fv < 65535.f)) bad = true;
            else {
                const int u = (int)fu, v = (int)fv;
                pvu[q] = ((unsigned)v << 16) | (unsigned)u;
                umin = min(umin, u); umax = max(umax, u);
                zmin = fminf(zmin, z); amax = fmaxf(amax, fabsf(a)); bmax = fmaxf(bmax, fabsf(b));
                if (i == g1 - 1 - l0) s_vlast = v;
                if (i == nl - 1) s_vend = v;
            }
        }
        if (lane == 63) edge[q][wave] = pvu[q];
    }
    // (one LDS atomic per WAVE and value: an atomic per thread on a handful of addresses took longer than everything else here)
    unsigned zb = __float_as_uint(zmin), ab = __float_as_uint(amax), bb = __float_as_uint(bmax);      // (non-negative floats order like their bits)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        umin = min(umin, __shfl_xor(umin, o, 64)); umax = max(umax, __shfl_xor(umax, o, 64));
        zb = min(zb, (unsigned)__shfl_xor((int)zb, o, 64)); ab = max(ab, (unsigned)__shfl_xor((int)ab, o, 64)); bb = max(bb, (unsigned)__shfl_xor((int)bb, o, 64));
    }
    if (lane == 0) { atomicMin(&s_umin, umin); atomicMax(&s_umax, umax); atomicMin(&s_zmin, zb); atomicMax(&s_amax, ab); atomicMax(&s_bmax, bb); }
    if (bad) s_bad = 1;
    // row-major order inside the thread's rounds is checked against the lane before (the point before in memory); lane 0 looks at the wave before
    bool viol = false;
#pragma unroll
    for (int q = 0; q < VS_PER; ++q) {
        const unsigned prev = __shfl_up(pvu[q], 1, 64);
        const int i = (int)threadIdx.x + q * VS_BLOCK;
        if (lane > 0 && i < nl && !(pvu[q] > prev)) viol = true;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < VS_PER; ++q) {
        const int i = (int)threadIdx.x + q * VS_BLOCK;
        if (lane == 0 && i > 0 && i < nl) {
            const unsigned prev = wave > 0 ? edge[q][wave - 1] : edge[q > 0 ? q - 1 : 0][VS_BLOCK / 64 - 1];
            if (!(pvu[q] > prev)) viol = true;
        }
    }
    const int u0 = s_umin, ur = s_umax - s_umin + 1;
    const int qo = (g0 - l0) / VS_BLOCK;                                  // the thread's first owned round (g0 - l0 is 0 or VS_HALO: a multiple of the block)
    const int v0 = (int)(__shfl(pvu[0], 0, 64) >> 16);                    // (wave 0 only: the first loaded point's row) - broadcast below
    __shared__ int s_v0, s_vfirst;
    if (threadIdx.x == 0) { s_v0 = v0; s_vfirst = (int)((qo == 0 ? pvu[0] : pvu[VS_HALO / VS_BLOCK < VS_PER ? VS_HALO / VS_BLOCK : 0]) >> 16); }
    __syncthreads();
    const int vbase = s_v0, vr = s_vend - s_v0 + 1;
    const float z_min = __uint_as_float(s_zmin);
    // how far apart (in pixels) two members of one voxel can be: the bound of the header, for points that need not be among the loaded
    // ones (a member shares a voxel with a loaded point: its z is at most a voxel smaller, its |x| at most a voxel larger)
    const float z_eff = z_min - voxel, rze = 1.f / z_eff;
    const float reach = voxel * rze * 1.001f;
    const int win_u = (int)floorf(cam.fx * reach * (1.f + (__uint_as_float(s_amax) * z_min + voxel) * rze) + 0.01f);
    const int win_v = (int)floorf(cam.fy * reach * (1.f + (__uint_as_float(s_bmax) * z_min + voxel) * rze) + 0.01f);
    bool ok = !s_bad && nl > 0 && z_eff > 0.f && vr > 0 && (long long)ur * vr <= VS_MAP && win_u <= VS_WIN_MAX && win_v <= VS_WIN_MAX;
    // a halo of whole rows that covers the windows (the first and the last loaded row may be cut: they must lie outside every window)
    if (ok && threadIdx.x == 0) {
        if (l0 > lo && !(vbase < s_vfirst - win_v)) viol = true;
        if (l1 < hi && !(s_vend > s_vlast + win_v)) viol = true;
    }
    if (__syncthreads_or(viol || !ok)) { if (threadIdx.x == 0) atomicMin(overflow, VS_FAILED); return; }
    // pass 2: the pixel map
#pragma unroll
    for (int q = 0; q < VS_PER; ++q) {
        const int i = (int)threadIdx.x + q * VS_BLOCK;
        if (i < nl) map[((int)(pvu[q] >> 16) - vbase) * ur + ((int)(pvu[q] & 0xffffu) - u0)] = (unsigned short)(i + 1);
    }
    __syncthreads();
    // pass 3: every owned point walks its window in row-major order = ascending input index
#pragma unroll
    for (int q = 0; q < VS_PER; ++q) {
        const int li = (int)threadIdx.x + q * VS_BLOCK;
        if (q < qo || li < g0 - l0 || li >= g1 - l0) continue;
        const size_t gp = (size_t)(l0 + li);
        const unsigned mk = pk[q];
        // this point's own window: the bound with ITS depth and tangents - never larger than the tile's, which the halo was checked against
        const float pz = xyz[3 * gp + 2], prz = __frcp_rn(pz - voxel), pa = fabsf(xyz[3 * gp]) * prz, pb = fabsf(xyz[3 * gp + 1]) * prz;
        const float r0 = voxel * prz * 1.001f;
        const int wu = min(win_u, (int)floorf(cam.fx * r0 * (1.f + pa + r0) + 0.01f)), wv = min(win_v, (int)floorf(cam.fy * r0 * (1.f + pb + r0) + 0.01f));
        const int pu_ = (int)(pvu[q] & 0xffffu) - u0, pv_ = (int)(pvu[q] >> 16) - vbase;
        const int ua = max(pu_ - wu, 0), ub = min(pu_ + wu, ur - 1), va = max(pv_ - wv, 0), vb = min(pv_ + wv, vr - 1);
        auto member = [&](int m) -> bool { return m >= 0 && m != li && key[m] == mk; };
        int leader = li, others = 0;
        for (int vv = va; vv <= vb; ++vv)
            for (int uu = ua; uu <= ub; ++uu) {
                const int m = (int)map[vv * ur + uu] - 1;
                if (member(m)) { ++others; leader = min(leader, m); }
            }
        const int g = l0 + li;
        voxel_of[g] = l0 + leader;
        if (leader == li) {
            // the leader sums its voxel: itself, then the others as the window walk meets them - ascending input index, the order of
            // registration.cpp:47-50 (the sum starts at 0, as there: a lone -0 becomes +0) - and divides by the count (:52-53)
            float ax = 0.f, ay = 0.f, az = 0.f;
            ax += xyz[3 * gp]; ay += xyz[3 * gp + 1]; az += pz;
            if (others > 0)
                for (int vv = va; vv <= vb; ++vv)
                    for (int uu = ua; uu <= ub; ++uu) {
                        const int m = (int)map[vv * ur + uu] - 1;
                        if (member(m)) { const size_t gm = (size_t)(l0 + m); ax += xyz[3 * gm]; ay += xyz[3 * gm + 1]; az += xyz[3 * gm + 2]; }
                    }
            const float fn = (float)(others + 1);
            mean_at[3 * gp] = ax / fn; mean_at[3 * gp + 1] = ay / fn; mean_at[3 * gp + 2] = az / fn;
        }
    }
}

// MODE 0: single pass (tiles numbered by ticket, decoupled look-back).  MODE 1 / 2: the same split in two for large inputs - count
// the leaders per tile (tile_sums), [exclusive scan], emit with the scanned prefix (tile_prefix): with tens of thousands of tiles the
// look-back's polling (device-scope loads in a loop from every tile in flight) slows the whole memory system down - 61 us of waiting
// per tile and 51 us for a leader wave's gathers in a batch of 256 clouds, against 9 and 18 us for one cloud.
template <int MODE>
__global__ __launch_bounds__(VH_BLOCK)
void k_vh_finalize(const float* __restrict__ xyz, const float* __restrict__ rgb, int total, const int* __restrict__ seg_off, int nseg, float inv,
                   const int* __restrict__ voxel_of, const int* __restrict__ vcnt, const int* __restrict__ members,
                   unsigned long long* __restrict__ desc, int* __restrict__ ticket, float* __restrict__ out_xyz, float* __restrict__ out_rgb,
                   int* __restrict__ rank_out /* per point: global first-occurrence rank of leaders */, int4* __restrict__ leaders /* optional */,
                   int* __restrict__ voff /* nseg + 2: the last entry receives the overflow flag */, int capacity /* voxels that fit out_xyz */,
                   const int* __restrict__ overflow_flag, int* __restrict__ host_result /* optional, pinned host memory: {voxels, overflow flag} */,
                   int* __restrict__ tile_sums /* MODE 1 */, const int* __restrict__ tile_prefix /* MODE 2 */,
                   int leader_known /* voxel_of[g] IS the voxel's smallest index (k_vs_group): only the leaders look at counts and rows */,
                   const float* __restrict__ mean_at /* with leader_known: the leaders' means are ready, at their own indices */) {
    __shared__ int s_ticket, s_excl, s_wave[VH_BLOCK / 64];
    if (*overflow_flag == VS_FAILED) {          // k_vs_group gave up on a tile: voxel_of is not valid, nothing here may follow it - the caller redoes the call
        if (blockIdx.x == 0 && threadIdx.x == 0) { voff[nseg + 1] = VS_FAILED; if (host_result) { host_result[0] = 0; host_result[1] = VS_FAILED; __threadfence_system(); } }
        return;
    }
    if (MODE == 0) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1) - VH_EMPTY;     // workgroups take their tiles in the order they start
        __syncthreads();
    }
    const int t = MODE == 0 ? s_ticket : (int)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = t * VH_ITEMS + threadIdx.x;
    const bool inside = g < total;
    // this point's voxel: its member row (the indices of its points in arrival order).  The smallest is the voxel's leader.
    const int j = inside ? voxel_of[g] : 0;
    // members = the claimer j + the `others` that found j's slot taken (their indices in j's row, arrival order)
    const bool look = inside && !leader_known;
    const int others = look ? min(VH_EMPTY - vcnt[j], VH_K - 1) : 0;     // (an overflowing voxel: the call is redone, whatever is written here is dropped)
    const int cnt = inside ? others + 1 : 0;
    int m[VH_K];
    {
        const int4* row = reinterpret_cast<const int4*>(members + (size_t)j * VH_K);
#pragma unroll
        for (int q = 0; q < VH_K / 4; ++q) {
            int4 r = make_int4(VH_EMPTY, VH_EMPTY, VH_EMPTY, VH_EMPTY);
            if (4 * q < others) r = row[q];                  // (a voxel of one point - four of five in a depth-camera cloud - is that point: its row is never fetched)
            m[4 * q] = 4 * q < others ? r.x : VH_EMPTY; m[4 * q + 1] = 4 * q + 1 < others ? r.y : VH_EMPTY;
            m[4 * q + 2] = 4 * q + 2 < others ? r.z : VH_EMPTY; m[4 * q + 3] = 4 * q + 3 < others ? r.w : VH_EMPTY;
        }
        m[VH_K - 1] = inside ? j : VH_EMPTY;                 // (others <= VH_K - 1: the last entry of the row is never used)
    }
    int first_member = VH_EMPTY;
#pragma unroll
    for (int q = 0; q < VH_K; ++q) first_member = min(first_member, m[q]);
    const bool lead = inside && (leader_known ? j == g : first_member == g);
    const int mine = lead ? 1 : 0;
    // workgroup scan of the leader flags
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wbase = 0, agg = 0;
#pragma unroll
    for (int w = 0; w < VH_BLOCK / 64; ++w) { if (w < wave) wbase += s_wave[w]; agg += s_wave[w]; }
    if (MODE == 1) { if (threadIdx.x == 0) tile_sums[t] = agg; return; }
    if (MODE == 2) { if (threadIdx.x == 0) s_excl = tile_prefix[t]; }
    // decoupled look-back (wave 0): publish the aggregate, then add up the predecessors' aggregates back to the nearest
    // inclusive prefix, 64 tiles per step.  Tiles are numbered by ticket, so every predecessor has started and publishes its
    // aggregate without waiting for anybody: the wait below always ends.
    if (MODE == 0 && wave == 0) {
        if (lane == 0) __hip_atomic_store(&desc[t], (t == 0 ? VH_ST_PREFIX : VH_ST_AGG) | (unsigned long long)(unsigned)agg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int excl = 0;
        for (int look = t - 1; look >= 0; look -= 64) {
            const int idx = look - lane;
            unsigned long long d = VH_ST_PREFIX;                                  // tiles before the first: prefix 0
            if (idx >= 0) {
                d = __hip_atomic_load(&desc[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                while ((d & VH_ST_MASK) == VH_ST_WAIT) { __builtin_amdgcn_s_sleep(1); d = __hip_atomic_load(&desc[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
            }
            const unsigned long long pm = __ballot((d & VH_ST_MASK) == VH_ST_PREFIX);
            const int first = pm ? __builtin_ctzll(pm) : 64;                       // nearest tile that already holds an inclusive prefix
            int v = lane <= first ? (int)(unsigned)(d & 0xffffffffull) : 0;
            v = wave_sum_i32(v);
            excl += v;
            if (pm) break;
        }
        if (lane == 0) {
            if (t > 0) __hip_atomic_store(&desc[t], VH_ST_PREFIX | (unsigned long long)(unsigned)(excl + agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    const int run = s_excl + wbase + incl - mine;        // global first-occurrence rank (of this lane's voxel if it leads one)
    const bool last_tile = (t + 1) * VH_ITEMS >= total;
    if (last_tile && threadIdx.x == VH_BLOCK - 1) {      // the grand total, and every cloud that starts at the very end (empty trailing clouds)
        const int total_v = s_excl + agg, ovf = *overflow_flag;   // (the flag was set by k_vh_insert, the launch before)
        voff[nseg] = total_v;
        voff[nseg + 1] = ovf;
        if (nseg > 1) for (int b = nseg - 1; b >= 0 && seg_off[b] >= total; --b) voff[b] = total_v;
        if (host_result) { host_result[0] = total_v; host_result[1] = ovf; __threadfence_system(); }   // straight into pinned host memory: no copy kernel
    }
    if (!inside) return;
    int lo = 0, b = 0;
    if (nseg > 1) { b = vh_segment(seg_off, nseg, g); lo = seg_off[b]; }
    if (g == lo) {                                        // first point of cloud b: its voxels start at this rank; so do the empty clouds right before it
        voff[b] = run;
        for (int bb = b - 1; bb >= 0 && seg_off[bb] == g; --bb) voff[bb] = run;
    }
    if (!lead) return;
    if (leader_known) {                                   // k_vs_group summed the voxel already: move its mean to its rank
        const size_t o = (size_t)run;
        if (rank_out) rank_out[g] = run;
        if (run >= capacity) return;
        out_xyz[3 * o] = mean_at[3 * (size_t)g]; out_xyz[3 * o + 1] = mean_at[3 * (size_t)g + 1]; out_xyz[3 * o + 2] = mean_at[3 * (size_t)g + 2];
        if (leaders) { int cx, cy, cz; vh_cell(xyz, (size_t)g, inv, cx, cy, cz); leaders[o] = make_int4(cx, cy, cz, g - lo); }
        return;
    }
    // ascending index order without moving anything: cnt rounds of "smallest index above the last one"
    float ax = 0.f, ay = 0.f, az = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    int last = -1;
    for (int r = 0; r < cnt; ++r) {
        int nxt = VH_EMPTY;
#pragma unroll
        for (int q = 0; q < VH_K; ++q) nxt = (m[q] > last && m[q] < nxt) ? m[q] : nxt;
        last = nxt;
        const size_t idx = (size_t)nxt;
        ax += xyz[3 * idx]; ay += xyz[3 * idx + 1]; az += xyz[3 * idx + 2];        // registration.cpp:47-50, ascending input index
        if (rgb) { cr += rgb[3 * idx]; cg += rgb[3 * idx + 1]; cb += rgb[3 * idx + 2]; }
    }
    const float fn = (float)cnt;
    const size_t o = (size_t)run;
    if (rank_out) rank_out[g] = run;
    if (run >= capacity) return;                          // the caller's buffer is too small: it learns the count and gets an error
    out_xyz[3 * o] = ax / fn; out_xyz[3 * o + 1] = ay / fn; out_xyz[3 * o + 2] = az / fn;   // :52-53
    if (rgb && out_rgb) { out_rgb[3 * o] = cr / fn; out_rgb[3 * o + 1] = cg / fn; out_rgb[3 * o + 2] = cb / fn; }
    if (leaders) { int cx, cy, cz; vh_cell(xyz, (size_t)g, inv, cx, cy, cz); leaders[o] = make_int4(cx, cy, cz, g - lo); }
}

// The hash-table path for B clouds stored back to back (d_seg_off: B + 1 device ints, nullptr for one cloud).  Outputs in
// first-occurrence order, cloud b at [voff[b], voff[b + 1]) of d_out_xyz (capacity `total` points); d_voff: B + 1 ints.
// *overflowed = 1: a voxel had more than VH_K members (the outputs are garbage, take the counting-sort path).
static int voxel_hash_first_order(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int total, const int* d_seg_off, int nseg, float voxel,
                                  float* d_out_xyz, float* d_out_rgb, int capacity, int* d_rank /* optional, total ints */,
                                  int4* d_leaders /* optional, capacity entries */, int* d_voff, int* h_result /* optional: pinned {voxels, overflow flag} */,
                                  const VoxelPinhole* pinhole = nullptr, const int* h_seg_off = nullptr /* with pinhole: nseg + 1 host offsets */) {
    hipStream_t s = ctx->stream;
    const float inv = 1.0f / voxel;  // registration.cpp:32
    const bool grouped_by_pixels = pinhole && h_seg_off && d_seg_off;      // k_vs_group instead of the table (the caller redoes the call without it if a tile fails)
    const int tiles = (total + VH_ITEMS - 1) / VH_ITEMS;
    // one fill: claim[slots] | vcnt[total] | desc[tiles] (u64) | ticket | overflow.  Grouped by pixels: no table, no counts, no member rows -
    // only the scan descriptors, the ticket and the overflow word are filled, and the leaders' means take 12 B per point
    size_t slots = 0, n_cnt = 0;
    if (!grouped_by_pixels) { slots = 4096; while (slots < 2 * (size_t)total) slots <<= 1; n_cnt = (size_t)total; }
    const size_t desc_at = (slots + n_cnt + 1) & ~(size_t)1;
    const size_t n_fill = desc_at + 2 * (size_t)tiles + 2;
    int* fill;
    TDV_TRY(ws_alloc(ctx, n_fill, &fill));
    int* claim = fill; int* vcnt = claim + slots;
    unsigned long long* desc = reinterpret_cast<unsigned long long*>(fill + desc_at);
    int* ticket = reinterpret_cast<int*>(desc + tiles);
    int* overflow = ticket + 1;
    int *members = nullptr, *voxel_of; float* mean_at = nullptr;
    if (grouped_by_pixels) TDV_TRY(ws_alloc(ctx, (size_t)total * 3, &mean_at));
    else TDV_TRY(ws_alloc(ctx, (size_t)total * VH_K, &members));
    TDV_TRY(ws_alloc(ctx, (size_t)total, &voxel_of));
    if (grouped_by_pixels) {
        // tiles: consecutive VS_TILE-point stretches of every cloud (a tile never spans two clouds)
        size_t nt = 0;
        for (int b = 0; b < nseg; ++b) nt += (size_t)((h_seg_off[b + 1] - h_seg_off[b] + VS_TILE - 1) / VS_TILE);
        int2* d_tiles;
        TDV_TRY(ws_alloc(ctx, nt, &d_tiles));
        TDV_TRY(pin_reserve(ctx, nt * sizeof(int2) + 64 + ((size_t)nseg + 2) * 4));   // (+ what the caller stages afterwards: the buffer must not move while the copy below is in flight)
        int2* h_tiles = reinterpret_cast<int2*>(ctx->pin);
        size_t k = 0;
        for (int b = 0; b < nseg; ++b)
            for (int g = h_seg_off[b]; g < h_seg_off[b + 1]; g += VS_TILE) h_tiles[k++] = make_int2(g, b);
        TDV_HIP(ctx, hipMemcpyAsync(d_tiles, h_tiles, nt * sizeof(int2), hipMemcpyHostToDevice, s));
        TDV_HIP(ctx, hipMemsetAsync(fill, 0x7f, n_fill * 4, s));
        k_vs_group<<<(unsigned)nt, VS_BLOCK, 0, s>>>(d_xyz, d_seg_off, d_tiles, inv, voxel, *pinhole, mean_at, voxel_of, overflow);
    } else {
        TDV_HIP(ctx, hipMemsetAsync(fill, 0x7f, n_fill * 4, s));
        k_vh_insert<<<(total + 255) / 256, 256, 0, s>>>(d_xyz, total, d_seg_off, nseg, inv, (unsigned)(slots - 1), claim, vcnt, members, voxel_of, overflow);
    }
    // (measured, us per call one pass / split: 184k points in pixel order 65 / 76; random order 250k 82 / 74, 500k 138 / 103, 1M 233 / 160, 4M 858 / 550;
    //  256 clouds of 184k 8,100 / 4,400 - tools/studies/voxel_split_threshold.py)
    static const int split_from = study_env("TDV_VOXEL_SPLIT_FROM") ? atoi(study_env("TDV_VOXEL_SPLIT_FROM")) : 224;   // tiles; tuning knob
    if (tiles < split_from)
        k_vh_finalize<0><<<tiles, VH_BLOCK, 0, s>>>(d_xyz, d_rgb, total, d_seg_off, nseg, inv, voxel_of, vcnt, members, desc, ticket,
                                                    d_out_xyz, d_out_rgb, d_rank, d_leaders, d_voff, capacity, overflow, h_result, nullptr, nullptr, grouped_by_pixels ? 1 : 0, mean_at);
    else {      // large inputs (a batch): count, scan, emit - launches without a wait inside
        int *tile_sums, *tile_prefix, *d_tot;
        TDV_TRY(ws_alloc(ctx, (size_t)tiles, &tile_sums));
        TDV_TRY(ws_alloc(ctx, (size_t)tiles, &tile_prefix));
        TDV_TRY(ws_alloc(ctx, 1, &d_tot));
        k_vh_finalize<1><<<tiles, VH_BLOCK, 0, s>>>(d_xyz, d_rgb, total, d_seg_off, nseg, inv, voxel_of, vcnt, members, desc, ticket,
                                                    d_out_xyz, d_out_rgb, d_rank, d_leaders, d_voff, capacity, overflow, h_result, tile_sums, nullptr, grouped_by_pixels ? 1 : 0, mean_at);
        TDV_TRY(exclusive_scan_dev(ctx, tile_sums, tiles, tile_prefix, d_tot));
        k_vh_finalize<2><<<tiles, VH_BLOCK, 0, s>>>(d_xyz, d_rgb, total, d_seg_off, nseg, inv, voxel_of, vcnt, members, desc, ticket,
                                                    d_out_xyz, d_out_rgb, d_rank, d_leaders, d_voff, capacity, overflow, h_result, nullptr, tile_prefix, grouped_by_pixels ? 1 : 0, mean_at);
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

// out[p] = in[idx[p]] for rows of three floats
__global__ void k_gather_rows3(const float* __restrict__ in, const int* __restrict__ idx, int n, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t q = (size_t)idx[p];
    out[3 * (size_t)p] = in[3 * q]; out[3 * (size_t)p + 1] = in[3 * q + 1]; out[3 * (size_t)p + 2] = in[3 * q + 2];
}

namespace {
struct VoxelKey {
    int x, y, z;
    bool operator==(const VoxelKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelKeyHash {  // the reference's combiner (registration.cpp:20-27)
    size_t operator()(const VoxelKey& k) const {
        size_t h = std::hash<int>()(k.x);
        h ^= std::hash<int>()(k.y) + 0x9e3779b9 + (h << 6) + (h >> 2);
        h ^= std::hash<int>()(k.z) + 0x9e3779b9 + (h << 6) + (h >> 2);
        return h;
    }
};

// The emulation below is tied to libstdc++'s <bits/hashtable.h> (it instantiates the library's own _Prime_rehash_policy):
// with any other standard library the real std::unordered_map is used instead (the TDV_VOXEL_REAL_MAP path), which is
// that library's order by construction.  tests/test_gpu_voxel.py holds the emulation against the real container, so a
// toolchain bump that changes the node-list manipulation is caught.
#ifdef __GLIBCXX__
#define TDV_HAVE_LIBSTDCXX_EMULATION 1
// Iteration order of a libstdc++ std::unordered_map (unique keys, std::__detail::_Mod_range_hashing, the library's own
// _Prime_rehash_policy object) after inserting DISTINCT keys in a given sequence — the same node-list manipulation as
// _Hashtable::_M_insert_unique_node / _M_insert_bucket_begin / _M_rehash_aux(unique) of <bits/hashtable.h>, on index
// arrays instead of heap nodes.  Only first occurrences change a container, so replaying the leaders (first point of
// every voxel, in input order) reproduces the reference's voxel order exactly, at a fraction of the cost of building
// the real map over all points; tests compare it with the real container.
class LibstdcxxInsertionOrder {
public:
    explicit LibstdcxxInsertionOrder(size_t expected) { next_.reserve(expected); code_.reserve(expected); buckets_.assign(1, EMPTY); }
    void insert(size_t code) {   // a key not inserted before
        const int node = (int)next_.size();
        next_.push_back(NONE); code_.push_back(code);
        const auto saved = policy_._M_state();
        const std::pair<bool, std::size_t> rh = policy_._M_need_rehash(buckets_.size(), (size_t)node, 1);
        if (rh.first) { try { rehash(rh.second); } catch (...) { policy_._M_reset(saved); throw; } }
        const size_t bkt = code % buckets_.size();
        if (buckets_[bkt] != EMPTY) {          // after the bucket's "before" node: the new node becomes the bucket's first
            int& after = link(buckets_[bkt]);
            next_[node] = after; after = node;
        } else {                               // at the very beginning of the list
            next_[node] = head_; head_ = node;
            if (next_[node] != NONE) buckets_[code_[next_[node]] % buckets_.size()] = node;
            buckets_[bkt] = BEFORE_BEGIN;
        }
    }
    template <class F> void for_each(F&& f) const { for (int p = head_; p != NONE; p = next_[p]) f(p); }   // p = insertion rank
private:
    static constexpr int NONE = -1, EMPTY = -1, BEFORE_BEGIN = -2;
    int& link(int before) { return before == BEFORE_BEGIN ? head_ : next_[before]; }
    void rehash(size_t n) {
        std::vector<int> nb(n, EMPTY);
        int p = head_; head_ = NONE;
        size_t bbegin_bkt = 0;
        while (p != NONE) {
            const int nxt = next_[p];
            const size_t bkt = code_[p] % n;
            if (nb[bkt] == EMPTY) {
                next_[p] = head_; head_ = p;
                nb[bkt] = BEFORE_BEGIN;
                if (next_[p] != NONE) nb[bbegin_bkt] = p;
                bbegin_bkt = bkt;
            } else {
                int& after = nb[bkt] == BEFORE_BEGIN ? head_ : next_[nb[bkt]];
                next_[p] = after; after = p;
            }
            p = nxt;
        }
        buckets_.swap(nb);
    }
    std::vector<int> next_; std::vector<size_t> code_; std::vector<int> buckets_;
    int head_ = NONE;
    std::__detail::_Prime_rehash_policy policy_;
};
#else
#define TDV_HAVE_LIBSTDCXX_EMULATION 0
#endif
}  // namespace

// exclusive scan of n ints on the ctx stream; *d_total receives the sum (device pointer)
int exclusive_scan_dev(tdv_ctx* ctx, const int* d_in, int n, int* d_out, int* d_total) {
    if (n <= 0) return TDV_OK;
    const int sblocks = (n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS;
    int* sums;
    TDV_TRY(ws_alloc(ctx, (size_t)sblocks, &sums));
    hipStream_t s = ctx->stream;
    k_scan_reduce<<<sblocks, 1024, 0, s>>>(d_in, n, sums, d_total, ctx->scan_ticket);
    k_scan_local<<<sblocks, 1024, 0, s>>>(d_in, n, sums, d_out);
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

size_t sort_pow2(size_t n) { size_t p = BT_TILE; while (p < n) p <<= 1; return p; }

static int voxel_downsample_impl(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int n, float voxel, int order,
                                 float* d_out_xyz, float* d_out_rgb, int capacity, int* n_out, bool full_sort, const VoxelBothOrders* both);

int voxel_downsample_dev(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int n, float voxel, int order,
                         float* d_out_xyz, float* d_out_rgb, int capacity, int* n_out, const VoxelBothOrders* both) {
    if (!ctx || !n_out || n < 0 || capacity < 0 || !(voxel > 0.f) || (n > 0 && !d_xyz)) return TDV_ERR_BAD_ARG;
    if (order != TDV_VOXEL_ORDER_FIRST && order != TDV_VOXEL_ORDER_REFERENCE) return TDV_ERR_BAD_ARG;
    if (both && (order != TDV_VOXEL_ORDER_REFERENCE || !both->first_xyz || !both->ref2first || !both->first2ref)) return TDV_ERR_BAD_ARG;
    *n_out = 0;
    if (n == 0) return TDV_OK;
    const bool legacy = study_env("TDV_VOXEL_LEGACY") != nullptr || study_env("TDV_VOXEL_SORT") != nullptr;   // A/B knobs: the counting-sort path / the full sort (read per call: the tests switch it)
    if (legacy) return voxel_downsample_impl(ctx, d_xyz, d_rgb, n, voxel, order, d_out_xyz, d_out_rgb, capacity, n_out, false, both);
    // hash-table path (memset + 2 kernels); first-occurrence order lands in the caller's buffer directly
    hipStream_t s = ctx->stream;
    const bool ref = order == TDV_VOXEL_ORDER_REFERENCE;
    int *d_voff, *d_rank = nullptr; int4* d_leaders = nullptr;
    float *tmp_xyz = d_out_xyz, *tmp_rgb = d_out_rgb;
    int cap_first = capacity;
    TDV_TRY(ws_alloc(ctx, 3, &d_voff));
    if (ref) {
        cap_first = n;
        TDV_TRY(ws_alloc(ctx, (size_t)n, &d_rank));
        TDV_TRY(ws_alloc(ctx, (size_t)n, &d_leaders));
        if (both) tmp_xyz = both->first_xyz; else TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &tmp_xyz));
        tmp_rgb = nullptr;
        if (d_rgb && d_out_rgb) TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &tmp_rgb));
    }
    TDV_TRY(pin_reserve(ctx, 64));
    int* h = reinterpret_cast<int*>(ctx->pin);      // the finalize kernel stores {voxel count, overflow flag} here itself: a D2H copy of 8 bytes is a 9-us blit kernel
    h[0] = -1; h[1] = 0;
    {
        ScopedTimer tm(ctx, TDV_TIMER_VOXEL);
        TDV_TRY(voxel_hash_first_order(ctx, d_xyz, (d_rgb && d_out_rgb) ? d_rgb : nullptr, n, nullptr, 1, voxel, tmp_xyz, tmp_rgb, cap_first, d_rank, d_leaders,
                                       d_voff, h));
    }
    TDV_HIP(ctx, hipStreamSynchronize(s));
    if (h[0] < 0) { snprintf(ctx->err, sizeof(ctx->err), "voxel: the result did not reach the host"); return TDV_ERR_INTERNAL; }
    if (h[1] != VH_EMPTY)      // a voxel with more than VH_K members (a coarse grid): the counting-sort path takes any count
        return voxel_downsample_impl(ctx, d_xyz, d_rgb, n, voxel, order, d_out_xyz, d_out_rgb, capacity, n_out, false, both);
    const int v = h[0];
    *n_out = v;
    if (v > capacity) return TDV_ERR_BAD_ARG;
    if (!ref) return TDV_OK;
    // The container order: on the device (the batch's way, ~17 rehash periods of small launches) when there are enough voxels for
    // the host replay's 22 ns per voxel + two copies to cost more - measured cross-over around 10k voxels (tools/studies/voxel_probe.py);
    // TDV_VOXEL_DEVICE_ORDER=0 / 1 forces either.  Colours ride along through the permutation.
    const int dev_env = getenv("TDV_VOXEL_DEVICE_ORDER") ? atoi(getenv("TDV_VOXEL_DEVICE_ORDER")) : -1;   // (read per call: the tests switch it)
    if (TDV_HAVE_LIBSTDCXX_EMULATION && (dev_env < 0 ? v >= 10000 : dev_env != 0)) {
        int *r2f, *f2r; int failed = 1;
        if (both) { r2f = both->ref2first; f2r = both->first2ref; }
        else { TDV_TRY(ws_alloc(ctx, (size_t)v, &r2f)); TDV_TRY(ws_alloc(ctx, (size_t)v, &f2r)); }
        int* d_voff1;
        TDV_TRY(ws_alloc(ctx, 2, &d_voff1));
        const int h_voff1[2] = {0, v};
        TDV_HIP(ctx, hipMemcpyAsync(d_voff1, h_voff1, 8, hipMemcpyHostToDevice, s));
        TDV_TRY(voxel_reference_order_batch_dev(ctx, 1, h_voff1, d_voff1, d_leaders, tmp_xyz, d_out_xyz, r2f, f2r, &failed));
        if (!failed) {
            if (tmp_rgb && d_out_rgb) {
                k_gather_rows3<<<(v + 255) / 256, 256, 0, s>>>(tmp_rgb, r2f, v, d_out_rgb);
                TDV_CHECK_LAUNCH(ctx);
                TDV_HIP(ctx, hipStreamSynchronize(s));
            }
            return TDV_OK;
        }
    }
    return voxel_reference_order(ctx, v, n, d_leaders, tmp_xyz, tmp_rgb, d_rank, 0, d_out_xyz, d_out_rgb, both);
}

// ---- the reference's container order ON THE DEVICE, for all clouds of a batch (round 3) ---------------------------------
// What the host replay above computes node by node has a closed form per rehash period.  Between two rehashes libstdc++ puts a
// new node at the FRONT of its bucket's run, and a bucket that was empty at the FRONT of the whole list (_M_insert_bucket_begin);
// a rehash walks the list front to back and re-inserts every node by the same rule (_M_rehash_aux).  So after a period with
// bucket count B the list is the REVERSE of: the period's insertion sequence - the previous list front to back, then the new
// keys in input order - grouped stably by bucket (code mod B), groups in order of first appearance.  With s(e) = position of
// element e in that sequence (its previous rank, or its own index if it is new: the previous list holds exactly the elements
// before it), first(b) = smallest s in bucket b, and F = the bucket sizes written at their first positions:
//     rank(e) = m - 1 - ( exclusive_scan(F)[first(bucket(e))] + #{x in bucket(e) : s(x) < s(e)} ),        m = elements so far
// One period = one memset + 3 small kernels + a scan over ALL clouds of the batch; ~17 periods reach 170k voxels (the bucket
// counts 13, 29, 59, 127 ... are libstdc++'s, taken from its own _Prime_rehash_policy on the host: they depend on the element
// count only).  Buckets hold their members' s in rows of RO_K; a fuller bucket (a degenerate hash) flags its cloud, which is then
// finished by the host replay.  No leader leaves the device, the lanes of the batch never wait for the host.
constexpr int RO_K = 16;
__global__ void k_ro_codes(const int4* __restrict__ leaders, int total_v, unsigned long long* __restrict__ code) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total_v) return;
    const int4 l = leaders[p];
    // VoxelKeyHash (registration.cpp:20-27) with std::hash<int> = the value converted to size_t
    unsigned long long h = (unsigned long long)(long long)l.x;
    h ^= (unsigned long long)(long long)l.y + 0x9e3779b9ull + (h << 6) + (h >> 2);
    h ^= (unsigned long long)(long long)l.z + 0x9e3779b9ull + (h << 6) + (h >> 2);
    code[p] = h;
}
// grid (ceil(n_next / 256), clouds).  A cloud takes part in the period while it has elements past n_k; m = its elements so far.
__global__ __launch_bounds__(256)
void k_ro_insert(const unsigned long long* __restrict__ code, const int* __restrict__ rank, const int* __restrict__ voff, int n_k, int n_next, unsigned nb,
                 int* __restrict__ fmax, int* __restrict__ cnt, int* __restrict__ rows, int* __restrict__ overflow) {
    const int b = blockIdx.y, v0 = voff[b], v = voff[b + 1] - v0;
    if (v <= n_k) return;
    const int m = min(v, n_next), e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m) return;
    const size_t P = (size_t)v0 + e;
    const int s = e < n_k ? rank[P] : e;
    const size_t slot = (size_t)b * nb + (size_t)(code[P] % (unsigned long long)nb);
    atomicMax(&fmax[slot], 0x7fffffff - s);                // first(b) = 0x7fffffff - fmax: a zero fill initialises it
    const int pos = atomicAdd(&cnt[slot], 1);
    if (pos < RO_K) rows[slot * RO_K + pos] = s; else overflow[b] = 1;
}
__global__ __launch_bounds__(256)
void k_ro_first_sizes(const int* __restrict__ voff, int n_k, int n_next, unsigned nb, const int* __restrict__ fmax, const int* __restrict__ cnt, int* __restrict__ F) {
    const int b = blockIdx.y;
    if (voff[b + 1] - voff[b] <= n_k) return;
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nb) return;
    const size_t slot = (size_t)b * nb + k;
    const int c = cnt[slot];
    if (c) F[(size_t)b * n_next + (0x7fffffff - fmax[slot])] = c;
}
__global__ __launch_bounds__(256)
void k_ro_rank(const unsigned long long* __restrict__ code, int* __restrict__ rank, const int* __restrict__ voff, int n_k, int n_next, unsigned nb,
               const int* __restrict__ fmax, const int* __restrict__ cnt, const int* __restrict__ rows, const int* __restrict__ S) {
    const int b = blockIdx.y, v0 = voff[b], v = voff[b + 1] - v0;
    if (v <= n_k) return;
    const int m = min(v, n_next), e = blockIdx.x * 256 + threadIdx.x;
    if (e >= m) return;
    const size_t P = (size_t)v0 + e;
    const int s = e < n_k ? rank[P] : e;
    const size_t slot = (size_t)b * nb + (size_t)(code[P] % (unsigned long long)nb);
    const int first = 0x7fffffff - fmax[slot], c = min(cnt[slot], RO_K);
    int before = S[(size_t)b * n_next + first] - S[(size_t)b * n_next];
    const int* row = rows + slot * RO_K;
    for (int q = 0; q < c; ++q) before += row[q] < s ? 1 : 0;
    rank[P] = m - 1 - before;                                // in place: a lane reads and writes only its own element's rank
}
// out[voff[b] + rank] = first-order voxel; the permutation both ways (positions local to the cloud)
__global__ __launch_bounds__(256)
void k_ro_permute(const float* __restrict__ first_xyz, const int* __restrict__ rank, const int* __restrict__ voff, const int* __restrict__ skip,
                  float* __restrict__ out_xyz, int* __restrict__ ref2first, int* __restrict__ first2ref) {
    const int b = blockIdx.y, v0 = voff[b], v = voff[b + 1] - v0, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= v || skip[b]) return;
    const size_t P = (size_t)v0 + e, O = (size_t)v0 + rank[P];
    out_xyz[3 * O] = first_xyz[3 * P]; out_xyz[3 * O + 1] = first_xyz[3 * P + 1]; out_xyz[3 * O + 2] = first_xyz[3 * P + 2];
    if (ref2first) { ref2first[O] = e; first2ref[P] = (int)(O - v0); }
}

#if TDV_HAVE_LIBSTDCXX_EMULATION
// (first element index, bucket count) of every rehash period a container of up to n_max elements goes through, from the
// library's own policy object
static void rehash_schedule(int n_max, std::vector<std::pair<int, unsigned>>& out) {
    static std::mutex mu; static std::vector<std::pair<int, unsigned>> cache; static int cached_to = 0;
    static std::__detail::_Prime_rehash_policy policy; static size_t buckets = 1;
    std::lock_guard<std::mutex> lock(mu);
    for (; cached_to < n_max; ++cached_to) {
        const auto rh = policy._M_need_rehash(buckets, (size_t)cached_to, 1);
        if (rh.first) { buckets = rh.second; cache.emplace_back(cached_to, (unsigned)buckets); }
    }
    out.clear();
    for (auto& p : cache) if (p.first < n_max) out.push_back(p);
}
#endif

// Reference order of every cloud of a batch on the device.  d_leaders / d_first_xyz / outputs are indexed by global first-
// occurrence position (cloud b at [h_voff[b], h_voff[b + 1])); d_voff: the same offsets on the device.  h_failed[b] = 1: a
// bucket of cloud b overflowed its row (or this build has no libstdc++): its slice of the outputs is untouched, the caller
// finishes it with voxel_reference_order.
int voxel_reference_order_batch_dev(tdv_ctx* ctx, int n_clouds, const int* h_voff, const int* d_voff, const int4* d_leaders, const float* d_first_xyz,
                                    float* d_out_xyz, int* d_ref2first, int* d_first2ref, int* h_failed) {
    for (int b = 0; b < n_clouds; ++b) h_failed[b] = TDV_HAVE_LIBSTDCXX_EMULATION ? 0 : 1;
#if TDV_HAVE_LIBSTDCXX_EMULATION
    const int total_v = h_voff[n_clouds];
    if (total_v == 0) return TDV_OK;
    int v_max = 0;
    for (int b = 0; b < n_clouds; ++b) v_max = std::max(v_max, h_voff[b + 1] - h_voff[b]);
    std::vector<std::pair<int, unsigned>> sched;
    rehash_schedule(v_max, sched);
    hipStream_t s = ctx->stream;
    unsigned long long* code; int *rank, *d_overflow;
    TDV_TRY(ws_alloc(ctx, (size_t)total_v, &code));
    TDV_TRY(ws_alloc(ctx, (size_t)total_v, &rank));
    TDV_TRY(ws_alloc(ctx, (size_t)n_clouds, &d_overflow));
    TDV_HIP(ctx, hipMemsetAsync(d_overflow, 0, (size_t)n_clouds * 4, s));
    k_ro_codes<<<(total_v + 255) / 256, 256, 0, s>>>(d_leaders, total_v, code);
    const WsMark mark = ws_mark(ctx);
    ScopedTimer tm(ctx, TDV_TIMER_VOXEL);
    for (size_t k = 0; k < sched.size(); ++k) {
        const int n_k = sched[k].first, n_next = std::min(k + 1 < sched.size() ? sched[k + 1].first : v_max, v_max);
        const unsigned nb = sched[k].second;
        ws_rewind(ctx, mark);                                  // every period reuses the same scratch (stream order keeps them apart)
        const size_t n_tab = (size_t)n_clouds * nb, n_F = (size_t)n_clouds * n_next;
        if (n_F + 1 > 0x7fffffffull) { for (int b = 0; b < n_clouds; ++b) h_failed[b] = 1; return TDV_OK; }   // (beyond the 32-bit scan: host replay)
        int *zero, *rows, *S, *d_tot;
        TDV_TRY(ws_alloc(ctx, 2 * n_tab + n_F, &zero));       // fmax | cnt | F: one fill
        TDV_TRY(ws_alloc(ctx, n_tab * RO_K, &rows));
        TDV_TRY(ws_alloc(ctx, n_F, &S));
        TDV_TRY(ws_alloc(ctx, 1, &d_tot));
        int *fmax = zero, *cnt = zero + n_tab, *F = zero + 2 * n_tab;
        TDV_HIP(ctx, hipMemsetAsync(zero, 0, (2 * n_tab + n_F) * 4, s));
        const dim3 ge((n_next + 255) / 256, n_clouds), gb((nb + 255) / 256, n_clouds);
        k_ro_insert<<<ge, 256, 0, s>>>(code, rank, d_voff, n_k, n_next, nb, fmax, cnt, rows, d_overflow);
        k_ro_first_sizes<<<gb, 256, 0, s>>>(d_voff, n_k, n_next, nb, fmax, cnt, F);
        TDV_TRY(exclusive_scan_dev(ctx, F, (int)n_F, S, d_tot));
        k_ro_rank<<<ge, 256, 0, s>>>(code, rank, d_voff, n_k, n_next, nb, fmax, cnt, rows, S);
        TDV_CHECK_LAUNCH(ctx);
    }
    k_ro_permute<<<dim3((v_max + 255) / 256, n_clouds), 256, 0, s>>>(d_first_xyz, rank, d_voff, d_overflow, d_out_xyz, d_ref2first, d_first2ref);
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, (size_t)n_clouds * 4));
    TDV_HIP(ctx, hipMemcpyAsync(ctx->pin, d_overflow, (size_t)n_clouds * 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    std::memcpy(h_failed, ctx->pin, (size_t)n_clouds * 4);
    ws_rewind(ctx, mark);
#endif
    return TDV_OK;
}

// All clouds of a batch at once (first-occurrence order; the reference order is finished per cloud with
// voxel_reference_order).  d_seg_off: n_clouds + 1 device offsets into d_xyz.  Voxels of cloud b end up at
// [h_voff[b], h_voff[b + 1]) of d_first_xyz (room for `total` points); d_rank / d_leaders (optional, `total` entries each) are what
// voxel_reference_order needs.  *overflowed: a voxel held more than VH_K points - nothing of the output is valid, the caller
// falls back to per-cloud calls.  d_voff_keep (optional): n_clouds + 2 device ints that receive the voxel offsets.  Synchronizes once.
int voxel_downsample_batch_dev(tdv_ctx* ctx, const float* d_xyz, int total, const int* d_seg_off, int n_clouds, float voxel,
                               float* d_first_xyz, int* d_rank, int4* d_leaders, int* h_voff, int* overflowed, int* d_voff_keep,
                               const float* pinhole4, const int* h_seg_off) {
    if (!ctx || n_clouds < 1 || total < 0 || !(voxel > 0.f) || !h_voff || !overflowed) return TDV_ERR_BAD_ARG;
    *overflowed = 0;
    for (int b = 0; b <= n_clouds; ++b) h_voff[b] = 0;
    if (total == 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    int* d_voff = d_voff_keep;                               // (n_clouds + 2 ints; the caller's when it wants the offsets on the device afterwards)
    if (!d_voff) TDV_TRY(ws_alloc(ctx, (size_t)n_clouds + 2, &d_voff));
    TDV_TRY(pin_reserve(ctx, ((size_t)n_clouds + 2) * 4));
    int* h = reinterpret_cast<int*>(ctx->pin);
    // Clouds unprojected from a depth image with the given intrinsics (pinhole4 = fx, fy, cx, cy; h_seg_off = the clouds' offsets on the
    // host) are grouped through pixel windows, without the table (k_vs_group); if a tile cannot be (coarse voxels, very long rows, a cloud
    // that is not what the caller said) the call is redone through the table.  TDV_VOXEL_PIXELS=0 (parity tests): the table at once.
    VoxelPinhole cam{};
    const char* px_env = getenv("TDV_VOXEL_PIXELS");
    bool by_pixels = pinhole4 && h_seg_off && !(px_env && atoi(px_env) == 0);
    if (by_pixels) { cam.fx = pinhole4[0]; cam.fy = pinhole4[1]; cam.cx = pinhole4[2]; cam.cy = pinhole4[3]; by_pixels = cam.fx > 0.f && cam.fy > 0.f; }
    std::vector<int> seg_copy;
    if (by_pixels) seg_copy.assign(h_seg_off, h_seg_off + n_clouds + 1);          // (the pinned staging below is reused by the tile table)
    for (int attempt = 0; attempt < 2; ++attempt) {
        const WsMark mark = ws_mark(ctx);
        {
            ScopedTimer tm(ctx, TDV_TIMER_VOXEL);
            TDV_TRY(voxel_hash_first_order(ctx, d_xyz, nullptr, total, d_seg_off, n_clouds, voxel, d_first_xyz, nullptr, total, d_rank, d_leaders, d_voff, nullptr,
                                           by_pixels ? &cam : nullptr, by_pixels ? seg_copy.data() : nullptr));
        }
        TDV_TRY(pin_reserve(ctx, ((size_t)n_clouds + 2) * 4));
        h = reinterpret_cast<int*>(ctx->pin);
        TDV_HIP(ctx, hipMemcpyAsync(h, d_voff, ((size_t)n_clouds + 2) * 4, hipMemcpyDeviceToHost, s));   // offsets, then the overflow flag
        TDV_HIP(ctx, hipStreamSynchronize(s));
        ctx->last_voxel_grouping = by_pixels ? 2 : 1;
        if (by_pixels && h[n_clouds + 1] == VS_FAILED) { by_pixels = false; ws_rewind(ctx, mark); continue; }   // a tile the pixel windows do not cover: the table
        break;
    }
    if (h[n_clouds + 1] != VH_EMPTY) { *overflowed = 1; return TDV_OK; }
    std::memcpy(h_voff, h, ((size_t)n_clouds + 1) * 4);
    return TDV_OK;
}

static int voxel_downsample_impl(tdv_ctx* ctx, const float* d_xyz, const float* d_rgb, int n, float voxel, int order,
                                 float* d_out_xyz, float* d_out_rgb, int capacity, int* n_out, bool full_sort, const VoxelBothOrders* both) {
    if (!ctx || !n_out || n < 0 || capacity < 0 || !(voxel > 0.f) || (n > 0 && !d_xyz)) return TDV_ERR_BAD_ARG;
    if (order != TDV_VOXEL_ORDER_FIRST && order != TDV_VOXEL_ORDER_REFERENCE) return TDV_ERR_BAD_ARG;
    if (both && (order != TDV_VOXEL_ORDER_REFERENCE || !both->first_xyz || !both->ref2first || !both->first2ref)) return TDV_ERR_BAD_ARG;
    *n_out = 0;
    if (n == 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    const float inv = 1.0f / voxel;  // registration.cpp:32
    size_t n_pow2 = BT_TILE;
    while (n_pow2 < (size_t)n) n_pow2 <<= 1;
    uint4* rec; int *leader, *rank, *sums, *d_total;
    TDV_TRY(ws_alloc(ctx, n_pow2, &rec));
    TDV_TRY(ws_alloc(ctx, (size_t)n, &rank));
    const int sblocks = (n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS;
    TDV_TRY(ws_alloc(ctx, (size_t)sblocks, &sums));
    TDV_TRY(ws_alloc(ctx, 1, &d_total));
    TDV_TRY(pin_reserve(ctx, 64));
    ScopedTimer tm(ctx, TDV_TIMER_VOXEL);
    static const bool force_sort = study_env("TDV_VOXEL_SORT") != nullptr;   // A/B knob: the full bitonic sort
    bool hashed = !force_sort && !full_sort;
    int* d_too_big = nullptr;
    float *mean_xyz = nullptr, *mean_rgb = nullptr;     // hashed path: means parked at their leader's input index
    if (hashed) {
        // 9 launches (round 1: 15): one memset, cell histogram, scan (2), scatter, bucket sort + leaders + means, scan (2), compact
        size_t nb = 4096;
        while (nb < 2 * (size_t)n) nb <<= 1;
        int *zeroed, *hist, *cursor, *start, *d_tot2; uint4* rec_in;
        TDV_TRY(ws_alloc(ctx, 2 * nb + 1 + (size_t)n, &zeroed));        // hist | cursor | too_big | leader: one memset
        hist = zeroed; cursor = zeroed + nb; d_too_big = zeroed + 2 * nb; leader = zeroed + 2 * nb + 1;
        TDV_TRY(ws_alloc(ctx, nb, &start));
        TDV_TRY(ws_alloc(ctx, 1, &d_tot2));
        TDV_TRY(ws_alloc(ctx, (size_t)n, &rec_in));
        TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &mean_xyz));
        if (d_rgb && d_out_rgb) TDV_TRY(ws_alloc(ctx, (size_t)n * 3, &mean_rgb));
        TDV_HIP(ctx, hipMemsetAsync(zeroed, 0, (2 * nb + 1 + (size_t)n) * 4, s));
        k_voxel_hist<<<(n + 255) / 256, 256, 0, s>>>(d_xyz, n, inv, (unsigned)(nb - 1), rec_in, hist);
        TDV_TRY(exclusive_scan_dev(ctx, hist, (int)nb, start, d_tot2));
        k_voxel_scatter<<<(n + 255) / 256, 256, 0, s>>>(rec_in, n, (unsigned)(nb - 1), start, cursor, rec);
        k_voxel_bucket_emit<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(rec, start, hist, (int)nb, n, d_too_big, d_xyz, mean_rgb ? d_rgb : nullptr, leader, mean_xyz, mean_rgb);
    } else {
        TDV_TRY(ws_alloc(ctx, (size_t)n, &leader));
        k_voxel_records<<<(unsigned)((n_pow2 + 255) / 256), 256, 0, s>>>(d_xyz, n, (int)n_pow2, inv, rec);
        TDV_TRY(sort_records_dev(ctx, rec, n_pow2));
        TDV_HIP(ctx, hipMemsetAsync(leader, 0, (size_t)n * 4, s));
        k_voxel_heads<<<(n + 255) / 256, 256, 0, s>>>(rec, n, leader);
    }
    k_scan_reduce<<<sblocks, 1024, 0, s>>>(leader, n, sums, d_total, ctx->scan_ticket);
    k_scan_local<<<sblocks, 1024, 0, s>>>(leader, n, sums, rank);
    TDV_CHECK_LAUNCH(ctx);
    int* h_total = reinterpret_cast<int*>(ctx->pin);
    h_total[1] = 0;
    TDV_HIP(ctx, hipMemcpyAsync(h_total, d_total, 4, hipMemcpyDeviceToHost, s));
    if (hashed) TDV_HIP(ctx, hipMemcpyAsync(h_total + 1, d_too_big, 4, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    if (hashed && h_total[1])   // a bucket too large for the per-lane sort (very coarse grid): redo with the full sort
        return voxel_downsample_impl(ctx, d_xyz, d_rgb, n, voxel, order, d_out_xyz, d_out_rgb, capacity, n_out, true, both);
    const int v = *h_total;
    *n_out = v;
    if (v > capacity) return TDV_ERR_BAD_ARG;
    if (order == TDV_VOXEL_ORDER_FIRST) {
        if (hashed) k_voxel_compact<<<(n + 255) / 256, 256, 0, s>>>(leader, rank, n, mean_xyz, mean_rgb, capacity, d_out_xyz, d_out_rgb);
        else k_voxel_means<<<(n + 255) / 256, 256, 0, s>>>(rec, n, d_xyz, d_rgb, rank, capacity, d_out_xyz, d_out_rgb);
        TDV_CHECK_LAUNCH(ctx);
        return TDV_OK;
    }
    // reference order: replay the container on the host to get its iteration order
    float *tmp_xyz, *tmp_rgb = nullptr;
    if (both) tmp_xyz = both->first_xyz;
    else TDV_TRY(ws_alloc(ctx, (size_t)v * 3, &tmp_xyz));
    if (d_rgb && d_out_rgb) TDV_TRY(ws_alloc(ctx, (size_t)v * 3, &tmp_rgb));
    if (hashed) k_voxel_compact<<<(n + 255) / 256, 256, 0, s>>>(leader, rank, n, mean_xyz, mean_rgb, v, tmp_xyz, tmp_rgb);
    else k_voxel_means<<<(n + 255) / 256, 256, 0, s>>>(rec, n, d_xyz, d_rgb, rank, v, tmp_xyz, tmp_rgb);
    TDV_CHECK_LAUNCH(ctx);
    // leaders (first point of every voxel: cell + input index) in input order, from the device; the host only replays those
    int4* d_leaders;
    TDV_TRY(ws_alloc(ctx, (size_t)v, &d_leaders));
    k_voxel_leader_list<<<(n + 255) / 256, 256, 0, s>>>(leader, rank, d_xyz, inv, n, d_leaders);
    TDV_CHECK_LAUNCH(ctx);
    return voxel_reference_order(ctx, v, n, d_leaders, tmp_xyz, tmp_rgb, rank, 0, d_out_xyz, d_out_rgb, both);
}

// The reference's container order for v voxels held in first-occurrence order (tmp_*): leaders (cell + input index of the first
// point of every voxel, at its first-occurrence rank) go to the host, which replays the container; d_rank[input index] =
// first-occurrence rank.  out[p] = tmp[rank[leader index of the p-th voxel of the container]].
int voxel_reference_order(tdv_ctx* ctx, int v, int n, const int4* d_leaders, const float* tmp_xyz, const float* tmp_rgb, const int* d_rank, int rank_base,
                          float* d_out_xyz, float* d_out_rgb, const VoxelBothOrders* both) {
    hipStream_t s = ctx->stream;
    int* d_order;
    TDV_TRY(ws_alloc(ctx, (size_t)v, &d_order));
    // through pinned memory both ways (a pageable std::vector made these two copies staged ones: 2.3 MB + 0.6 MB per C4 instance)
    const size_t pin_order_off = align_up((size_t)v * sizeof(int4), 64);
    TDV_TRY(pin_reserve(ctx, pin_order_off + (size_t)v * 4));
    const int4* leaders = reinterpret_cast<const int4*>(ctx->pin);
    int* order_pinned = reinterpret_cast<int*>(ctx->pin + pin_order_off);
    TDV_HIP(ctx, hipMemcpyAsync(ctx->pin, d_leaders, (size_t)v * sizeof(int4), hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    const bool real_map = !TDV_HAVE_LIBSTDCXX_EMULATION || study_env("TDV_VOXEL_REAL_MAP") != nullptr;   // A/B knob: a real std::unordered_map instead of the emulation
    const auto t_host0 = std::chrono::steady_clock::now();
    int n_first = 0;
    struct { int* p; int* n; void push_back(int x) { p[(*n)++] = x; } } order_first{order_pinned, &n_first};
    for (int r = 0; r < v; ++r)
        if (leaders[r].w < 0 || leaders[r].w >= n) { snprintf(ctx->err, sizeof(ctx->err), "voxel: bad leader index %d", leaders[r].w); return TDV_ERR_INTERNAL; }
    if (real_map) {
        // later members of a voxel never change the container (grid[key] finds the node), so inserting the first
        // occurrences in input order builds exactly the reference's table
        std::unordered_map<VoxelKey, int, VoxelKeyHash> grid;
        for (int r = 0; r < v; ++r) grid.emplace(VoxelKey{leaders[r].x, leaders[r].y, leaders[r].z}, leaders[r].w);
        if ((int)grid.size() != v) {
            snprintf(ctx->err, sizeof(ctx->err), "voxel: host replay found %zu voxels, device %d", grid.size(), v);
            return TDV_ERR_INTERNAL;
        }
        for (auto& kv : grid) order_first.push_back(kv.second);
    } else {
#if TDV_HAVE_LIBSTDCXX_EMULATION
        LibstdcxxInsertionOrder order((size_t)v);
        VoxelKeyHash hasher;
        for (int r = 0; r < v; ++r) order.insert(hasher(VoxelKey{leaders[r].x, leaders[r].y, leaders[r].z}));
        order.for_each([&](int r) { order_first.push_back(leaders[r].w); });
#endif
    }
    if (getenv("TDV_DEBUG")) fprintf(stderr, "[tdv] voxel reference order: %d leaders replayed in %.3f ms (%s)\n", v,
                                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count(), real_map ? "std::unordered_map" : "emulation");
    if (n_first != v) { snprintf(ctx->err, sizeof(ctx->err), "voxel: host replay listed %d voxels, device %d", n_first, v); return TDV_ERR_INTERNAL; }
    TDV_HIP(ctx, hipMemcpyAsync(d_order, order_pinned, (size_t)v * 4, hipMemcpyHostToDevice, s));
    k_voxel_permute<<<(v + 255) / 256, 256, 0, s>>>(tmp_xyz, tmp_rgb, d_rank, rank_base, d_order, v, d_out_xyz, d_out_rgb,
                                                    both ? both->ref2first : nullptr, both ? both->first2ref : nullptr);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipStreamSynchronize(s));  // the pinned staging is reused by the next call
    return TDV_OK;
}

}  // namespace tdv
