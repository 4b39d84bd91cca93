// ICP on gfx950: brute-force nearest-neighbour correspondence search + normal equations +
// small solve, the whole loop resident on the device.
//
// Replaces GPURegistration::icpRefine (/root/reference/src/gpu_impl.cpp:141-260, kernels
// cuda/icp.cu:14-55 and :90-142); results follow the CPU oracle Registration::icpRefine
// (/root/reference/src/registration.cpp:297-414), including where the CUDA path differs
// (transformed point in J and r, inclusive threshold, n_corr<3 break, point-to-point mode).
//
// Kernel design (VALU-bound, not HBM- and not MFMA-bound: 8 f32 ops per pair, K=3):
//  * icp_nn_scan: each lane keeps NN_SPL transformed source points in VGPRs; the target cloud
//    is read as structure-of-arrays through the SCALAR data path (wave-uniform s_load_dwordx16
//    of 16 x, 16 y, 16 z), so every distance op is one VALU instruction with an SGPR operand and
//    neither LDS nor vector-memory instructions sit in the inner loop.  Argmin is two-level:
//    the loop tracks min d2 and the first CHUNK of 16 targets that reached it (strict <),
//    8 + 0.5 + 0.3 VALU ops per pair; the exact index inside the chunk is recovered later by
//    re-evaluating 16 distances.  d2 = dx*dx + (dy*dy + dz*dz) with no FMA contraction, so the
//    argmin is bit-identical to the CPU scan (lowest index wins ties).
//    The grid is (source blocks) x (target splits) so that >> 256 workgroups are in flight;
//    splits are combined in split order with strict <, which preserves the tie rule.
//  * icp_accumulate: one lane per source; resolves the index, applies the sqrt-free inclusive
//    threshold, builds J = [p x n | n], r = (p-q).n (or the point-to-point moments) and reduces
//    in a FIXED order (wave64 shuffles -> LDS -> one slab per block); f64 accumulators.
//    The block that finishes last (atomic ticket) folds the slabs in fixed order, one lane solves the 6x6 system
//    (pivoted LDL^T) or the 3x3 Kabsch SVD, updates T on the device and evaluates convergence: two launches per
//    iteration.
//  * icp_nn_pruned (large clouds, same correspondences bit for bit): the target is put in Morton order once per
//    call; ONE WAVE PER SOURCE POINT tests the target's 4096- and 64-point bounding boxes one per lane, visits them
//    best-first, and evaluates the points of the boxes that can hold a neighbour within the bound one per lane
//    (bound = the inclusive acceptance threshold, then the best distance found).  Float subtraction, multiplication
//    and addition are monotone under round-to-nearest, so the box bound needs no margin; ties keep the lowest
//    original target index as the scan does.  accumulate/solve run unchanged.
// No float atomics anywhere: two runs give identical bits.
#include "tdv_internal.hpp"
#include "device_linalg.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <algorithm>

namespace tdv {

constexpr int NN_SPL = 2;      // source points per lane
constexpr int NN_CH = 16;      // targets per chunk (one s_load_dwordx16 per coordinate)
#ifndef NN_BLOCK_VALUE
#define NN_BLOCK_VALUE 128
#endif
#ifndef NN_WG_TARGET
#define NN_WG_TARGET 12288
#endif
constexpr int NN_BLOCK = NN_BLOCK_VALUE;    // two waves per workgroup (no LDS, no barrier): many workgroups with few target splits,
                                 // so the per-split partials (8 B per source per split) stay small
constexpr int NN_SRC_PER_BLOCK = NN_SPL * NN_BLOCK;
constexpr double PRUNED_MIN_PAIRS = 1e8;   // auto mode: pruned search from this many source x target pairs ...
constexpr int PRUNED_MIN_TARGETS = 4096;    // ... and targets (measured: 50k x 10k 0.042 vs 0.081 ms, 128k x 9.4k see DESIGN)
constexpr int ACC_NV = 32;     // reduction slots per block (29 used p2plane, 17 p2point)

struct IcpState {
    float T[16];       // current transform, column-major
    float res_T[16];   // result.transformation
    float fitness, rmse;
    int n_corr;        // accepted correspondences of the last evaluated iteration
    int applied;       // iterations whose update was applied
    int done;          // loop finished (converged / n_corr < 3)
    int iter;          // iterations evaluated
    int last_n_corr_applied;
    int pad;
};

__global__ void k_aos_to_soa_pad(const float* __restrict__ aos, int n, int n_pad, float pad,
                                 float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    if (i < n) { x[i] = aos[3 * i]; y[i] = aos[3 * i + 1]; z[i] = aos[3 * i + 2]; }
    else { x[i] = pad; y[i] = pad; z[i] = pad; }
}

__device__ __forceinline__ void transform_point(const float* __restrict__ T, float sx, float sy, float sz,
                                                float& px, float& py, float& pz) {
    // p = R*s + t with each row evaluated as r0*sx + (r1*sy + r2*sz), then + t (column-major T)
    px = (T[0] * sx + (T[4] * sy + T[8] * sz)) + T[12];
    py = (T[1] * sx + (T[5] * sy + T[9] * sz)) + T[13];
    pz = (T[2] * sx + (T[6] * sy + T[10] * sz)) + T[14];
}

__global__ __launch_bounds__(NN_BLOCK)
void k_icp_nn_scan(const float* __restrict__ src, int ns, int ns_pad,
                   const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                   int n_chunks, int chunks_per_split,
                   const IcpState* __restrict__ st,
                   float* __restrict__ pd2, int* __restrict__ pchunk) {
    if (st->done) return;
    // XCD-aware deal (round 4, as k_ransac_score_fast): workgroups are dispatched x-fastest and an XCD takes every eighth of them, so with
    // the plain mapping every XCD's L2 pulled ALL target splits (PMC round 3: 49.7 MB per launch for 7.2 MB).  When the split count is a
    // multiple of 8 an XCD takes every eighth SPLIT and all source blocks of it: its L2 holds an eighth of the targets.
    int bxi = blockIdx.x, split = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, k = lin >> 3;
        bxi = k % gridDim.x; split = (k / gridDim.x) * 8 + (lin & 7);
    }
    const int c0 = split * chunks_per_split;
    const int c1 = min(n_chunks, c0 + chunks_per_split);
    const int base = bxi * NN_SRC_PER_BLOCK + threadIdx.x;
    float px[NN_SPL], py[NN_SPL], pz[NN_SPL], best[NN_SPL];
    int bc[NN_SPL];
#pragma unroll
    for (int s = 0; s < NN_SPL; ++s) {
        int i = min(base + s * NN_BLOCK, ns - 1);
        transform_point(st->T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px[s], py[s], pz[s]);
        best[s] = FLT_MAX; bc[s] = 0;
    }
    for (int c = c0; c < c1; ++c) {
        const int j = c * NN_CH;
        float qx[NN_CH], qy[NN_CH], qz[NN_CH];
#pragma unroll
        for (int t = 0; t < NN_CH; ++t) { qx[t] = tx[j + t]; qy[t] = ty[j + t]; qz[t] = tz[j + t]; }
#pragma unroll
        for (int s = 0; s < NN_SPL; ++s) {
            float m = FLT_MAX;
#pragma unroll
            for (int t = 0; t < NN_CH; ++t) {
                float dx = px[s] - qx[t], dy = py[s] - qy[t], dz = pz[s] - qz[t];
                float d2 = dx * dx + (dy * dy + dz * dz);
                m = fminf(m, d2);
            }
            bool lt = m < best[s];
            best[s] = lt ? m : best[s];
            bc[s] = lt ? j : bc[s];
        }
    }
#pragma unroll
    for (int s = 0; s < NN_SPL; ++s) {
        size_t o = (size_t)split * ns_pad + base + s * NN_BLOCK;
        pd2[o] = best[s]; pchunk[o] = bc[s];
    }
}

// ---- exact pruned search --------------------------------------------------------------------------------------
#ifndef PN_WAVES_VALUE
#define PN_WAVES_VALUE 1
#endif
constexpr int PN_WAVES = PN_WAVES_VALUE;   // waves per workgroup (measured at 200k: 1 wave 0.169 ms, 4: 0.177, 16: 0.217)
#ifndef PN_PTS_VALUE
#define PN_PTS_VALUE 1
#endif
// Consecutive source points per wave; their results leave as ONE store per array.  With 1 every result is its own 4-B
// partial-line write (13 MB of WRITE_SIZE per launch for 1.6 MB of results at 200k), which costs bytes but no time: the
// kernel is bound by its box walk, and longer waves balance worse — measured at 200k x 200k (tools/studies/icp_probe.py,
// profiles/r2/history/icp_nn_points_per_wave.jsonl): 1: 0.131 ms, 4: 0.141, 8: 0.144, 16: 0.155, 32: 0.166 ms per launch.
constexpr int PN_PTS = PN_PTS_VALUE;

// lower bound of fl(d2) between point p and any point inside box idx (see knn.hip: box_lower_bound)
__device__ __forceinline__ float point_box_lb(const float* __restrict__ box, int count, int idx, float px, float py, float pz) {
    const float gx = fmaxf(0.f, fmaxf(box[idx] - px, px - box[(size_t)3 * count + idx]));
    const float gy = fmaxf(0.f, fmaxf(box[(size_t)count + idx] - py, py - box[(size_t)4 * count + idx]));
    const float gz = fmaxf(0.f, fmaxf(box[(size_t)2 * count + idx] - pz, pz - box[(size_t)5 * count + idx]));
    return gx * gx + (gy * gy + gz * gz);
}

// One wave per source point at a time, PN_PTS consecutive points per wave (lane p keeps point p's result and the wave
// stores them together).  The 4096-point boxes are tested one per lane and visited best-first (smallest lower
// bound first, until it exceeds the bound); inside one, its 64 leaves (64 points each) are tested one per lane, and
// the points of a leaf that passes are evaluated one per lane (coalesced).  bound = the inclusive acceptance
// threshold, then the best distance found so far.  Each lane keeps its own (d2, original index) minimum; the wave
// minimum in that lexicographic order is the brute-force scan's answer (lowest index on ties).
__global__ __launch_bounds__(64 * PN_WAVES)
void k_icp_nn_pruned(const float* __restrict__ src, int ns,
                     const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                     const int* __restrict__ torig, int nt,
                     const float* __restrict__ lbox, int n_leaf, const float* __restrict__ tbox, int n_top,
                     const IcpState* __restrict__ st, float tau,
                     float* __restrict__ out_d2, int* __restrict__ out_idx) {
    if (st->done) return;
    const int lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * PN_WAVES + (threadIdx.x >> 6)) * PN_PTS;
    if (i0 >= ns) return;   // wave-uniform
    float res_d = FLT_MAX; int res_o = 0;   // lane p: the result of point i0 + p
    for (int p = 0; p < PN_PTS && i0 + p < ns; ++p) {
    const int i = i0 + p;
    float px, py, pz;
    transform_point(st->T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px, py, pz);
    float B = tau;           // wave-uniform; a target passes iff d2 <= B
    float bd = INFINITY;     // this lane's best
    int bo = INT_MAX;
    for (int tb = 0; tb < n_top; tb += 64) {
        const int t = tb + lane;
        float lbt = t < n_top ? point_box_lb(tbox, n_top, t, px, py, pz) : INFINITY;
        while (true) {
            // best-first over the remaining 4096-point boxes of this group
            float m = lbt;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
            if (!(m <= B) || m == INFINITY) break;   // +inf marks visited / absent boxes whatever the bound is
            const int tl = __ffsll((long long)__ballot(lbt == m)) - 1;   // wave-uniform lane of that box
            if (lane == tl) lbt = INFINITY;                               // visited
            const int u = (tb + tl) * 64 + lane;
            const float lbl = u < n_leaf ? point_box_lb(lbox, n_leaf, u, px, py, pz) : INFINITY;
            unsigned long long lmask = __ballot(u < n_leaf && lbl <= B);
            while (lmask) {
                const int bl = __ffsll((long long)lmask) - 1;
                lmask &= lmask - 1;
                if (__shfl(lbl, bl, 64) > B) continue;   // the bound may have dropped since the test
                const int j = ((tb + tl) * 64 + bl) * 64 + lane;   // arrays are padded with +inf to a multiple of 256
                float dx = px - tx[j], dy = py - ty[j], dz = pz - tz[j];
                float d2 = dx * dx + (dy * dy + dz * dz);
                const bool cand = j < nt && d2 <= B && d2 <= bd;
                if (__any(cand)) {
                    const int o = torig[j];
                    const bool take = cand && (d2 < bd || o < bo);
                    bd = take ? d2 : bd;
                    bo = take ? o : bo;
                    float nb = bd;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) nb = fminf(nb, __shfl_xor(nb, off, 64));
                    B = fminf(B, nb);
                }
            }
        }
    }
    // lexicographic (d2, index) minimum over the lanes
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64); const int oo = __shfl_xor(bo, off, 64);
        const bool take = od < bd || (od == bd && oo < bo);
        bd = take ? od : bd; bo = take ? oo : bo;
    }
    if (lane == p) { res_d = bo == INT_MAX ? FLT_MAX : bd; res_o = bo == INT_MAX ? 0 : bo; }
    }
    if (lane < PN_PTS && i0 + lane < ns) { out_d2[i0 + lane] = res_d; out_idx[i0 + lane] = res_o; }
}

// ---- hash-grid search ------------------------------------------------------------------------------------------------
// ICP accepts a correspondence only within `thr`, and the reference's pipeline sets thr = 0.4 x the voxel size: far below
// the point spacing.  One LANE per source point looks its neighbourhood up in a hash grid over the target and verifies
// what it finds with the scan's distance expression, instead of one WAVE per source point walking box levels.
// Cells are 2.2 thr wide, so per axis a neighbour within thr lies in the query's cell or in ONE adjacent cell, the one on
// the side of the cell the query sits in: 8 probes of an open-addressing table (8-B entries: 32-bit tag of the cell key, list head), all loaded
// before the first is looked at.  The points of a cell form a linked list of original indices (order irrelevant: the
// lowest (d2, index) is kept).  Every candidate is verified (d2 <= tau), so a tag collision or the garbage cell of a
// far-away query can only cost time.  Completeness: in cell units a pair within thr is at most r = 1/2.2 (1 + 1e-6) apart
// per axis; a computed coordinate fl(x * inv_cell) is within e = 2^-23 |coordinate| (two roundings) of the real one, i.e.
// e <= 2^-6 while |coordinate| < 2^17 - which k_grid_insert checks on every target (a query within thr of a target is in
// range with it).  A query with computed fraction f < 0.5 in cell k sees a neighbour's computed coordinate inside
// [k - r - 2e, k + 0.5 + r + 2e) and r + 2e < 0.486: cells k - 1 and k; symmetrically k and k + 1 for f >= 0.5.
constexpr unsigned GRID_EMPTY = ~0u;
#ifndef GRID_SLOTS_PER_POINT_X4
#define GRID_SLOTS_PER_POINT_X4 8      // table slots per target point, in quarters (8 = the power of two >= 2 n)
#endif
constexpr float GRID_CELL_FACTOR = 2.2f;
constexpr float GRID_MAX_COORD = 131072.0f;      // 2^17 cells
constexpr int GRID_PAD = 16;                    // slots past the table's end: probing never wraps (an insertion that would run past them makes the grid unusable)
constexpr int GRID_MAX_PER_CELL = 32;            // average over the occupied cells above which the grid is not used (measured at 200k x 200k, threshold in spacings: 2 -> 19 per cell, grid 0.13 ms vs walk 0.20; 4 -> 77 per cell, 0.37 vs 0.21)
struct __attribute__((aligned(8))) GridEntry { unsigned tag; int head; };   // tag: 32 bits of the cell key's hash (two cells that share a tag and a probe chain share a list: every candidate is verified anyway)
__device__ __forceinline__ unsigned long long grid_key(int ix, int iy, int iz) {
    return ((unsigned long long)((unsigned)ix & 0x1fffffu) << 42) | ((unsigned long long)((unsigned)iy & 0x1fffffu) << 21) | (unsigned long long)((unsigned)iz & 0x1fffffu);
}
__device__ __forceinline__ unsigned grid_slot(unsigned long long key, int shift) { return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> shift); }
__device__ __forceinline__ unsigned grid_tag(unsigned long long key) {
    const unsigned t = (unsigned)((key * 0xD6E8FEB86659FD93ull) >> 32);
    return t == GRID_EMPTY ? GRID_EMPTY - 1u : t;
}

// flags[0]: a coordinate out of range / non-finite; flags[1]: occupied cells
__global__ __launch_bounds__(256)
void k_grid_insert(const float* __restrict__ tgt, int nt, float inv_cell, GridEntry* table, float4* __restrict__ node,
                   unsigned mask, int shift, int* __restrict__ flags) {
    __shared__ int s_new, s_bad;
    if (threadIdx.x == 0) { s_new = 0; s_bad = 0; }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    bool fresh = false, bad = false;
    if (j < nt) {
        const float x = tgt[3 * j], y = tgt[3 * j + 1], z = tgt[3 * j + 2];
        const float fx = x * inv_cell, fy = y * inv_cell, fz = z * inv_cell;
        bad = !(fabsf(fx) < GRID_MAX_COORD && fabsf(fy) < GRID_MAX_COORD && fabsf(fz) < GRID_MAX_COORD);
        if (!bad) {
            const unsigned long long key = grid_key((int)floorf(fx), (int)floorf(fy), (int)floorf(fz));
            const unsigned tag = grid_tag(key);
            unsigned slot = grid_slot(key, shift);
            const unsigned last = mask + (unsigned)GRID_PAD - 2u;    // the two slots after it stay empty: a two-entry probe at `last` reads inside the table and ends there
            for (;;) {
                unsigned k = table[slot].tag;
                if (k == GRID_EMPTY) { k = atomicCAS(&table[slot].tag, GRID_EMPTY, tag); if (k == GRID_EMPTY) { fresh = true; break; } }
                if (k == tag) break;
                if (++slot > last) { bad = true; break; }            // no wrap-around (half full at most: a run this long at the very end is a freak; the caller falls back to the box walk)
            }
            if (!bad) node[j] = make_float4(x, y, z, __int_as_float(atomicExch(&table[slot].head, j)));   // the point and the next index of its cell's list
        }
    }
    const unsigned long long mf = __ballot(fresh), mb = __ballot(bad);
    if ((threadIdx.x & 63) == 0) { if (mf) atomicAdd(&s_new, __popcll(mf)); if (mb) atomicAdd(&s_bad, 1); }
    __syncthreads();
    if (threadIdx.x == 0) { if (s_new) atomicAdd(&flags[1], s_new); if (s_bad) atomicOr(&flags[0], 1); }
}

// the search of one query point p: lowest (d2, original index) among the targets with d2 <= tau, (INFINITY, INT_MAX) if none
__device__ __forceinline__ void grid_nearest(const GridEntry* __restrict__ table, const float4* __restrict__ node, unsigned mask, int shift,
                                             float inv_cell, float px, float py, float pz, float tau, float& bd, int& bo) {
    const float gx = px * inv_cell, gy = py * inv_cell, gz = pz * inv_cell;
    const float kx = floorf(gx), ky = floorf(gy), kz = floorf(gz);
    const int cx = (int)kx, cy = (int)ky, cz = (int)kz;
    const int sx = (gx - kx < 0.5f) ? -1 : 1, sy = (gy - ky < 0.5f) ? -1 : 1, sz = (gz - kz < 0.5f) ? -1 : 1;   // the adjacent cell that can matter
    // The eight cells are probed TOGETHER, two consecutive table entries per cell and round trip: a wave waits once per step
    // of the longest probe run among its 512 cells, not once per step of every cell in turn (200k x 200k: 33.3 -> 30.6 us;
    // 800k sources: 81.9 -> 73.7 us; tools/studies/icp_grid_scaling.py).  Probing does not wrap (k_grid_insert).
    unsigned tag[8], slot[8]; uint4 e[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned long long key = grid_key(cx + ((c & 1) ? sx : 0), cy + ((c & 2) ? sy : 0), cz + ((c & 4) ? sz : 0));
        tag[c] = grid_tag(key);
        slot[c] = grid_slot(key, shift);
        e[c] = *reinterpret_cast<const uint4*>(&table[slot[c]]);   // 16 bytes at an 8-byte-aligned address
    }
    (void)mask;
    int j[8];
    unsigned pending = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        j[c] = -1;
        if (e[c].x == tag[c]) j[c] = (int)e[c].y;
        else if (e[c].x != GRID_EMPTY) {
            if (e[c].z == tag[c]) j[c] = (int)e[c].w;
            else if (e[c].z != GRID_EMPTY) { slot[c] += 2u; pending |= 1u << c; }
        }
    }
    while (pending) {
#pragma unroll
        for (int c = 0; c < 8; ++c) if (pending & (1u << c)) e[c] = *reinterpret_cast<const uint4*>(&table[slot[c]]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (!(pending & (1u << c))) continue;
            pending &= ~(1u << c);
            if (e[c].x == tag[c]) j[c] = (int)e[c].y;
            else if (e[c].x != GRID_EMPTY) {
                if (e[c].z == tag[c]) j[c] = (int)e[c].w;
                else if (e[c].z != GRID_EMPTY) { slot[c] += 2u; pending |= 1u << c; }
            }
        }
    }
    bd = INFINITY; bo = INT_MAX;
    auto take = [&](const float4 t, int idx) {
        const float ex = px - t.x, ey = py - t.y, ez = pz - t.z;
        const float d2 = ex * ex + (ey * ey + ez * ez);    // the scan's expression
        if (d2 <= tau && (d2 < bd || (d2 == bd && idx < bo))) { bd = d2; bo = idx; }
    };
    float4 first[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) first[c] = j[c] >= 0 ? node[j[c]] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));   // the cells' first points, together
    int nx[8];
    bool more = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        nx[c] = -1;
        if (j[c] < 0) continue;
        take(first[c], j[c]);
        nx[c] = __float_as_int(first[c].w);
        more |= nx[c] >= 0;
    }
    // the rest of the lists (cells hold one or two points): the eight lists advance TOGETHER, one round trip per step of the
    // longest list among a wave's 512 cells instead of a sum over the cells
    while (more) {
        float4 t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) if (nx[c] >= 0) t[c] = node[nx[c]];
        more = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (nx[c] < 0) continue;
            take(t[c], nx[c]);
            nx[c] = __float_as_int(t[c].w);
            more |= nx[c] >= 0;
        }
    }
}

// One lane per source point.  (Running this search inside k_icp_accumulate - one launch per iteration, no correspondence
// arrays in between - was measured slower, 110 us against 44 + 35 at 200k x 200k: that kernel's 29 double accumulators
// leave too few waves per SIMD to hide the probes' latency.)
__global__ __launch_bounds__(256)
void k_icp_nn_grid(const float* __restrict__ src, int ns, const GridEntry* __restrict__ table, const float4* __restrict__ node,
                   unsigned mask, int shift, float inv_cell, const IcpState* __restrict__ st, float tau,
                   float* __restrict__ out_d2, int* __restrict__ out_idx) {
    if (st->done) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    float px, py, pz;
    transform_point(st->T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px, py, pz);
    float bd; int bo;
    grid_nearest(table, node, mask, shift, inv_cell, px, py, pz, tau, bd, bo);
    out_d2[i] = bo == INT_MAX ? FLT_MAX : bd;
    out_idx[i] = bo == INT_MAX ? 0 : bo;
}

// Sum over the 64 lanes with DPP row operations on the two halves of the double (no LDS crossbar round trips, unlike
// __shfl_down): valid in lane 63.  A fixed order, like the shuffle tree it replaces.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v += dpp_f64<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v);    // row_half_mirror
    v += dpp_f64<0x140, 0xF>(v);    // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_f64<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3 (a disabled row receives +0)
    v += dpp_f64<0x143, 0xC>(v);    // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// The update of one iteration from the folded sums tot[] (thread 0 of the folding block): registration.cpp:361-411.
// (iter, prev_rmse, Tcur: the state as the kernel read it at its start - no other launch writes it in between - so that the
// tail of the iteration does not begin with another round trip to memory)
// REF (reference-order accumulation, below): tot[] holds float sums; in point-to-point mode tot[2..4] / tot[5..7] are the two
// MEANS (already divided, registration.cpp:380-381) and tot[8..16] the centred cross-covariance of :383-386.
template <int MODE, bool REF = false>
__device__ void icp_update(const double* tot, int ns, IcpState* st, int fixed_iterations, int iter, float prev_rmse, const float* Tcur, float* solve_ws /* LDS, 54 words */) {
    const int n_corr = (int)(tot[0] + 0.5);
    st->iter = iter + 1;
    st->n_corr = n_corr;
    if (n_corr < 3) {  // registration.cpp:361 — break, keeping the previous result
        if (!fixed_iterations) st->done = 1;
        return;
    }
    float delta[16];
    for (int i = 0; i < 16; ++i) delta[i] = 0.f;
    delta[0] = delta[5] = delta[10] = delta[15] = 1.f;
    if (MODE == 0) {
        float ATA[36], nb[6], x[6];
        int k = 2;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) { float val = (float)tot[k++]; ATA[a * 6 + b] = val; ATA[b * 6 + a] = val; }
        for (int a = 0; a < 6; ++a) nb[a] = -(float)tot[k++];
        dl::ldlt6_solve(ATA, nb, x, solve_ws);
        dl::Mat3 dR = dl::euler_xyz(x[0], x[1], x[2]);
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) delta[c * 4 + r] = dl::el(dR, r, c);
        delta[12] = x[3]; delta[13] = x[4]; delta[14] = x[5];
    } else {
        const double n = (double)n_corr;
        double sm[3] = {tot[2], tot[3], tot[4]};
        double tm[3] = {tot[5], tot[6], tot[7]};
        if (!REF) for (int a = 0; a < 3; ++a) { sm[a] /= n; tm[a] /= n; }
        dl::Mat3 H;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) dl::el(H, a, b) = REF ? (float)tot[8 + a * 3 + b] : (float)(tot[8 + a * 3 + b] - n * sm[a] * tm[b]);
        dl::Mat3 dR = dl::kabsch_rotation(H);
        float smf[3] = {(float)sm[0], (float)sm[1], (float)sm[2]};
        float tmf[3] = {(float)tm[0], (float)tm[1], (float)tm[2]};
        float rx, ry, rz;
        dl::mulv3(dR, smf[0], smf[1], smf[2], rx, ry, rz);
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) delta[c * 4 + r] = dl::el(dR, r, c);
        delta[12] = tmf[0] - rx; delta[13] = tmf[1] - ry; delta[14] = tmf[2] - rz;
    }
    float Tn[16];
    dl::mul44(delta, Tcur, Tn);
    for (int i = 0; i < 16; ++i) { st->T[i] = Tn[i]; st->res_T[i] = Tn[i]; }
    const float rmse = sqrtf((float)tot[1] / (float)n_corr);
    st->rmse = rmse;
    st->fitness = (float)n_corr / (float)ns;
    st->applied = iter + 1;
    st->last_n_corr_applied = n_corr;
    if (!fixed_iterations && iter > 0 && fabsf(prev_rmse - rmse) < 1e-6f) st->done = 1;  // registration.cpp:406
}


// MODE 0: point-to-plane (21 upper-triangular JtJ + 6 Jtr), MODE 1: point-to-point moments,
// MODE 2: outputs only (no accumulation beyond count / error).
// One launch per iteration: every block reduces its points to one slab (fixed order: per lane -> wave64 shuffles ->
// LDS); the block that finishes LAST (atomic ticket) folds all slabs in a fixed order, solves, and updates the state
// on the device — the result does not depend on which block that is.
template <int MODE, int ACC_PPT>   // ACC_PPT source points per thread (summed per lane in index order)
__global__ __launch_bounds__(256)
void k_icp_accumulate(const float* __restrict__ src, int ns, int ns_pad,
                      const float* __restrict__ tgt, const float* __restrict__ tgt_normals,
                      const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                      int nsplit, const float* __restrict__ pd2, const int* __restrict__ pchunk, int direct,
                      IcpState* st, float tau_accept, int fixed_iterations,
                      double* slabs, unsigned* ticket,
                      int* __restrict__ out_corr, float* __restrict__ out_d2, uint8_t* __restrict__ out_acc) {
    if (st->done) return;
    const int iter0 = st->iter; const float rmse0 = st->rmse;
    double v[ACC_NV];
#pragma unroll
    for (int k = 0; k < ACC_NV; ++k) v[k] = 0.0;
    float T[12];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) T[c * 3 + r] = st->T[c * 4 + r];
    const float Tb[4] = {st->T[3], st->T[7], st->T[11], st->T[15]};     // bottom row (0 0 0 1 unless the caller's start pose says otherwise)
#pragma unroll 1
    for (int q = 0; q < ACC_PPT; ++q) {
        const int i = blockIdx.x * (256 * ACC_PPT) + q * 256 + threadIdx.x;
        if (i >= ns) continue;
        const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
        // p = R*s + t with each row evaluated as r0*sx + (r1*sy + r2*sz), then + t (see transform_point)
        const float px = (T[0] * sx + (T[3] * sy + T[6] * sz)) + T[9];
        const float py = (T[1] * sx + (T[4] * sy + T[7] * sz)) + T[10];
        const float pz = (T[2] * sx + (T[5] * sy + T[8] * sz)) + T[11];
        float best = FLT_MAX; int bc = 0;
        for (int s = 0; s < nsplit; ++s) {
            float d = pd2[(size_t)s * ns_pad + i];
            int c = pchunk[(size_t)s * ns_pad + i];
            if (d < best) { best = d; bc = c; }
        }
        int idx = 0;
        if (direct) {   // pruned / grid search: (best, bc) are the final (d2, target index)
            idx = bc;
        } else if (best < FLT_MAX) {
            idx = bc;
#pragma unroll
            for (int t = NN_CH - 1; t >= 0; --t) {
                float dx = px - tx[bc + t], dy = py - ty[bc + t], dz = pz - tz[bc + t];
                float d2 = dx * dx + (dy * dy + dz * dz);
                if (d2 == best) idx = bc + t;
            }
        }
        const bool acc = best <= tau_accept;
        if (out_corr) out_corr[i] = idx;
        if (out_d2) out_d2[i] = best;
        if (out_acc) out_acc[i] = acc ? 1 : 0;
        if (!acc) continue;
        v[0] += 1.0; v[1] += (double)best;
        const float qx = tgt[3 * idx], qy = tgt[3 * idx + 1], qz = tgt[3 * idx + 2];
        if (MODE == 0) {
            const float nx = tgt_normals[3 * idx], ny = tgt_normals[3 * idx + 1], nz = tgt_normals[3 * idx + 2];
            const float J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
            const float ex = px - qx, ey = py - qy, ez = pz - qz;
            const float r = ex * nx + (ey * ny + ez * nz);
            int k = 2;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) v[k++] += (double)(J[a] * J[b]);
#pragma unroll
            for (int a = 0; a < 6; ++a) v[k++] += (double)(J[a] * r);
        } else if (MODE == 1) {
            const double P[3] = {px, py, pz}, Q[3] = {qx, qy, qz};
            v[2] += P[0]; v[3] += P[1]; v[4] += P[2];
            v[5] += Q[0]; v[6] += Q[1]; v[7] += Q[2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) v[8 + a * 3 + b] += P[a] * Q[b];
        }
    }
    constexpr int NV = MODE == 0 ? 29 : (MODE == 1 ? 17 : 2);
    __shared__ double red[8][ACC_NV];
    __shared__ double tot[ACC_NV];
    __shared__ float solve_ws[56];
    __shared__ bool is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum_lane63(v[k]);
        if (lane == 63) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < ACC_NV) {
        double s = 0.0;
        if (threadIdx.x < NV) s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        slabs[(size_t)blockIdx.x * ACC_NV + threadIdx.x] = s;
    }
    // ---- last block: fold, solve, update -----------------------------------------------------------------------
    // Release by the wave that wrote the slab (wave 0; thread 0 then moves the ticket), acquire by the block that folds.
    // __threadfence() by every thread was a write-back AND an invalidate of the XCD's L2 (buffer_wbl2 + buffer_inv) from all
    // four waves of all 196 blocks - invalidating the target lines the blocks still running were gathering.
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int nblocks = gridDim.x;
    const int vv = threadIdx.x & 31, g = threadIdx.x >> 5;
    {
        // slabs were written by other blocks of this launch: the agent-scope fence above makes them visible to plain loads
        // sixteen slab rows per thread and round, all loads of a round in flight together (the fold is a chain of round trips:
        // 196 blocks at 200k points were 7 rounds of 4 loads, now 2 rounds of 16); fixed order, as before
        const double* sl = slabs;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int b2 = g; b2 < nblocks; b2 += 128) {
            double a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = b2 + 8 * j < nblocks ? sl[(size_t)(b2 + 8 * j) * ACC_NV + vv] : 0.0;
#pragma unroll
            for (int j = 0; j < 16; j += 4) { s0 += a[j]; s1 += a[j + 1]; s2 += a[j + 2]; s3 += a[j + 3]; }
        }
        red[g][vv] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (threadIdx.x < ACC_NV)
        tot[vv] = ((red[0][vv] + red[1][vv]) + (red[2][vv] + red[3][vv])) + ((red[4][vv] + red[5][vv]) + (red[6][vv] + red[7][vv]));
    __syncthreads();
    if (threadIdx.x != 0) return;
    *ticket = 0u;   // ready for the next launch (stream order)
    if (MODE == 2) st->n_corr = (int)(tot[0] + 0.5);
    else {
        float T16[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) { T16[c * 4] = T[c * 3]; T16[c * 4 + 1] = T[c * 3 + 1]; T16[c * 4 + 2] = T[c * 3 + 2]; T16[c * 4 + 3] = Tb[c]; }
        icp_update<MODE>(tot, ns, st, fixed_iterations, iter0, rmse0, T16, solve_ws);
    }
}

// ---- reference-order accumulation (round 4; TDV_ICP_ACCUMULATE_REFERENCE) ------------------------------------------------
// The CPU path adds every accepted correspondence to its sums one after the other, in float, in ascending source index
// (/root/reference/src/registration.cpp:340-341 n_corr and total_error, :353-354 ATA and ATb; point-to-point: the two means
// :376-381, then the centred cross-covariance :383-386).  A float sum depends on its order, so the f64 tree above agrees with
// it to ~1e-7 only - which once moved a translation by 1.27e-6 m on a 5-iteration C5 instance (round 3's log) and could, in
// principle, flip the stopping rule |delta rmse| < 1e-6 (:406).  In this mode the sums ARE the reference's, bit for bit.
//
// What bounds it: a sum in the reference's rounding is ONE chain of dependent float additions - lane k of one wave owns
// accumulator k, and a wave issues a VALU instruction every 4 cycles whatever else the chip does.  Everything is arranged so
// that this wave executes one v_add_f32 and a quarter of a 16-byte LDS read per ACCEPTED correspondence and nothing else:
//   k_icp_flags / k_icp_scan_counts / k_icp_rows   (whole chip) the accepted correspondences, in source order, as dense 32-byte
//                records {d2, J[6], r} resp. {d2, p, q}: block counts, their exclusive scan, then every block writes its
//                records at its offset (order-preserving compaction: rejected points - 3 of 4 at the pipeline's 0.4-voxel
//                threshold - cost the chain nothing).  n_corr is the scan's total.
//   k_icp_fold_ref  ONE workgroup of 9 waves.  Waves 1-8 are loaders, one record slot per lane: they fetch tile t + 4 (512
//                records) into a 3-deep register ring, expand tile t + 1 into the 28 per-accumulator term rows
//                (J[a] * J[b], J[a] * r: the products the CPU forms, unfused) in LDS, double buffered; wave 0 adds tile t, lane
//                k walking row k in order.  A first version streamed 28 precomputed floats per point: 924 us per fold at
//                200k points, bound by what ONE CU can pull from HBM (~11 B per cycle, MI355X_MICROARCH.md) - hence the
//                8-float records and the expansion on the CU.  Then the one-thread solve / update / stopping rule of the
//                tree mode, on float sums.
// Selectable per ctx like the search; the f64 tree stays the default.
constexpr int REF_NR_PLANE = 28;     // term rows, point-to-plane: d2, 21 upper-triangular J^T J, 6 J^T r
constexpr int REF_NR_POINT = 9;      // term rows, point-to-point: pass 0 d2, p.xyz, q.xyz (7); pass 1 the 9 centred products
constexpr int REF_TILE = 512;        // records per LDS tile
constexpr int REF_LD = REF_TILE + 4; // row pitch in LDS (floats): lane k's 16-byte reads start 4 banks after lane k-1's
constexpr int REF_RING = 3;          // register stages of a loader lane (tiles in flight ahead of the expansion)

// nearest target of source i from the search's output: (best d2, index) - what k_icp_accumulate does in line
__device__ __forceinline__ void resolve_nn(int i, int ns_pad, int nsplit, const float* __restrict__ pd2, const int* __restrict__ pchunk, int direct,
                                           const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                                           float px, float py, float pz, float& best, int& idx) {
    best = FLT_MAX; int bc = 0;
    for (int s = 0; s < nsplit; ++s) {
        const float d = pd2[(size_t)s * ns_pad + i];
        const int c = pchunk[(size_t)s * ns_pad + i];
        if (d < best) { best = d; bc = c; }
    }
    idx = 0;
    if (direct) idx = bc;
    else if (best < FLT_MAX) {
        idx = bc;
#pragma unroll
        for (int t = NN_CH - 1; t >= 0; --t) {
            const float dx = px - tx[bc + t], dy = py - ty[bc + t], dz = pz - tz[bc + t];
            if (dx * dx + (dy * dy + dz * dz) == best) idx = bc + t;
        }
    }
}

// accepted correspondences per block of 256 source points
__global__ __launch_bounds__(256)
void k_icp_flags(int ns, int ns_pad, int nsplit, const float* __restrict__ pd2, const IcpState* __restrict__ st, float tau_accept, int* __restrict__ cnt) {
    if (st->done) return;
    __shared__ int s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    float best = FLT_MAX;
    if (i < ns) for (int s = 0; s < nsplit; ++s) best = fminf(best, pd2[(size_t)s * ns_pad + i]);
    const unsigned long long m = __ballot(i < ns && best <= tau_accept);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_c, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_c;
}

// off[b] = cnt[0] + ... + cnt[b - 1], off[nblocks] = the total (n_corr); one workgroup, chunks of 1024 counts
__global__ __launch_bounds__(1024)
void k_icp_scan_counts(const int* __restrict__ cnt, int nblocks, const IcpState* __restrict__ st, int* __restrict__ off) {
    if (st->done) return;
    __shared__ int s_w[16], s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int c = b < nblocks ? cnt[b] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(inc, o, 64); if (lane >= o) inc += a; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int base = s_carry;
        for (int w = 0; w < wave; ++w) base += s_w[w];
        if (b < nblocks) off[b] = base + inc - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[nblocks] = s_carry;
}

// the accepted correspondences as records rec[2 * pos], rec[2 * pos + 1] (pos ascending with the source index):
//   point-to-plane {d2, J0, J1, J2 | J3, J4, J5, r}     point-to-point {d2, px, py, pz | qx, qy, qz, 0}
template <int MODE>
__global__ __launch_bounds__(256)
void k_icp_rows(const float* __restrict__ src, int ns, int ns_pad,
                const float* __restrict__ tgt, const float* __restrict__ tgt_normals,
                const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                int nsplit, const float* __restrict__ pd2, const int* __restrict__ pchunk, int direct,
                const IcpState* __restrict__ st, float tau_accept, const int* __restrict__ off, float4* __restrict__ rec) {
    if (st->done) return;
    __shared__ int s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool acc = false; float best = FLT_MAX; int idx = 0; float px = 0.f, py = 0.f, pz = 0.f;
    if (i < ns) {
        transform_point(st->T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px, py, pz);
        resolve_nn(i, ns_pad, nsplit, pd2, pchunk, direct, tx, ty, tz, px, py, pz, best, idx);
        acc = best <= tau_accept;
    }
    const unsigned long long m = __ballot(acc);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    if (!acc) return;
    int pos = off[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    const float qx = tgt[3 * idx], qy = tgt[3 * idx + 1], qz = tgt[3 * idx + 2];
    if (MODE == 0) {
        const float nx = tgt_normals[3 * idx], ny = tgt_normals[3 * idx + 1], nz = tgt_normals[3 * idx + 2];
        const float ex = px - qx, ey = py - qy, ez = pz - qz;
        rec[2 * (size_t)pos] = make_float4(best, py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx);      // J = [p x n | n], registration.cpp:346-349
        rec[2 * (size_t)pos + 1] = make_float4(nx, ny, nz, ex * nx + (ey * ny + ez * nz));                      // r = (p - q) . n, :351
    } else {
        rec[2 * (size_t)pos] = make_float4(best, px, py, pz);
        rec[2 * (size_t)pos + 1] = make_float4(qx, qy, qz, 0.f);
    }
}

// lane's running sum += the entries [0, m) of its LDS row, in order (m a multiple of 4): REF_CHAIN_G x 16 bytes of LDS reads, then
// 4 x REF_CHAIN_G dependent adds with the waits counted down (the compiler's schedule, checked in the ISA).  Reading one group
// AHEAD of the adds (two register groups) was measured slower at every group size (173 us per fold at 200k points -> 237-325).
#ifndef REF_CHAIN_G
#define REF_CHAIN_G 8
#endif
__device__ __forceinline__ float ref_chain(float s, const float* __restrict__ row, int m) {
    constexpr int G = REF_CHAIN_G;
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(row);
    int q = 0;
    for (; q + G <= m / 4; q += G) {
        float4 a[G];
#pragma unroll
        for (int j = 0; j < G; ++j) a[j] = r4[q + j];
#pragma unroll
        for (int j = 0; j < G; ++j) { s += a[j].x; s += a[j].y; s += a[j].z; s += a[j].w; }
    }
    for (; q < m / 4; ++q) { const float4 a = r4[q]; s += a.x; s += a.y; s += a.z; s += a.w; }
    return s;
}

// the per-accumulator terms of one record into column `col` of an LDS tile (rows[k][col]); PASS 1 (point-to-point): centred products
template <int MODE, int PASS>
__device__ __forceinline__ void ref_expand(float (*rows)[REF_LD], int col, const float4 a, const float4 b, bool live, const float* __restrict__ means) {
    if (MODE == 0) {
        const float J[6] = {a.y, a.z, a.w, b.x, b.y, b.z};
        rows[0][col] = a.x;
        int k = 1;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = u; v < 6; ++v) rows[k++][col] = J[u] * J[v];       // J[v] * J[u] is the same float: ATA stays symmetric bit for bit
#pragma unroll
        for (int u = 0; u < 6; ++u) rows[k++][col] = J[u] * b.w;
    } else if (PASS == 0) {
        rows[0][col] = a.x; rows[1][col] = a.y; rows[2][col] = a.z; rows[3][col] = a.w; rows[4][col] = b.x; rows[5][col] = b.y; rows[6][col] = b.z;
    } else {
        const float P[3] = {a.y - means[0], a.z - means[1], a.w - means[2]}, Q[3] = {b.x - means[3], b.y - means[4], b.z - means[5]};
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) rows[u * 3 + v][col] = live ? P[u] * Q[v] : 0.f;      // registration.cpp:385 (a padding slot adds +0)
    }
}

template <int MODE, int PASS>
__device__ __forceinline__ float ref_fold_pass(const float4* __restrict__ rec, int n_corr, float (*buf)[REF_NR_PLANE][REF_LD], const float* __restrict__ means) {
    constexpr int NSUM = MODE == 0 ? REF_NR_PLANE : (PASS == 0 ? 7 : 9);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool loader = wave >= 1;
    const int slot = (int)threadIdx.x - 64;                   // a loader lane's record slot inside a tile
    const int ntiles = (n_corr + REF_TILE - 1) / REF_TILE;
    float ring[REF_RING][8];                                   // a loader lane's records in flight (compile-time indices only)
    float s = 0.f;
#define TDV_REF_ISSUE(X, D) do { const long long pos__ = (long long)(X) * REF_TILE + slot;                                                   \
        float4 a__ = make_float4(0.f, 0.f, 0.f, 0.f), b__ = a__;           /* zeros past the end: their terms are +0 */                       \
        if ((X) < ntiles && pos__ < n_corr) { a__ = rec[2 * pos__]; b__ = rec[2 * pos__ + 1]; }                                               \
        ring[D][0] = a__.x; ring[D][1] = a__.y; ring[D][2] = a__.z; ring[D][3] = a__.w; ring[D][4] = b__.x; ring[D][5] = b__.y; ring[D][6] = b__.z; ring[D][7] = b__.w; } while (0)
#define TDV_REF_EXPAND(X, D) do { if ((X) < ntiles) ref_expand<MODE, PASS>(buf[(X) & 1], slot, make_float4(ring[D][0], ring[D][1], ring[D][2], ring[D][3]),   \
        make_float4(ring[D][4], ring[D][5], ring[D][6], ring[D][7]), (long long)(X) * REF_TILE + slot < n_corr, means); } while (0)
    if (loader) {
        TDV_REF_ISSUE(0, 0); TDV_REF_ISSUE(1, 1); TDV_REF_ISSUE(2, 2);
        TDV_REF_EXPAND(0, 0);
        TDV_REF_ISSUE(3, 0);
    }
    __syncthreads();
    // period x: loaders expand tile x + 1 out of ring[(x + 1) % 3] and refill that stage with tile x + 4; wave 0 adds tile x.
    // Unrolled by the ring's depth so that every register stage is a compile-time index.
    static_assert(REF_RING == 3, "the period loop below is written for three stages");
    for (int t = 0; t < ntiles; t += REF_RING) {
        if (loader) { TDV_REF_EXPAND(t + 1, 1); TDV_REF_ISSUE(t + 4, 1); }
        else if (wave == 0 && lane < NSUM && t < ntiles) s = ref_chain(s, buf[t & 1][lane], (min(REF_TILE, n_corr - t * REF_TILE) + 3) & ~3);
        __syncthreads();
        if (loader) { TDV_REF_EXPAND(t + 2, 2); TDV_REF_ISSUE(t + 5, 2); }
        else if (wave == 0 && lane < NSUM && t + 1 < ntiles) s = ref_chain(s, buf[(t + 1) & 1][lane], (min(REF_TILE, n_corr - (t + 1) * REF_TILE) + 3) & ~3);
        __syncthreads();
        if (loader) { TDV_REF_EXPAND(t + 3, 0); TDV_REF_ISSUE(t + 6, 0); }
        else if (wave == 0 && lane < NSUM && t + 2 < ntiles) s = ref_chain(s, buf[(t + 2) & 1][lane], (min(REF_TILE, n_corr - (t + 2) * REF_TILE) + 3) & ~3);
        __syncthreads();
    }
#undef TDV_REF_ISSUE
#undef TDV_REF_EXPAND
    return s;
}

template <int MODE>
__global__ __launch_bounds__(64 + REF_TILE)
void k_icp_fold_ref(const float4* __restrict__ rec, const int* __restrict__ d_total, int ns, IcpState* st, int fixed_iterations) {
    if (st->done) return;
    __shared__ __attribute__((aligned(16))) float buf[2][REF_NR_PLANE][REF_LD];      // (point-to-point uses 9 of the 28 rows)
    __shared__ double tot[ACC_NV];
    __shared__ float means[6];
    __shared__ float solve_ws[56];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_corr = *d_total;
    if (threadIdx.x < ACC_NV) tot[threadIdx.x] = 0.0;
    __syncthreads();
    float s = ref_fold_pass<MODE, 0>(rec, n_corr, buf, means);
    if (wave == 0) {
        if (MODE == 0) { if (lane < REF_NR_PLANE) tot[1 + lane] = (double)s; }
        else if (lane == 0) tot[1] = (double)s;
        else if (lane < 7) means[lane - 1] = s / static_cast<float>(n_corr);          // registration.cpp:380-381 (n_corr < 3: unused below)
    }
    __syncthreads();
    if (MODE == 1) {
        s = ref_fold_pass<MODE, 1>(rec, n_corr, buf, means);
        if (wave == 0 && lane < 9) tot[8 + lane] = (double)s;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    tot[0] = (double)n_corr;
    if (MODE == 1) for (int a = 0; a < 6; ++a) tot[2 + a] = (double)means[a];
    icp_update<MODE, true>(tot, ns, st, fixed_iterations, st->iter, st->rmse, st->T, solve_ws);
}

// ---- the whole loop in ONE launch, for small problems (round 3) ----------------------------------------------------------
// A 400 x 400 instance (config C5's size) spent 13 iterations' worth of launches - two per iteration, each 9-12 us of pure
// latency - and a state read-back per burst: 140 us of an instance's 340 us of kernels.  Here one workgroup of 16 waves keeps
// the target in LDS and runs search, normal equations, solve, update and the stopping rule for every iteration itself, then
// stores the final state straight into pinned host memory.
//  * search: one lane per source point scans the targets in LDS in ascending order with strict < and the scan's expression
//    d2 = dx*dx + (dy*dy + dz*dz): the scan's answer.
//  * accumulation: exactly k_icp_accumulate<MODE, 1>'s tree - 256 consecutive points form a block whose four wave sums (DPP) are
//    added as (w0 + w1) + (w2 + w3) into a slab, slabs folded in the same pattern - so a call gives the same bits whichever path
//    its size selects.
constexpr int SM_MAX_N = 2048;           // sources and targets the one-launch loop takes
constexpr long long SM_MAX_PAIRS_SINGLE = 1ll << 18;    // ... in a single call (tools/studies/icp_small_probe.py)
constexpr long long SM_MAX_PAIRS_BATCH = 1ll << 20;     // ... per problem of a batch
// A grid of several workgroups runs one problem each (the batch's small instances against the shared model): problem b takes the
// source points [src_off[b], src_off[b + 1]) of src0 and the states st_in[b] / st_out[b]; src_off == nullptr: one problem.
// 1,024 lanes and room for 2,048 x 2,048 points.
template <int MODE, int SM_THREADS, int SM_CAP, bool REF = false>   // REF: reference-order accumulation (see k_icp_fold_ref), tile by tile in LDS
__global__ __launch_bounds__(SM_THREADS)
void k_icp_small(const float* __restrict__ src0, int ns0, const int* __restrict__ src_off, const float* __restrict__ tgt, const float* __restrict__ tgt_normals, int nt,
                 const IcpState* __restrict__ st_in0, float tau_accept, int max_iterations, int fixed_iterations,
                 IcpState* __restrict__ st_out0, IcpState* __restrict__ st_host) {
    const int prob = blockIdx.x;
    const float* __restrict__ src = src_off ? src0 + (size_t)src_off[prob] * 3 : src0;
    const int ns = src_off ? src_off[prob + 1] - src_off[prob] : ns0;
    const IcpState* __restrict__ st_in = st_in0 + prob;
    IcpState* __restrict__ st_out = st_out0 + prob;
    if (ns == 0) { if (threadIdx.x == 0) *st_out = *st_in; return; }   // (an instance without points: the caller ignores its state)
    __shared__ __attribute__((aligned(16))) float tx[SM_CAP], ty[SM_CAP], tz[SM_CAP];
    __shared__ float sbest[SM_CAP];
    __shared__ int sidx[SM_CAP];
    __shared__ double red[SM_THREADS / 64][ACC_NV];       // wave sums, four per virtual block
    __shared__ double slab[SM_CAP / 256][ACC_NV];
    __shared__ double fold[8][ACC_NV];
    __shared__ double tot[ACC_NV];
    __shared__ float solve_ws[56];
    __shared__ IcpState st;
    constexpr int NRR = MODE == 0 ? REF_NR_PLANE : REF_NR_POINT;
    __shared__ __attribute__((aligned(16))) float rrows[REF ? NRR : 1][REF_LD];
    __shared__ float rmeans[6];
    static_assert(!REF || SM_THREADS >= REF_TILE, "one point per thread and tile");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt4 = (nt + 3) / 4;                          // targets in chunks of four, the last one padded with +inf (never a minimum)
    for (int j = threadIdx.x; j < nt4 * 4; j += SM_THREADS) {
        const bool in = j < nt;
        tx[j] = in ? tgt[3 * j] : INFINITY; ty[j] = in ? tgt[3 * j + 1] : INFINITY; tz[j] = in ? tgt[3 * j + 2] : INFINITY;
    }
    if (threadIdx.x == 0) st = *st_in;
    __syncthreads();
    constexpr int NV = MODE == 0 ? 29 : 17;
    const int nblocks = (ns + 255) / 256;
    for (int it = 0; it < max_iterations; ++it) {
        float T[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) T[k] = st.T[k];
        const int iter0 = st.iter; const float rmse0 = st.rmse;
        // (a) nearest target of every source point: one LANE per source, the targets read from LDS at a wave-uniform address
        // (a broadcast, no bank conflict) in ascending order with strict <: the scan's answer.  (One WAVE per source with a
        // cross-lane (d2, index) minimum at the end was 3x slower at 400 x 400: six dependent shuffle rounds per source.)
        for (int i = threadIdx.x; i < ns; i += SM_THREADS) {
            float px, py, pz;
            transform_point(T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px, py, pz);
            // four targets per step (three 16-byte LDS reads at a wave-uniform address): the running minimum and the first CHUNK
            // that reached it (strict <), then the lowest target of that chunk with that distance - the scan kernel's two-level argmin
            float best = FLT_MAX; int bc = 0;
            const float4* __restrict__ X4 = reinterpret_cast<const float4*>(tx);
            const float4* __restrict__ Y4 = reinterpret_cast<const float4*>(ty);
            const float4* __restrict__ Z4 = reinterpret_cast<const float4*>(tz);
#pragma unroll 2
            for (int c = 0; c < nt4; ++c) {
                const float4 X = X4[c], Y = Y4[c], Z = Z4[c];
                const float ax = px - X.x, ay = py - Y.x, az = pz - Z.x, bx = px - X.y, by = py - Y.y, bz = pz - Z.y;
                const float cx = px - X.z, cy = py - Y.z, cz = pz - Z.z, ex = px - X.w, ey = py - Y.w, ez = pz - Z.w;
                const float d0 = ax * ax + (ay * ay + az * az), d1 = bx * bx + (by * by + bz * bz);
                const float d2 = cx * cx + (cy * cy + cz * cz), d3 = ex * ex + (ey * ey + ez * ez);
                const float m = fminf(fminf(d0, d1), fminf(d2, d3));
                if (m < best) { best = m; bc = c; }
            }
            int bi = 0;
            if (best < FLT_MAX) {
                bi = 4 * bc;
#pragma unroll
                for (int t = 3; t >= 0; --t) {
                    const int j = 4 * bc + t;
                    const float dx = px - tx[j], dy = py - ty[j], dz = pz - tz[j];
                    if (dx * dx + (dy * dy + dz * dz) == best) bi = j;
                }
            }
            sbest[i] = best; sidx[i] = best < FLT_MAX ? bi : 0;
        }
        __syncthreads();
        if constexpr (REF) {
            // (b') the reference's sums: tiles of REF_TILE points - every thread expands its point's record into the per-accumulator term
            // rows in LDS (k_icp_fold_ref's ref_expand; a rejected point's terms are +0: s + 0 == s), lane k of wave 0 adds row k in
            // index order, the chain continuing from tile to tile
            int n_acc = 0;
            constexpr int NPASS = MODE == 0 ? 1 : 2;
            for (int pass = 0; pass < NPASS; ++pass) {
                const int nsum = MODE == 0 ? REF_NR_PLANE : (pass == 0 ? 7 : 9);
                float s = 0.f;
                for (int base = 0; base < ns; base += REF_TILE) {
                    bool acc = false;
                    if (threadIdx.x < REF_TILE) {
                        const int i = base + threadIdx.x;
                        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
                        if (i < ns) {
                            const float best = sbest[i];
                            acc = best <= tau_accept;
                            if (acc) {
                                float px, py, pz;
                                transform_point(T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px, py, pz);
                                const int idx = sidx[i];
                                const float qx = tgt[3 * idx], qy = tgt[3 * idx + 1], qz = tgt[3 * idx + 2];
                                if (MODE == 0) {
                                    const float nx = tgt_normals[3 * idx], ny = tgt_normals[3 * idx + 1], nz = tgt_normals[3 * idx + 2];
                                    const float ex = px - qx, ey = py - qy, ez = pz - qz;
                                    ra = make_float4(best, py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx);
                                    rb = make_float4(nx, ny, nz, ex * nx + (ey * ny + ez * nz));
                                } else { ra = make_float4(best, px, py, pz); rb = make_float4(qx, qy, qz, 0.f); }
                            }
                        }
                        if (pass == 0) ref_expand<MODE, 0>(rrows, threadIdx.x, ra, rb, acc, rmeans);
                        else ref_expand<MODE, 1>(rrows, threadIdx.x, ra, rb, acc, rmeans);
                    }
                    const int c = __syncthreads_count(acc);
                    if (pass == 0) n_acc += c;
                    if (wave == 0 && lane < nsum) s = ref_chain(s, rrows[lane], (min(REF_TILE, ns - base) + 3) & ~3);
                    __syncthreads();
                }
                if (wave == 0 && lane < nsum) {
                    if (MODE == 0) tot[1 + lane] = (double)s;
                    else if (pass == 0) { if (lane == 0) tot[1] = (double)s; else rmeans[lane - 1] = s / static_cast<float>(n_acc); }
                    else tot[8 + lane] = (double)s;
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                tot[0] = (double)n_acc;
                if (MODE == 1) for (int a = 0; a < 6; ++a) tot[2 + a] = (double)rmeans[a];
                icp_update<MODE, true>(tot, ns, &st, fixed_iterations, iter0, rmse0, T, solve_ws);
            }
            __syncthreads();
            if (st.done) break;
            continue;
        }
        // (b) slabs: virtual block B = points [256 B, 256 B + 256), one point per lane
        for (int B0 = 0; B0 < nblocks; B0 += SM_THREADS / 256) {
            const int B = B0 + (threadIdx.x >> 8);
            const int i = B * 256 + (threadIdx.x & 255);
            double v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = 0.0;
            if (B < nblocks && i < ns) {
                float px, py, pz;
                transform_point(T, src[3 * i], src[3 * i + 1], src[3 * i + 2], px, py, pz);
                const float best = sbest[i]; const int idx = sidx[i];
                if (best <= tau_accept) {
                    v[0] = 1.0; v[1] = (double)best;
                    const float qx = tgt[3 * idx], qy = tgt[3 * idx + 1], qz = tgt[3 * idx + 2];
                    if (MODE == 0) {
                        const float nx = tgt_normals[3 * idx], ny = tgt_normals[3 * idx + 1], nz = tgt_normals[3 * idx + 2];
                        const float J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
                        const float ex = px - qx, ey = py - qy, ez = pz - qz;
                        const float r = ex * nx + (ey * ny + ez * nz);
                        int k = 2;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int b = a; b < 6; ++b) v[k++] = (double)(J[a] * J[b]);
#pragma unroll
                        for (int a = 0; a < 6; ++a) v[k++] = (double)(J[a] * r);
                    } else {
                        const double P[3] = {px, py, pz}, Q[3] = {qx, qy, qz};
                        v[2] = P[0]; v[3] = P[1]; v[4] = P[2];
                        v[5] = Q[0]; v[6] = Q[1]; v[7] = Q[2];
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int b = 0; b < 3; ++b) v[8 + a * 3 + b] = P[a] * Q[b];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const double s = wave_sum_lane63(v[k]);
                if (lane == 63) red[wave][k] = s;
            }
            __syncthreads();
            if (B < nblocks && (threadIdx.x & 255) < ACC_NV) {
                const int k = threadIdx.x & 255, w0 = (threadIdx.x >> 8) * 4;
                slab[B][k] = k < NV ? (red[w0][k] + red[w0 + 1][k]) + (red[w0 + 2][k] + red[w0 + 3][k]) : 0.0;
            }
            __syncthreads();
        }
        // (c) fold, in k_icp_accumulate's pattern (group g of 32 threads takes slabs g, g + 8, ...: here at most one each)
        if (threadIdx.x < 256) {
            const int vv = threadIdx.x & 31, g = threadIdx.x >> 5;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            if (g < nblocks) s0 += slab[g][vv];
            fold[g][vv] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (threadIdx.x < ACC_NV) {
            const int vv = threadIdx.x;
            tot[vv] = ((fold[0][vv] + fold[1][vv]) + (fold[2][vv] + fold[3][vv])) + ((fold[4][vv] + fold[5][vv]) + (fold[6][vv] + fold[7][vv]));
        }
        __syncthreads();
        // (d) solve, update, stopping rule
        if (threadIdx.x == 0) icp_update<MODE>(tot, ns, &st, fixed_iterations, iter0, rmse0, T, solve_ws);
        __syncthreads();
        if (st.done) break;
    }
    if (threadIdx.x == 0) {
        *st_out = st;
        if (st_host) { *st_host = st; __threadfence_system(); }
    }
}

namespace {

struct NnPlan {
    int ns_pad, nt_pad, n_chunks, blocks_x, nsplit, chunks_per_split, acc_blocks, acc_ppt;
};

NnPlan make_plan(int ns, int nt) {
    NnPlan p;
    p.ns_pad = (int)align_up((size_t)ns, NN_SRC_PER_BLOCK);
    p.nt_pad = (int)align_up((size_t)nt, NN_CH);
    p.n_chunks = p.nt_pad / NN_CH;
    p.blocks_x = p.ns_pad / NN_SRC_PER_BLOCK;
    // aim for ~12k workgroups (48 per CU: measured best at 200k x 200k) but keep >= 16 chunks (256 targets) per split
    int want = (NN_WG_TARGET + p.blocks_x - 1) / p.blocks_x;
    int max_split = std::max(1, p.n_chunks / 16);
    p.nsplit = std::max(1, std::min(std::min(want, max_split), 64));
    p.chunks_per_split = (p.n_chunks + p.nsplit - 1) / p.nsplit;
    p.nsplit = (p.n_chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    static const int ppt_env = study_env("TDV_ICP_PPT") ? atoi(study_env("TDV_ICP_PPT")) : 0;
    // only 1, 2, 4 and 8 points per thread are instantiated (TDV_ACC below); anything else would size the grid for a kernel that is never launched
    p.acc_ppt = (kStudyBuild && (ppt_env == 1 || ppt_env == 2 || ppt_env == 4 || ppt_env == 8)) ? ppt_env : 4;   // 4: make_plan's default; the brute-force path uses 1
    p.acc_blocks = (ns + 256 * p.acc_ppt - 1) / (256 * p.acc_ppt);
    return p;
}

struct IcpBuffers {
    float *tx, *ty, *tz, *pd2; int* pchunk; double* slabs; IcpState* st; unsigned* ticket;
};

int alloc_buffers(tdv_ctx* ctx, const NnPlan& p, IcpBuffers& b) {
    float* soa = nullptr;
    TDV_TRY(ws_alloc(ctx, (size_t)3 * p.nt_pad, &soa));
    b.tx = soa; b.ty = soa + p.nt_pad; b.tz = soa + 2 * (size_t)p.nt_pad;
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * p.ns_pad, &b.pd2));
    TDV_TRY(ws_alloc(ctx, (size_t)p.nsplit * p.ns_pad, &b.pchunk));
    TDV_TRY(ws_alloc(ctx, (size_t)p.acc_blocks * ACC_NV, &b.slabs));
    TDV_TRY(ws_alloc(ctx, 1, &b.st));
    b.ticket = ctx->scan_ticket + 1;     // the ctx's persistent ticket words (zero between launches: the last workgroup resets it)
    return TDV_OK;
}

}  // namespace

int cell_grid_build(tdv_ctx* ctx, const float* d_tgt, int nt, float thr, CellGrid* g) {
    if (!ctx || !d_tgt || !g || nt <= 0) return TDV_ERR_BAD_ARG;
    *g = CellGrid{};
    g->thr = thr; g->n = nt;
    const float cell = GRID_CELL_FACTOR * thr;
    const float inv_cell = 1.0f / cell;
    if (!(thr > 0.f) || !(cell <= FLT_MAX) || !(inv_cell > 0.f) || !(inv_cell <= FLT_MAX)) return TDV_OK;   // usable = 0
    size_t size = 1024; int log2 = 10;
    while (size < (size_t)GRID_SLOTS_PER_POINT_X4 * (size_t)nt / 4) { size <<= 1; ++log2; }
    GridEntry* table; float4* node; int* flags;
    TDV_TRY(ws_alloc(ctx, size + GRID_PAD, &table));
    TDV_TRY(ws_alloc(ctx, (size_t)nt, &node));
    TDV_TRY(ws_alloc(ctx, 2, &flags));
    hipStream_t s = ctx->stream;
    TDV_HIP(ctx, hipMemsetAsync(table, 0xff, (size + GRID_PAD) * sizeof(GridEntry), s));      // key = empty, head = -1
    TDV_HIP(ctx, hipMemsetAsync(flags, 0, 8, s));
    k_grid_insert<<<(nt + 255) / 256, 256, 0, s>>>(d_tgt, nt, inv_cell, table, node, (unsigned)(size - 1), 64 - log2, flags);
    TDV_CHECK_LAUNCH(ctx);
    TDV_TRY(pin_reserve(ctx, 64));
    TDV_HIP(ctx, hipMemcpyAsync(ctx->pin, flags, 8, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    const int* h = reinterpret_cast<const int*>(ctx->pin);
    const int bad = h[0], cells = h[1];
    g->table = table; g->node = node; g->mask = (unsigned)(size - 1); g->shift = 64 - log2; g->inv_cell = inv_cell;
    static const int max_per_cell = study_env("TDV_GRID_MAX_PER_CELL") ? atoi(study_env("TDV_GRID_MAX_PER_CELL")) : GRID_MAX_PER_CELL;   // tuning knob
    g->usable = (!bad && cells > 0 && (long long)nt <= (long long)max_per_cell * cells) ? 1 : 0;
    return TDV_OK;
}

int icp_run_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, const float* d_tgt_normals, int nt,
                const float* T0, float thr, int max_iterations, int point_to_plane, int fixed_iterations,
                tdv_icp_result* out, const SortedCloud* tgt_sorted, const CellGrid* tgt_grid) {
    if (!ctx || !d_src || !d_tgt || !T0 || !out || ns < 0 || nt < 0 || max_iterations < 0) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    // result defaults: registration.cpp:309-311
    std::memcpy(out->T, T0, 64);
    out->fitness = 0.f; out->rmse = 0.f; out->iterations = 0; out->n_corr = 0;
    if (max_iterations == 0) return TDV_OK;
    if (ns == 0 || nt == 0) return TDV_OK;  // n_corr == 0 < 3 -> break at the first iteration
    const float tau = tau_le(thr);
    // search: brute force for small problems (the two Morton sorts cost more than they save), exact pruned walk
    // for large ones; both give the same correspondences bit for bit
    bool pruned = ctx->icp_search == TDV_ICP_SEARCH_PRUNED ||
                  (ctx->icp_search == TDV_ICP_SEARCH_AUTO && nt >= PRUNED_MIN_TARGETS && (double)ns * (double)nt >= PRUNED_MIN_PAIRS);
    if (!(tau < FLT_MAX)) pruned = false;   // unbounded threshold: keep the scan's handling of overflowing distances
    // hash grid: on request, or by size like the pruned walk; used when its build says the cells are small enough
    CellGrid cg{};
    if (tau < FLT_MAX && (ctx->icp_search == TDV_ICP_SEARCH_GRID || (ctx->icp_search == TDV_ICP_SEARCH_AUTO && pruned))) {
        if (tgt_grid && tgt_grid->n == nt && tgt_grid->thr == thr) cg = *tgt_grid;
        else TDV_TRY(cell_grid_build(ctx, d_tgt, nt, thr, &cg));
        if (cg.usable) pruned = true;           // results arrive in the pruned walk's format (one entry per source, original indices)
        else if (ctx->icp_search == TDV_ICP_SEARCH_GRID) pruned = true;   // fall back to the walk
    }
    ctx->last_icp_search = cg.usable ? TDV_ICP_SEARCH_GRID : (pruned ? TDV_ICP_SEARCH_PRUNED : TDV_ICP_SEARCH_BRUTE);
    NnPlan p = make_plan(ns, nt);
    if (pruned) p.nsplit = 1;
    else if (!study_env("TDV_ICP_PPT")) { p.acc_ppt = 1; p.acc_blocks = (ns + 255) / 256; }   // measured: 50k x 10k brute 7.6k vs 6.4k iters/s
    const bool ref_acc = ctx->icp_accumulate == TDV_ICP_ACCUMULATE_REFERENCE;
    TDV_TRY(pin_reserve(ctx, 2 * sizeof(IcpState)));
    IcpState* h = reinterpret_cast<IcpState*>(ctx->pin);
    std::memset(h, 0, sizeof(IcpState));
    std::memcpy(h->T, T0, 64); std::memcpy(h->res_T, T0, 64);
    hipStream_t s = ctx->stream;
    // small problems: the whole loop in one launch (k_icp_small), same bits as the launches below
    const bool small_off = getenv("TDV_ICP_SMALL") && atoi(getenv("TDV_ICP_SMALL")) == 0;   // A/B knob (read per call: the tests switch it)
    // (one workgroup walks all ns x nt pairs of an iteration: measured per iteration 22 us at 400 x 398 against 28 us for the two
    //  launches, but 65 us at 1,000 x 1,000 and 237 us at 2,000 x 2,000 against 25-27: a single call only takes this path while the pair
    //  count is small; a BATCH of such problems is another matter - there the grid is the parallelism, icp_small_batch_dev)
    if (!small_off && !pruned && !cg.usable && ns <= SM_MAX_N && nt <= SM_MAX_N && (long long)ns * nt <= SM_MAX_PAIRS_SINGLE) {
        IcpState* d_st;
        TDV_TRY(ws_alloc(ctx, 2, &d_st));
        IcpState* h_res = h + 1;                          // the kernel stores its final state here itself (pinned memory: no copy kernel)
        h_res->iter = -1;
        TDV_HIP(ctx, hipMemcpyAsync(d_st, h, sizeof(IcpState), hipMemcpyHostToDevice, s));
        {
            ScopedTimer tm(ctx, TDV_TIMER_ICP_NN);
            if (point_to_plane && d_tgt_normals) {
                if (ref_acc) k_icp_small<0, 1024, SM_MAX_N, true><<<1, 1024, 0, s>>>(d_src, ns, nullptr, d_tgt, d_tgt_normals, nt, d_st, tau, max_iterations, fixed_iterations, d_st + 1, h_res);
                else k_icp_small<0, 1024, SM_MAX_N><<<1, 1024, 0, s>>>(d_src, ns, nullptr, d_tgt, d_tgt_normals, nt, d_st, tau, max_iterations, fixed_iterations, d_st + 1, h_res);
            } else {
                if (ref_acc) k_icp_small<1, 1024, SM_MAX_N, true><<<1, 1024, 0, s>>>(d_src, ns, nullptr, d_tgt, nullptr, nt, d_st, tau, max_iterations, fixed_iterations, d_st + 1, h_res);
                else k_icp_small<1, 1024, SM_MAX_N><<<1, 1024, 0, s>>>(d_src, ns, nullptr, d_tgt, nullptr, nt, d_st, tau, max_iterations, fixed_iterations, d_st + 1, h_res);
            }
        }
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipStreamSynchronize(s));
        if (h_res->iter < 0) { snprintf(ctx->err, sizeof(ctx->err), "icp: the result did not reach the host"); return TDV_ERR_INTERNAL; }
        std::memcpy(out->T, h_res->res_T, 64);
        out->fitness = h_res->fitness; out->rmse = h_res->rmse; out->iterations = h_res->applied; out->n_corr = h_res->last_n_corr_applied;
        return TDV_OK;
    }
    IcpBuffers b;
    TDV_TRY(alloc_buffers(ctx, p, b));
    TDV_HIP(ctx, hipMemcpyAsync(b.st, h, sizeof(IcpState), hipMemcpyHostToDevice, s));
    SortedCloud st{};
    if (cg.usable) {
        // nothing to prepare
    } else if (pruned) {
        if (tgt_sorted && tgt_sorted->n == nt) st = *tgt_sorted;   // the batch orders the shared model once
        else TDV_TRY(spatial_sort_cloud(ctx, d_tgt, nt, st));
    } else {
        k_aos_to_soa_pad<<<(p.nt_pad + 255) / 256, 256, 0, s>>>(d_tgt, nt, p.nt_pad, INFINITY, b.tx, b.ty, b.tz);
    }
    TDV_CHECK_LAUNCH(ctx);
    const bool p2pl = point_to_plane && d_tgt_normals;
    const int direct = pruned ? 1 : 0;
    // reference-order accumulation: dense records of the accepted correspondences + one-workgroup ordered fold instead of k_icp_accumulate
    const int ref_blocks = (ns + 255) / 256;
    float4* ref_rec = nullptr; int *ref_cnt = nullptr, *ref_off = nullptr;
    if (ref_acc) {
        TDV_TRY(ws_alloc(ctx, (size_t)2 * ns, &ref_rec));
        TDV_TRY(ws_alloc(ctx, (size_t)ref_blocks, &ref_cnt));
        TDV_TRY(ws_alloc(ctx, (size_t)ref_blocks + 1, &ref_off));
    }
    const GridEntry* gtable = cg.usable ? reinterpret_cast<const GridEntry*>(cg.table) : nullptr;
    const float4* gnode = cg.usable ? reinterpret_cast<const float4*>(cg.node) : nullptr;
    const dim3 grid(p.blocks_x, p.nsplit);
    // iterations are enqueued in bursts between two looks at the state; once `done` is set the remaining launches of a burst
    // return at once.  The reference's stopping rule fires after 3-4 iterations in the pipeline's setting (0.4-voxel
    // threshold from a RANSAC start), so the first burst is short; fixed-iteration runs take long bursts.
    int it = 0;
    while (it < max_iterations) {
        const int poll = fixed_iterations ? 32 : (it == 0 ? 4 : 8);
        int burst = std::min(poll, max_iterations - it);
        for (int k = 0; k < burst; ++k) {
            {
                ScopedTimer tm(ctx, TDV_TIMER_ICP_NN);
                if (gtable)
                    k_icp_nn_grid<<<(ns + 255) / 256, 256, 0, s>>>(d_src, ns, gtable, gnode, cg.mask, cg.shift, cg.inv_cell, b.st, tau, b.pd2, b.pchunk);
                else if (pruned)
                    k_icp_nn_pruned<<<(ns + PN_WAVES * PN_PTS - 1) / (PN_WAVES * PN_PTS), 64 * PN_WAVES, 0, s>>>(
                        d_src, ns, st.sx, st.sy, st.sz, st.orig, nt, st.lbox, st.n_leaf, st.tbox, st.n_top, b.st, tau, b.pd2, b.pchunk);
                else
                    k_icp_nn_scan<<<grid, NN_BLOCK, 0, s>>>(d_src, ns, p.ns_pad, b.tx, b.ty, b.tz, p.n_chunks,
                                                            p.chunks_per_split, b.st, b.pd2, b.pchunk);
            }
            if (ref_acc) {
                k_icp_flags<<<ref_blocks, 256, 0, s>>>(ns, p.ns_pad, p.nsplit, b.pd2, b.st, tau, ref_cnt);
                k_icp_scan_counts<<<1, 1024, 0, s>>>(ref_cnt, ref_blocks, b.st, ref_off);
                if (p2pl) {
                    k_icp_rows<0><<<ref_blocks, 256, 0, s>>>(d_src, ns, p.ns_pad, d_tgt, d_tgt_normals, b.tx, b.ty, b.tz, p.nsplit, b.pd2, b.pchunk, direct, b.st, tau, ref_off, ref_rec);
                    k_icp_fold_ref<0><<<1, 64 + REF_TILE, 0, s>>>(ref_rec, ref_off + ref_blocks, ns, b.st, fixed_iterations);
                } else {
                    k_icp_rows<1><<<ref_blocks, 256, 0, s>>>(d_src, ns, p.ns_pad, d_tgt, nullptr, b.tx, b.ty, b.tz, p.nsplit, b.pd2, b.pchunk, direct, b.st, tau, ref_off, ref_rec);
                    k_icp_fold_ref<1><<<1, 64 + REF_TILE, 0, s>>>(ref_rec, ref_off + ref_blocks, ns, b.st, fixed_iterations);
                }
            } else if (p2pl) {
#define TDV_ACC1(MM, PP, NRM) k_icp_accumulate<MM, PP><<<p.acc_blocks, 256, 0, s>>>(d_src, ns, p.ns_pad, d_tgt, NRM, b.tx, b.ty, b.tz, \
                                   p.nsplit, b.pd2, b.pchunk, direct, b.st, tau, fixed_iterations, b.slabs, b.ticket, nullptr, nullptr, nullptr)
#ifdef TDV_STUDY
#define TDV_ACC(MM, NRM) do { if (p.acc_ppt == 8) TDV_ACC1(MM, 8, NRM); else if (p.acc_ppt == 4) TDV_ACC1(MM, 4, NRM); else if (p.acc_ppt == 2) TDV_ACC1(MM, 2, NRM); else TDV_ACC1(MM, 1, NRM); } while (0)
#else
#define TDV_ACC(MM, NRM) do { if (p.acc_ppt == 4) TDV_ACC1(MM, 4, NRM); else TDV_ACC1(MM, 1, NRM); } while (0)     // (2 and 8 points per thread: study build)
#endif
                TDV_ACC(0, d_tgt_normals);
            } else {
                TDV_ACC(1, nullptr);
            }
        }
        TDV_CHECK_LAUNCH(ctx);
        it += burst;
        TDV_HIP(ctx, hipMemcpyAsync(h, b.st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
        TDV_HIP(ctx, hipStreamSynchronize(s));
        if (h->done) break;
    }
    std::memcpy(out->T, h->res_T, 64);
    out->fitness = h->fitness; out->rmse = h->rmse; out->iterations = h->applied; out->n_corr = h->last_n_corr_applied;
    return TDV_OK;
}

int icp_small_max_points() { return SM_MAX_N; }
long long icp_small_max_pairs_batch() { return SM_MAX_PAIRS_BATCH; }

// icp_run_dev for n_prob small problems against one target in ONE launch: problem b = source points [d_src_off[b], d_src_off[b+1])
// of d_src (each at most icp_small_max_points(), as nt), start pose T0s[b] (host, column-major).  Results as icp_run_dev's, bit for
// bit (the same kernel).  One upload, one launch, one download.
int icp_small_batch_dev(tdv_ctx* ctx, const float* d_src, const int* d_src_off, int n_prob, const float* d_tgt, const float* d_tgt_normals, int nt,
                        const float* T0s, float thr, int max_iterations, int point_to_plane, tdv_icp_result* out, int ns_max) {
    if (!ctx || !d_src || !d_src_off || !d_tgt || !T0s || !out || n_prob < 0 || nt <= 0 || nt > SM_MAX_N || max_iterations < 0) return TDV_ERR_BAD_ARG;
    if (n_prob == 0) return TDV_OK;
    hipStream_t s = ctx->stream;
    const float tau = tau_le(thr);
    IcpState* d_st;
    TDV_TRY(ws_alloc(ctx, (size_t)2 * n_prob, &d_st));
    TDV_TRY(pin_reserve(ctx, (size_t)n_prob * sizeof(IcpState)));
    IcpState* h = reinterpret_cast<IcpState*>(ctx->pin);
    std::memset(h, 0, (size_t)n_prob * sizeof(IcpState));
    for (int b = 0; b < n_prob; ++b) { std::memcpy(h[b].T, T0s + 16 * (size_t)b, 64); std::memcpy(h[b].res_T, T0s + 16 * (size_t)b, 64); }
    TDV_HIP(ctx, hipMemcpyAsync(d_st, h, (size_t)n_prob * sizeof(IcpState), hipMemcpyHostToDevice, s));
    if (max_iterations > 0) {
        ScopedTimer tm(ctx, TDV_TIMER_ICP_NN);
        const bool p2pl = point_to_plane && d_tgt_normals;
        // (A quarter-size shape - 256 lanes, more problems resident at once - was measured against this one on C5's 1,024 instances in
        // round 3: 1.59 ms against 1.33 ms.  The pass lasts as long as its slowest problem and a lone workgroup iterates faster with
        // 16 waves; the variant is gone, profiles/r3/history keeps the numbers.)
        (void)ns_max;
        if (ctx->icp_accumulate == TDV_ICP_ACCUMULATE_REFERENCE) {
            if (p2pl) k_icp_small<0, 1024, SM_MAX_N, true><<<n_prob, 1024, 0, s>>>(d_src, 0, d_src_off, d_tgt, d_tgt_normals, nt, d_st, tau, max_iterations, 0, d_st + n_prob, nullptr);
            else k_icp_small<1, 1024, SM_MAX_N, true><<<n_prob, 1024, 0, s>>>(d_src, 0, d_src_off, d_tgt, nullptr, nt, d_st, tau, max_iterations, 0, d_st + n_prob, nullptr);
        } else {
            if (p2pl) k_icp_small<0, 1024, SM_MAX_N><<<n_prob, 1024, 0, s>>>(d_src, 0, d_src_off, d_tgt, d_tgt_normals, nt, d_st, tau, max_iterations, 0, d_st + n_prob, nullptr);
            else k_icp_small<1, 1024, SM_MAX_N><<<n_prob, 1024, 0, s>>>(d_src, 0, d_src_off, d_tgt, nullptr, nt, d_st, tau, max_iterations, 0, d_st + n_prob, nullptr);
        }
        TDV_CHECK_LAUNCH(ctx);
        TDV_HIP(ctx, hipMemcpyAsync(h, d_st + n_prob, (size_t)n_prob * sizeof(IcpState), hipMemcpyDeviceToHost, s));
    }
    TDV_HIP(ctx, hipStreamSynchronize(s));
    ctx->last_icp_search = TDV_ICP_SEARCH_BRUTE;
    for (int b = 0; b < n_prob; ++b) {
        std::memcpy(out[b].T, h[b].res_T, 64);
        out[b].fitness = h[b].fitness; out[b].rmse = h[b].rmse; out[b].iterations = h[b].applied; out[b].n_corr = h[b].last_n_corr_applied;
    }
    return TDV_OK;
}

int icp_correspondences_dev(tdv_ctx* ctx, const float* d_src, int ns, const float* d_tgt, int nt,
                            const float* T, float thr, IcpOutputs outs, int* n_corr) {
    if (!ctx || !d_src || !d_tgt || !T || ns <= 0 || nt <= 0) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    NnPlan p = make_plan(ns, nt);
    p.acc_ppt = 1; p.acc_blocks = (ns + 255) / 256;   // one point per thread in the outputs-only pass
    bool pruned = (ctx->icp_search == TDV_ICP_SEARCH_PRUNED || ctx->icp_search == TDV_ICP_SEARCH_GRID) && tau_le(thr) < FLT_MAX;
    CellGrid grid{};
    if (pruned && ctx->icp_search == TDV_ICP_SEARCH_GRID) TDV_TRY(cell_grid_build(ctx, d_tgt, nt, thr, &grid));
    ctx->last_icp_search = grid.usable ? TDV_ICP_SEARCH_GRID : (pruned ? TDV_ICP_SEARCH_PRUNED : TDV_ICP_SEARCH_BRUTE);
    if (pruned) p.nsplit = 1;
    IcpBuffers b;
    TDV_TRY(alloc_buffers(ctx, p, b));
    TDV_TRY(pin_reserve(ctx, sizeof(IcpState)));
    IcpState* h = reinterpret_cast<IcpState*>(ctx->pin);
    std::memset(h, 0, sizeof(IcpState));
    std::memcpy(h->T, T, 64);
    hipStream_t s = ctx->stream;
    TDV_HIP(ctx, hipMemcpyAsync(b.st, h, sizeof(IcpState), hipMemcpyHostToDevice, s));
    const float tau = tau_le(thr);
    if (grid.usable) {
        k_icp_nn_grid<<<(ns + 255) / 256, 256, 0, s>>>(d_src, ns, reinterpret_cast<const GridEntry*>(grid.table), reinterpret_cast<const float4*>(grid.node),
                                                       grid.mask, grid.shift, grid.inv_cell, b.st, tau, b.pd2, b.pchunk);
    } else if (pruned) {   // explicit request only: entries beyond the threshold come back as corr 0 / d2 FLT_MAX
        SortedCloud st{};
        TDV_TRY(spatial_sort_cloud(ctx, d_tgt, nt, st));
        k_icp_nn_pruned<<<(ns + PN_WAVES * PN_PTS - 1) / (PN_WAVES * PN_PTS), 64 * PN_WAVES, 0, s>>>(
            d_src, ns, st.sx, st.sy, st.sz, st.orig, nt, st.lbox, st.n_leaf, st.tbox, st.n_top, b.st, tau, b.pd2, b.pchunk);
    } else {
        k_aos_to_soa_pad<<<(p.nt_pad + 255) / 256, 256, 0, s>>>(d_tgt, nt, p.nt_pad, INFINITY, b.tx, b.ty, b.tz);
        k_icp_nn_scan<<<dim3(p.blocks_x, p.nsplit), NN_BLOCK, 0, s>>>(d_src, ns, p.ns_pad, b.tx, b.ty, b.tz, p.n_chunks,
                                                                      p.chunks_per_split, b.st, b.pd2, b.pchunk);
    }
    k_icp_accumulate<2, 1><<<p.acc_blocks, 256, 0, s>>>(d_src, ns, p.ns_pad, d_tgt, nullptr, b.tx, b.ty, b.tz, p.nsplit,
                                                        b.pd2, b.pchunk, pruned ? 1 : 0, b.st, tau, 0, b.slabs, b.ticket, outs.corr, outs.d2, outs.accepted);
    TDV_CHECK_LAUNCH(ctx);
    TDV_HIP(ctx, hipMemcpyAsync(h, b.st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    if (n_corr) *n_corr = h->n_corr;
    return TDV_OK;
}

}  // namespace tdv
