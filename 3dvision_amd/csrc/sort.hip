// Stable LSD radix sort of (64-bit key, 32-bit value) pairs on the low `end_bit` bits of the key, hand-written for gfx950 (round 4;
// rounds 2-3 called rocPRIM's device radix sort here).  Used by the descriptor index build (csrc/fmatch.hip: two sorts of ~150k
// (key, row) pairs once per model) - a launch-bound size, so the design goal is FEW launches of short kernels, not bandwidth:
//   per pass of up to 8 bits, two launches
//     k_rs_count    every workgroup (2,048 pairs) counts its digits into a digit-major table [256][workgroups]
//     k_rs_scatter  every workgroup ranks its pairs stably and writes them: a pair's place = the exclusive scan of the table (digit-major
//                   order IS the output order) at [digit][workgroup] + the pairs of the same digit before it in the workgroup.  Up to
//                   RS_SELF_SCAN_MAX workgroups each scatter workgroup works its 256 scan values out of the raw table itself (a quarter
//                   of a row per thread, all loads in flight: ~76 KB from L2 at 150k pairs); above that the LAST workgroup of k_rs_count
//                   to finish (a ticket) scans the table in place - the release / acquire fences of that hand-over write back and
//                   invalidate an XCD's L2 and cost 10-20 us per launch, which only a large sort amortises.  Pairs are held striped (pair r * 1024 + thread), so "before it" is
//                   (round, wave, lane) order: within a wave a match-any of the digit (8 ballots) gives the rank among the lanes and the
//                   group's count, the 32 (round, wave) groups are then chained per digit by 256 threads through LDS.
//   the digit width is end_bit / passes rounded up, so 33 bits are five passes of 7 bits and 42 bits six of 7.
// No look-back, no spinning.
// Measured against rocPRIM on the index build's two sorts: profiles/r4/history/radix_sort.md.
#include <cstring>
#include "tdv_internal.hpp"

namespace tdv {

constexpr int RS_THREADS = 1024, RS_WAVES = RS_THREADS / 64, RS_ROUNDS = 2, RS_TILE = RS_THREADS * RS_ROUNDS, RS_GROUPS = RS_ROUNDS * RS_WAVES;
constexpr int RS_BINS = 256;
constexpr int RS_SELF_SCAN_MAX = 1024;      // workgroups (2 M pairs) up to which k_rs_scatter scans the raw table itself

// the lanes of the wave that hold a pair (valid) with the same digit as this lane
__device__ __forceinline__ unsigned long long rs_peers(unsigned d, bool valid, int bits) {
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
    }
    return peers;
}

template <bool SCAN>
__global__ __launch_bounds__(RS_THREADS)
void k_rs_count(const unsigned long long* __restrict__ keys, size_t n, int shift, unsigned mask, int bits, int nb, int* __restrict__ table, unsigned* __restrict__ ticket) {
    __shared__ int hist[RS_BINS];
    __shared__ bool is_last;
    __shared__ int wsum[RS_WAVES];
    if (threadIdx.x < RS_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        const bool valid = i < n;
        const unsigned d = valid ? (unsigned)(keys[i] >> shift) & mask : 0u;
        const unsigned long long peers = rs_peers(d, valid, bits);      // one LDS atomic per digit and wave: the high digits of a key are a few values
        if (valid && (peers & lt) == 0ull) atomicAdd(&hist[d], (int)__popcll(peers));
    }
    __syncthreads();
    if (threadIdx.x < RS_BINS) table[(size_t)threadIdx.x * nb + blockIdx.x] = hist[threadIdx.x];
    if (!SCAN) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the counts are visible device-wide before the ticket moves
        is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // acquire: the other workgroups' counts
    // exclusive scan of the whole table by this workgroup: a thread owns `per` consecutive entries - it adds them up (independent loads, all
    // in flight), the 1,024 sums are scanned through LDS, and a second walk writes the running prefix.  (First version: 1,024 entries per
    // round with a load, three barriers and a store in each - a memory latency per round, 16.7 us per launch at 37 workgroups.)
    // (relaxed device-scope atomic loads, not `volatile`: the compiler waits for every volatile access before it issues the next one -
    //  19 + 19 memory latencies per thread, 30 us per launch - while these stay in flight together)
    const int total = RS_BINS * nb;
    const int per = (total + RS_THREADS - 1) / RS_THREADS;
    const int i0 = min(threadIdx.x * per, total), i1 = min(i0 + per, total);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int sum = 0;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) sum += __hip_atomic_load(&table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll 8
    for (int i = i0; i < i1; ++i) {
        const int v = __hip_atomic_load(&table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        table[i] = run; run += v;
    }
    if (threadIdx.x == 0) *ticket = 0u;                           // ready for the next launch
}

template <bool SCANNED>
__global__ __launch_bounds__(RS_THREADS)
void k_rs_scatter(const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ vals_in, size_t n, int shift, unsigned mask, int bits, int nb,
                  const int* __restrict__ table, unsigned long long* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    __shared__ unsigned char cnt[RS_GROUPS][RS_BINS];          // pairs of a digit in a (round, wave) group: at most 64
    __shared__ unsigned short offs[RS_GROUPS][RS_BINS];        // ... in the groups before it
    __shared__ int gbase[RS_BINS];
    {
        unsigned* z = reinterpret_cast<unsigned*>(&cnt[0][0]);
#pragma unroll
        for (int k = 0; k < RS_GROUPS * RS_BINS / 4 / RS_THREADS; ++k) z[k * RS_THREADS + threadIdx.x] = 0u;
        static_assert(RS_GROUPS * RS_BINS / 4 % RS_THREADS == 0, "the count table is cleared in whole rounds");
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (SCANNED) {
        if (threadIdx.x < RS_BINS) gbase[threadIdx.x] = table[(size_t)threadIdx.x * nb + blockIdx.x];
    } else {
        // this workgroup's 256 values of the table's exclusive scan, out of the raw counts: thread (part, digit) adds up a quarter of the
        // digit's row - all of it, and what lies before this workgroup's column
        __shared__ int part_all[RS_THREADS / RS_BINS][RS_BINS], part_pre[RS_THREADS / RS_BINS][RS_BINS];
        __shared__ int wtot[RS_BINS / 64];
        constexpr int PARTS = RS_THREADS / RS_BINS;
        const int d = threadIdx.x & (RS_BINS - 1), part = threadIdx.x / RS_BINS;
        const int per = (nb + PARTS - 1) / PARTS, b0 = min(part * per, nb), b1 = min(b0 + per, nb);
        const int* __restrict__ row = table + (size_t)d * nb;
        int all = 0, pre = 0;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) { const int v = row[b]; all += v; pre += b < (int)blockIdx.x ? v : 0; }
        part_all[part][d] = all; part_pre[part][d] = pre;
        __syncthreads();
        if (threadIdx.x < RS_BINS) {
            int rowsum = 0, rowpre = 0;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) { rowsum += part_all[q][d]; rowpre += part_pre[q][d]; }
            int incl = rowsum;                                    // exclusive scan of the 256 row sums: the pairs of all smaller digits
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
            if (lane == 63) wtot[wave] = incl;
            part_all[0][d] = incl - rowsum + rowpre;              // (own slot: read above by this thread only)
        }
        __syncthreads();
        if (threadIdx.x < RS_BINS) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += wtot[w];
            gbase[d] = part_all[0][d] + before;
        }
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    unsigned long long key[RS_ROUNDS]; unsigned val[RS_ROUNDS]; int digit[RS_ROUNDS], rank[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0ull; val[r] = valid ? vals_in[i] : 0u;
        const unsigned d = (unsigned)(key[r] >> shift) & mask;
        digit[r] = valid ? (int)d : -1;
        const unsigned long long peers = rs_peers(d, valid, bits);
        rank[r] = __popcll(peers & lt);
        if (valid && rank[r] == 0) cnt[r * RS_WAVES + wave][d] = (unsigned char)__popcll(peers);
    }
    __syncthreads();
    if (threadIdx.x < RS_BINS) {
        int run = 0;
#pragma unroll 8
        for (int g = 0; g < RS_GROUPS; ++g) { offs[g][threadIdx.x] = (unsigned short)run; run += cnt[g][threadIdx.x]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if (digit[r] < 0) continue;
        const size_t dst = (size_t)gbase[digit[r]] + offs[r * RS_WAVES + wave][digit[r]] + rank[r];
        keys_out[dst] = key[r]; vals_out[dst] = val[r];
    }
}

int radix_sort_pairs_dev(tdv_ctx* ctx, const unsigned long long* d_keys_in, unsigned long long* d_keys_out,
                         const unsigned* d_vals_in, unsigned* d_vals_out, size_t n, int end_bit) {
    if (!ctx || end_bit < 1 || end_bit > 64 || (n > 0 && (!d_keys_in || !d_keys_out || !d_vals_in || !d_vals_out))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    if (n > (size_t)1 << 30) return TDV_ERR_BAD_ARG;              // (places are 32-bit)
    const int passes = (end_bit + 7) / 8, bits = (end_bit + passes - 1) / passes;
    const int nb = (int)((n + RS_TILE - 1) / RS_TILE);
    int* table = nullptr; unsigned long long* tkeys = nullptr; unsigned* tvals = nullptr;
    TDV_TRY(ws_alloc(ctx, (size_t)RS_BINS * nb, &table));
    if (passes > 1) { TDV_TRY(ws_alloc(ctx, n, &tkeys)); TDV_TRY(ws_alloc(ctx, n, &tvals)); }
    hipStream_t s = ctx->stream;
    unsigned* ticket = ctx->scan_ticket + 9;
    const unsigned long long* kin = d_keys_in; const unsigned* vin = d_vals_in;
    for (int p = 0; p < passes; ++p) {
        // the last pass writes the caller's buffers; the passes before it alternate between them and the temporaries
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        unsigned long long* kout = to_out ? d_keys_out : tkeys; unsigned* vout = to_out ? d_vals_out : tvals;
        const int shift = p * bits, width = std::min(bits, end_bit - shift);
        const unsigned mask = (1u << width) - 1u;
        if (nb <= RS_SELF_SCAN_MAX) {
            k_rs_count<false><<<nb, RS_THREADS, 0, s>>>(kin, n, shift, mask, width, nb, table, ticket);
            k_rs_scatter<false><<<nb, RS_THREADS, 0, s>>>(kin, vin, n, shift, mask, width, nb, table, kout, vout);
        } else {
            k_rs_count<true><<<nb, RS_THREADS, 0, s>>>(kin, n, shift, mask, width, nb, table, ticket);
            k_rs_scatter<true><<<nb, RS_THREADS, 0, s>>>(kin, vin, n, shift, mask, width, nb, table, kout, vout);
        }
        kin = kout; vin = vout;
    }
    TDV_CHECK_LAUNCH(ctx);
    return TDV_OK;
}

}  // namespace tdv
