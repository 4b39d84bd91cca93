// Stable radix sort of (64-bit key, 32-bit value) pairs on the low `end_bit` bits of the key: rocPRIM's device radix sort —
// a library primitive for a plain sort, as hipBLASLt would be for a plain GEMM.  It replaces the two full bitonic sorts of
// 16-byte records in the descriptor index build (csrc/fmatch.hip: ~30 launches and 0.19 ms each at 150k rows).
#include <cstring>
#include "tdv_internal.hpp"
#include <rocprim/rocprim.hpp>

namespace tdv {

int radix_sort_pairs_dev(tdv_ctx* ctx, const unsigned long long* d_keys_in, unsigned long long* d_keys_out,
                         const unsigned* d_vals_in, unsigned* d_vals_out, size_t n, int end_bit) {
    if (!ctx || end_bit < 1 || end_bit > 64 || (n > 0 && (!d_keys_in || !d_keys_out || !d_vals_in || !d_vals_out))) return TDV_ERR_BAD_ARG;
    if (n == 0) return TDV_OK;
    size_t bytes = 0;
    TDV_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, 0u, (unsigned)end_bit, ctx->stream));
    unsigned char* tmp = nullptr;
    TDV_TRY(ws_alloc(ctx, bytes + 256, &tmp));
    TDV_HIP(ctx, rocprim::radix_sort_pairs(tmp, bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, 0u, (unsigned)end_bit, ctx->stream));
    return TDV_OK;
}

}  // namespace tdv
