// Context, workspace arena, timers and the host helpers that are part of the path's semantics
// (RANSAC index stream, sqrt-free thresholds, pose composition).
#include "tdv_internal.hpp"
#include <cmath>
#include <cfloat>
#include <algorithm>

namespace tdv {

int ws_reset(tdv_ctx* ctx) {
    // coalesce: if the previous call needed more than one block, replace them by one block
    if (ctx->blocks.size() > 1) {
        size_t total = 0;
        TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& b : ctx->blocks) { total += b.cap; (void)hipFree(b.p); }
        ctx->blocks.clear();
        total = align_up(total + total / 4, (size_t)1 << 20);
        char* p = nullptr;
        TDV_HIP(ctx, hipMalloc((void**)&p, total));
        ctx->blocks.push_back({p, total});
    }
    ctx->cur_block = 0; ctx->cur_off = 0; ctx->used_total = 0;
    return TDV_OK;
}

int ws_alloc_bytes(tdv_ctx* ctx, size_t bytes, void** out) {
    bytes = align_up(bytes ? bytes : 1, 256);
    while (ctx->cur_block < ctx->blocks.size()) {
        auto& b = ctx->blocks[ctx->cur_block];
        if (ctx->cur_off + bytes <= b.cap) {
            *out = b.p + ctx->cur_off;
            ctx->cur_off += bytes; ctx->used_total += bytes;
            return TDV_OK;
        }
        ctx->cur_block++; ctx->cur_off = 0;
    }
    size_t cap = std::max(bytes, (size_t)64 << 20);
    if (!ctx->blocks.empty()) cap = std::max(cap, ctx->blocks.back().cap);
    char* p = nullptr;
    TDV_HIP(ctx, hipMalloc((void**)&p, cap));
    ctx->blocks.push_back({p, cap});
    ctx->cur_block = ctx->blocks.size() - 1;
    ctx->cur_off = bytes; ctx->used_total += bytes;
    *out = p;
    return TDV_OK;
}

int pin_reserve(tdv_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pin_cap) return TDV_OK;
    if (ctx->pin) { TDV_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->pin); ctx->pin = nullptr; ctx->pin_cap = 0; }
    size_t cap = align_up(bytes, (size_t)1 << 16);
    TDV_HIP(ctx, hipHostMalloc((void**)&ctx->pin, cap, hipHostMallocDefault));
    ctx->pin_cap = cap;
    return TDV_OK;
}

static hipEvent_t get_event(tdv_ctx* ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

hipEvent_t event_acquire(tdv_ctx* ctx) { return get_event(ctx); }
void event_release(tdv_ctx* ctx, hipEvent_t e) { if (e) ctx->event_pool.push_back(e); }

ScopedTimer::ScopedTimer(tdv_ctx* c, int s) : ctx(c), slot(s) {
    if (!ctx->timing || slot < 0) return;     // slot < 0: nothing to time here
    a = get_event(ctx); b = get_event(ctx);
    if (a) (void)hipEventRecord(a, ctx->stream);
}
ScopedTimer::~ScopedTimer() {
    if (!ctx->timing || slot < 0 || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->timers[slot].pending.emplace_back(a, b);
}

// ---- mt19937 + libstdc++-11 uniform_int_distribution<size_t> (Lemire's nearly-divisionless
// method on 32-bit draws when the range fits in 32 bits; for larger ranges libstdc++ composes
// two draws — clouds never reach 2^32 points, so that branch is rejected up front).
namespace {
struct Mt19937 {
    uint32_t mt[624]; int idx;
    explicit Mt19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
};
inline uint32_t lemire(Mt19937& g, uint32_t range) {  // uniform in [0, range)
    uint64_t product = (uint64_t)g.next() * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (uint32_t)(-range) % range;
        while (low < threshold) {
            product = (uint64_t)g.next() * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}
}  // namespace

void mt19937_lemire_triples(uint32_t seed, uint64_t n, int count, uint64_t* out) {
    TripleStream ts(seed, n);
    for (int i = 0; i < count; ++i) ts.next(out + 3 * (size_t)i);
}

// the raw 32-bit outputs of mt19937(seed), for the device-side index sampler of the batched RANSAC (ransac.hip)
void mt19937_raw(uint32_t seed, size_t count, uint32_t* out) {
    Mt19937 g(seed);
    for (size_t i = 0; i < count; ++i) out[i] = g.next();
}

struct TripleStream::Impl { Mt19937 g; explicit Impl(uint32_t seed) : g(seed) {} };
TripleStream::TripleStream(uint32_t seed, uint64_t n) : impl_(new Impl(seed)), n_(n) {}
TripleStream::~TripleStream() { delete impl_; }
void TripleStream::next(uint64_t* out3) {
    if (n_ == ((uint64_t)1 << 32)) {  // urange == urng range: one raw draw each
        for (int k = 0; k < 3; ++k) out3[k] = impl_->g.next();
        return;
    }
    const uint32_t range = (uint32_t)n_;
    for (int k = 0; k < 3; ++k) out3[k] = lemire(impl_->g, range);
}

float tau_le(float thr) {  // largest f with sqrtf(f) <= thr ; returns -1 if none (thr < 0 or NaN)
    if (!(thr >= 0.f)) return -1.f;
    if (std::isinf(thr)) return FLT_MAX;
    float c = thr * thr;
    if (std::isinf(c)) c = FLT_MAX;
    while (c > 0.f && std::sqrt(c) > thr) c = std::nextafterf(c, -1.f);
    while (c < FLT_MAX && std::sqrt(std::nextafterf(c, INFINITY)) <= thr) c = std::nextafterf(c, INFINITY);
    return c;
}
float tau_lt(float thr) {  // smallest f >= 0 with sqrtf(f) >= thr ; d2 < f <=> sqrtf(d2) < thr
    if (!(thr > 0.f)) return 0.f;  // nothing is < thr when thr <= 0 (d2 >= 0) ; NaN -> no inliers
    if (std::isinf(thr)) return INFINITY;
    float c = thr * thr;
    if (std::isinf(c)) c = FLT_MAX;
    while (c < FLT_MAX && std::sqrt(c) < thr) c = std::nextafterf(c, INFINITY);
    while (c > 0.f && std::sqrt(std::nextafterf(c, -1.f)) >= thr) c = std::nextafterf(c, -1.f);
    return c;
}

}  // namespace tdv

using namespace tdv;

extern "C" {

const char* tdv_version(void) { return "3dvision_amd 0.1 (gfx950)"; }

const char* tdv_status_string(int s) {
    switch (s) {
        case TDV_OK: return "ok";
        case TDV_ERR_NO_DEVICE: return "no HIP device";
        case TDV_ERR_BAD_ARG: return "bad argument";
        case TDV_ERR_OOM: return "out of memory";
        case TDV_ERR_LAUNCH: return "HIP launch/copy failure";
        default: return "internal error";
    }
}

int tdv_device_count(int* count) {
    if (!count) return TDV_ERR_BAD_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return TDV_OK;
}

int tdv_ctx_create(int device, tdv_ctx** out) {
    if (!out) return TDV_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return TDV_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) return TDV_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return TDV_ERR_NO_DEVICE;
    tdv_ctx* c = new tdv_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return TDV_ERR_NO_DEVICE; }
    c->own_stream = true;
    // one persistent device word: the "last workgroup" ticket of the scans (reset by the kernel that uses it)
    // (cleared on the ctx's own stream and waited for: a null-stream memset is not ordered against a non-blocking stream)
    if (hipMalloc((void**)&c->scan_ticket, 64) != hipSuccess || hipMemsetAsync(c->scan_ticket, 0, 64, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; return TDV_ERR_OOM;
    }
    if (const char* e = getenv("TDV_ICP_SEARCH")) {
        if (!strcmp(e, "brute")) c->icp_search = TDV_ICP_SEARCH_BRUTE;
        else if (!strcmp(e, "pruned")) c->icp_search = TDV_ICP_SEARCH_PRUNED;
        else if (!strcmp(e, "grid")) c->icp_search = TDV_ICP_SEARCH_GRID;
    }
    if (const char* e = getenv("TDV_ICP_ACCUMULATE")) {
        if (!strcmp(e, "reference")) c->icp_accumulate = TDV_ICP_ACCUMULATE_REFERENCE;
    }
    *out = c;
    return TDV_OK;
}

int tdv_ctx_set_icp_accumulation(tdv_ctx* ctx, int mode) {
    if (!ctx || (mode != TDV_ICP_ACCUMULATE_TREE && mode != TDV_ICP_ACCUMULATE_REFERENCE)) return TDV_ERR_BAD_ARG;
    ctx->icp_accumulate = mode;
    return TDV_OK;
}

int tdv_ctx_set_icp_search(tdv_ctx* ctx, int mode) {
    if (!ctx || mode < TDV_ICP_SEARCH_AUTO || mode > TDV_ICP_SEARCH_GRID) return TDV_ERR_BAD_ARG;
    ctx->icp_search = mode;
    return TDV_OK;
}

int tdv_ctx_set_ransac_score(tdv_ctx* ctx, int mode) {
    if (!ctx || (mode != TDV_RANSAC_SCORE_FAST && mode != TDV_RANSAC_SCORE_EXACT && !(tdv::kStudyBuild && mode == TDV_RANSAC_SCORE_MATRIX))) return TDV_ERR_BAD_ARG;   // MATRIX: study build only
    ctx->ransac_score_mode = mode;
    return TDV_OK;
}

double tdv_ctx_last_ransac_rescore(tdv_ctx* ctx) { return ctx ? ctx->last_ransac_rescore : -1.0; }

double tdv_ctx_last_ransac_scored(tdv_ctx* ctx) { return ctx ? ctx->last_ransac_scored : 1.0; }

int tdv_ctx_last_icp_search(tdv_ctx* ctx) { return ctx ? ctx->last_icp_search : 0; }
int tdv_ctx_last_feature_match_path(tdv_ctx* ctx) { return ctx ? ctx->last_fm_path : 0; }
int tdv_ctx_last_voxel_grouping(tdv_ctx* ctx) { return ctx ? ctx->last_voxel_grouping : 0; }
int tdv_ctx_last_batch_lanes(tdv_ctx* ctx) { return ctx ? ctx->last_batch_lanes : 0; }
unsigned long long tdv_ctx_workspace_bytes(tdv_ctx* ctx) {
    unsigned long long total = 0;
    for (tdv_ctx* c = ctx; c; c = c->helper)
        for (auto& b : c->blocks) total += b.cap;
    return total;
}

int tdv_ctx_set_stream(tdv_ctx* ctx, void* s) {
    if (!ctx) return TDV_ERR_BAD_ARG;
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    ctx->stream = (hipStream_t)s;
    ctx->own_stream = false;
    return TDV_OK;
}
void* tdv_ctx_get_stream(tdv_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int tdv_ctx_synchronize(tdv_ctx* ctx) {
    if (!ctx) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TDV_OK;
}

void tdv_ctx_destroy(tdv_ctx* ctx) {
    if (!ctx) return;
    if (ctx->helper) { tdv_ctx_destroy(ctx->helper); ctx->helper = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->blocks) (void)hipFree(b.p);
    if (ctx->scan_ticket) (void)hipFree(ctx->scan_ticket);
    if (ctx->chain_status) (void)hipFree(ctx->chain_status);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (auto& t : ctx->timers) for (auto& p : t.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* tdv_last_error(tdv_ctx* ctx) { return ctx ? ctx->err : ""; }

int tdv_timing_enable(tdv_ctx* ctx, int on) {
    if (!ctx) return TDV_ERR_BAD_ARG;
    ctx->timing = on != 0;
    return TDV_OK;
}

int tdv_timing_read(tdv_ctx* ctx, int slot, double* total_ms, int* launches) {
    if (!ctx || slot < 0 || slot >= TDV_TIMER_COUNT) return TDV_ERR_BAD_ARG;
    TDV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto& t = ctx->timers[slot];
    for (auto& p : t.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { t.total_ms += ms; t.launches++; }
        ctx->event_pool.push_back(p.first); ctx->event_pool.push_back(p.second);
    }
    t.pending.clear();
    if (ctx->helper) {   // the batched pipeline's second lane times its kernels on its own stream
        double hms = 0.0; int hl = 0;
        if (tdv_timing_read(ctx->helper, slot, &hms, &hl) == TDV_OK) { t.total_ms += hms; t.launches += hl; }
    }
    if (total_ms) *total_ms = t.total_ms;
    if (launches) *launches = t.launches;
    t.total_ms = 0.0; t.launches = 0;
    return TDV_OK;
}

int tdv_sample_triples(uint32_t seed, uint64_t n, int count, uint64_t* out) {
    if (!out || count < 0 || n == 0 || n > ((uint64_t)1 << 32)) return TDV_ERR_BAD_ARG;
    mt19937_lemire_triples(seed, n, count, out);
    return TDV_OK;
}

// out = E * inverse(T), all column-major 4x4 (src/pipeline.cpp:136-137).  General inverse by
// the adjugate with double intermediates, rounded once to float.
int tdv_pose_compose(const float* E, const float* T, float* out) {
    if (!E || !T || !out) return TDV_ERR_BAD_ARG;
    double a[16], inv[16], cof[16];
    for (int i = 0; i < 16; ++i) a[i] = T[i];
    auto A = [&](int r, int c) { return a[c * 4 + r]; };
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            int rr[3], cc[3], k = 0;
            for (int i = 0; i < 4; ++i) if (i != r) rr[k++] = i;
            k = 0;
            for (int i = 0; i < 4; ++i) if (i != c) cc[k++] = i;
            double m = A(rr[0], cc[0]) * (A(rr[1], cc[1]) * A(rr[2], cc[2]) - A(rr[1], cc[2]) * A(rr[2], cc[1]))
                     - A(rr[0], cc[1]) * (A(rr[1], cc[0]) * A(rr[2], cc[2]) - A(rr[1], cc[2]) * A(rr[2], cc[0]))
                     + A(rr[0], cc[2]) * (A(rr[1], cc[0]) * A(rr[2], cc[1]) - A(rr[1], cc[1]) * A(rr[2], cc[0]));
            cof[c * 4 + r] = ((r + c) & 1) ? -m : m;
        }
    double det = 0;
    for (int c = 0; c < 4; ++c) det += A(0, c) * cof[c * 4 + 0];
    if (det == 0.0) return TDV_ERR_BAD_ARG;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv[c * 4 + r] = cof[r * 4 + c] / det;
    float invf[16], res[16];
    for (int i = 0; i < 16; ++i) invf[i] = (float)inv[i];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float acc = E[0 * 4 + i] * invf[j * 4 + 0];
            for (int k = 1; k < 4; ++k) acc = E[k * 4 + i] * invf[j * 4 + k] + acc;
            res[j * 4 + i] = acc;
        }
    for (int i = 0; i < 16; ++i) out[i] = res[i];
    return TDV_OK;
}

}  // extern "C"
