// Multi-GPU entry points of the C ABI: the instance-sharded job needs ONE broadcast (the prepared
// reference model: points | normals | FPFH, 156 B per point) and ONE gather (76-B results); nothing on an instance's data
// path is collective.  The reference has no multi-GPU code at all (it fans instances out over host threads,
// /root/reference/src/pipeline.cpp:321-327); this is the same fan-out across processes, one per GPU, over RCCL / xGMI.
//
// The communicator is the CALLER's (an ncclComm_t of the RCCL the host program uses: one rank per process, created with
// ncclCommInitRank from an id the host distributes however it likes).  RCCL is not linked: its three entry points are
// looked up at first use among the symbols already loaded into the process (the library that created the communicator),
// then by dlopen("librccl.so.1") — so a single-GPU deployment never loads RCCL, and a process that uses PyTorch's bundled
// RCCL and one that links /opt/rocm's both work.  Collectives are enqueued on the ctx's stream.
#include "tdv_internal.hpp"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

// the slice of <rccl/rccl.h> used here (ABI-stable since NCCL 2.x)
typedef void* comm_t;
typedef int (*fn_bcast)(const void*, void*, size_t, int /*ncclDataType_t*/, int, comm_t, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, comm_t, hipStream_t);
typedef int (*fn_count)(const comm_t, int*);
typedef int (*fn_rank)(const comm_t, int*);
typedef const char* (*fn_errstr)(int);
constexpr int kNcclChar = 0, kNcclFloat32 = 7;

struct Rccl {
    fn_bcast bcast = nullptr; fn_allgather allgather = nullptr; fn_count count = nullptr; fn_rank rank = nullptr; fn_errstr errstr = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

void resolve() {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclBroadcast")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
    }
    g_rccl.bcast = (fn_bcast)dlsym(h, "ncclBroadcast");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.count = (fn_count)dlsym(h, "ncclCommCount");
    g_rccl.rank = (fn_rank)dlsym(h, "ncclCommUserRank");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.bcast && g_rccl.allgather && g_rccl.count && g_rccl.rank;
}

int rccl_fail(tdv_ctx* ctx, int rc, const char* what) {
    std::snprintf(ctx->err, sizeof(ctx->err), "%s failed: %s", what, g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error");
    return TDV_ERR_LAUNCH;
}
#define TDV_RCCL(ctx, call, what) do { int rc__ = (call); if (rc__ != 0) return rccl_fail((ctx), rc__, (what)); } while (0)

int prepare(tdv_ctx* ctx, void* comm, int* world, int* rank) {
    if (!ctx || !comm) return TDV_ERR_BAD_ARG;
    std::call_once(g_once, resolve);
    if (!g_rccl.ok) { std::snprintf(ctx->err, sizeof(ctx->err), "RCCL (librccl.so.1) is not available in this process"); return TDV_ERR_NO_DEVICE; }
    TDV_HIP(ctx, hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    TDV_RCCL(ctx, g_rccl.count(comm, world), "ncclCommCount");
    TDV_RCCL(ctx, g_rccl.rank(comm, rank), "ncclCommUserRank");
    return TDV_OK;
}

// Every rank-local argument is checked BEFORE the first payload collective, and the verdict is made identical on all
// ranks by one all-gather of a small header (4 ints per rank): a rank that returned on its own between two collectives would
// leave the others blocked in RCCL for ever.  After the header every rank takes the same branch and returns the same status.
struct Header { int ok, a, b, c; };

int exchange_header(tdv_ctx* ctx, void* comm, int world, int rank, Header mine, Header** all_out) {
    hipStream_t s = ctx->stream;
    Header *d_send, *d_recv;
    TDV_TRY(tdv::ws_alloc(ctx, 1, &d_send));
    TDV_TRY(tdv::ws_alloc(ctx, (size_t)world, &d_recv));
    TDV_TRY(tdv::pin_reserve(ctx, sizeof(Header) * (size_t)(world + 1)));
    Header* h = reinterpret_cast<Header*>(ctx->pin);
    h[world] = mine;
    TDV_HIP(ctx, hipMemcpyAsync(d_send, h + world, sizeof(Header), hipMemcpyHostToDevice, s));
    TDV_RCCL(ctx, g_rccl.allgather(d_send, d_recv, sizeof(Header), kNcclChar, comm, s), "ncclAllGather(header)");
    TDV_HIP(ctx, hipMemcpyAsync(h, d_recv, sizeof(Header) * (size_t)world, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    *all_out = h;
    (void)rank;
    return TDV_OK;
}

}  // namespace

extern "C" {

int tdv_broadcast_model(tdv_ctx* ctx, void* rccl_comm, int root, float* d_xyz, float* d_normals, float* d_fpfh, int capacity, int* n_model) {
    int world = 0, rank = 0;
    TDV_TRY(prepare(ctx, rccl_comm, &world, &rank));
    // `root` must be the same on every rank (it is an argument of the collective itself, as in ncclBroadcast); everything else
    // is rank-local and goes through the header.
    if (root < 0 || root >= world) return TDV_ERR_BAD_ARG;
    TDV_TRY(tdv::ws_reset(ctx));
    hipStream_t s = ctx->stream;
    Header mine;
    mine.ok = (n_model && capacity >= 0 && (rank != root || (*n_model >= 0 && *n_model <= capacity))) ? 1 : 0;
    mine.a = (rank == root && n_model) ? *n_model : 0;          // the point count travels in the root's header
    mine.b = capacity;
    mine.c = (d_normals ? 1 : 0) | ((d_xyz && d_fpfh) ? 2 : 0); // bit 1: this rank passed the two mandatory buffers (needed only if the model is not empty)
    Header* all;
    TDV_TRY(exchange_header(ctx, rccl_comm, world, rank, mine, &all));
    const int n = all[root].a;
    int bad_rank = -1, min_cap = all[0].b, with_normals = 1;
    for (int r = 0; r < world; ++r) {
        if ((!all[r].ok || (n > 0 && !(all[r].c & 2))) && bad_rank < 0) bad_rank = r;     // an empty model needs no buffers anywhere
        if (all[r].b < min_cap) min_cap = all[r].b;
        with_normals &= all[r].c & 1;
    }
    if (bad_rank >= 0) { std::snprintf(ctx->err, sizeof(ctx->err), "tdv_broadcast_model: invalid arguments on rank %d (no payload was sent)", bad_rank); return TDV_ERR_BAD_ARG; }
    if (n_model) *n_model = n;
    if (n > min_cap) { std::snprintf(ctx->err, sizeof(ctx->err), "model of %d points does not fit the smallest capacity over ranks (%d)", n, min_cap); return TDV_ERR_BAD_ARG; }
    if (n == 0) return TDV_OK;
    // the pack, three in-place broadcasts.  Normals travel only when EVERY rank passed a buffer (a model without normals is
    // legal: point-to-point ICP); a rank that did pass one while another did not keeps its buffer untouched.
    TDV_RCCL(ctx, g_rccl.bcast(d_xyz, d_xyz, (size_t)n * 3, kNcclFloat32, root, rccl_comm, s), "ncclBroadcast(points)");
    if (with_normals) TDV_RCCL(ctx, g_rccl.bcast(d_normals, d_normals, (size_t)n * 3, kNcclFloat32, root, rccl_comm, s), "ncclBroadcast(normals)");
    TDV_RCCL(ctx, g_rccl.bcast(d_fpfh, d_fpfh, (size_t)n * 33, kNcclFloat32, root, rccl_comm, s), "ncclBroadcast(fpfh)");
    TDV_HIP(ctx, hipStreamSynchronize(s));
    return TDV_OK;
}

int tdv_gather_results(tdv_ctx* ctx, void* rccl_comm, const tdv_instance_result* local, int n_local, int slots_per_rank,
                       tdv_instance_result* all) {
    int world = 0, rank = 0;
    TDV_TRY(prepare(ctx, rccl_comm, &world, &rank));
    TDV_TRY(tdv::ws_reset(ctx));
    hipStream_t s = ctx->stream;
    Header mine;
    mine.ok = (n_local >= 0 && slots_per_rank >= n_local && all && (n_local == 0 || local)) ? 1 : 0;
    mine.a = slots_per_rank; mine.b = n_local; mine.c = 0;
    Header* hdr;
    TDV_TRY(exchange_header(ctx, rccl_comm, world, rank, mine, &hdr));
    for (int r = 0; r < world; ++r) {
        if (!hdr[r].ok) { std::snprintf(ctx->err, sizeof(ctx->err), "tdv_gather_results: invalid arguments on rank %d (nothing was gathered)", r); return TDV_ERR_BAD_ARG; }
        if (hdr[r].a != hdr[0].a) { std::snprintf(ctx->err, sizeof(ctx->err), "tdv_gather_results: slots_per_rank differs between ranks (%d on rank 0, %d on rank %d)", hdr[0].a, hdr[r].a, r); return TDV_ERR_BAD_ARG; }
    }
    if (slots_per_rank == 0) return TDV_OK;
    const size_t rec = sizeof(tdv_instance_result), mine_b = (size_t)slots_per_rank * rec, total = mine_b * (size_t)world;
    char *d_send, *d_recv;
    TDV_TRY(tdv::ws_alloc(ctx, mine_b, &d_send));
    TDV_TRY(tdv::ws_alloc(ctx, total, &d_recv));
    TDV_TRY(tdv::pin_reserve(ctx, total));
    std::memset(ctx->pin, 0, mine_b);
    for (int i = n_local; i < slots_per_rank; ++i) reinterpret_cast<tdv_instance_result*>(ctx->pin)[i].status = -1;   // unused slot
    if (n_local) std::memcpy(ctx->pin, local, (size_t)n_local * rec);
    TDV_HIP(ctx, hipMemcpyAsync(d_send, ctx->pin, mine_b, hipMemcpyHostToDevice, s));
    TDV_RCCL(ctx, g_rccl.allgather(d_send, d_recv, mine_b, kNcclChar, rccl_comm, s), "ncclAllGather(results)");
    TDV_HIP(ctx, hipMemcpyAsync(ctx->pin, d_recv, total, hipMemcpyDeviceToHost, s));
    TDV_HIP(ctx, hipStreamSynchronize(s));
    std::memcpy(all, ctx->pin, total);
    return TDV_OK;
}

}  // extern "C"
