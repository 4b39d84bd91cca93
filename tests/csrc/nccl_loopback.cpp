// Test infrastructure, not product code: an in-process stand-in for the RCCL entry points csrc/comm.hip resolves
// (ncclBroadcast, ncclAllGather, ncclCommCount, ncclCommUserRank, ncclGetErrorString) so that tdv_broadcast_model and
// tdv_gather_results can run with MORE THAN ONE RANK on a box with one GPU: every "rank" is a host thread of this process with
// its own tdv_ctx (own stream, own workspace) on GPU 0; a collective is a rendezvous of the ranks' host threads (mutex +
// condition variable) around device-to-device copies on the caller's stream.
//
// What it is for: the LOGIC of the C ABI's collectives - the header exchange that makes every rank take the same branch and
// return the same status, slot layout, root handling, the "normals only if every rank has them" rule.  What it is NOT: RCCL.
// Nothing about xGMI, RCCL's protocols or its stream semantics is exercised; that needs the driver's multi-GPU node.
//
// A rank that waits longer than the group's timeout for the others (i.e. the product left a rank alone inside a collective -
// the deadlock this test exists to catch) gets ncclInternalError instead of hanging the test.
// Loaded with RTLD_GLOBAL before the first tdv_* collective, in a process that has not loaded the real RCCL
// (tests/loopback_comm_worker.py).
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <vector>

namespace {
struct Group {
    int world = 0, timeout_ms = 10000;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; long generation = 0;
    std::vector<const void*> send; std::vector<size_t> bytes; std::vector<int> op;
    int calls = 0, timeouts = 0, mismatches = 0;
};
struct Comm { Group* g; int rank; };

bool barrier(Group* g) {
    std::unique_lock<std::mutex> lk(g->m);
    const long gen = g->generation;
    if (++g->arrived == g->world) { g->arrived = 0; ++g->generation; g->cv.notify_all(); return true; }
    if (!g->cv.wait_for(lk, std::chrono::milliseconds(g->timeout_ms), [&] { return g->generation != gen; })) { --g->arrived; ++g->timeouts; return false; }
    return true;
}
size_t dtype_size(int dt) { return (dt == 0 || dt == 1) ? 1 : (dt == 6 || dt == 9) ? 2 : (dt == 4 || dt == 5 || dt == 8) ? 8 : 4; }   // ncclChar/Uint8, Float16/Bfloat16, (U)Int64/Float64, the rest 4

// post this rank's operands, meet the others, check that everybody is in the same collective with the same size
int enter(Comm* c, int op, const void* send, size_t bytes, hipStream_t s) {
    Group* g = c->g;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;       // what this rank enqueued before the collective is complete: its send buffer is readable
    { std::lock_guard<std::mutex> lk(g->m); g->send[c->rank] = send; g->bytes[c->rank] = bytes; g->op[c->rank] = op; ++g->calls; }
    if (!barrier(g)) return 3;
    for (int r = 0; r < g->world; ++r)
        if (g->op[r] != op || g->bytes[r] != bytes) { std::lock_guard<std::mutex> lk(g->m); ++g->mismatches; return 4; }
    return 0;
}
int leave(Comm* c, hipStream_t s) {
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    return barrier(c->g) ? 0 : 3;                              // nobody reuses a send buffer before every rank has read it
}
}  // namespace

extern "C" {

int ncclCommCount(const void* comm, int* count) { *count = static_cast<const Comm*>(comm)->g->world; return 0; }
int ncclCommUserRank(const void* comm, int* rank) { *rank = static_cast<const Comm*>(comm)->rank; return 0; }
const char* ncclGetErrorString(int rc) {
    switch (rc) { case 0: return "no error"; case 1: return "loop-back: HIP error"; case 3: return "loop-back: a rank waited for the others past the timeout (a rank left the collective sequence)";
                  case 4: return "loop-back: the ranks are in different collectives or pass different sizes"; default: return "loop-back: error"; }
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    const size_t bytes = sendcount * dtype_size(datatype);
    if (int rc = enter(c, 1, sendbuff, bytes, stream)) return rc;
    for (int r = 0; r < c->g->world; ++r)
        if (hipMemcpyAsync(static_cast<char*>(recvbuff) + (size_t)r * bytes, c->g->send[r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    return leave(c, stream);
}

int ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    const size_t bytes = count * dtype_size(datatype);
    if (root < 0 || root >= c->g->world) return 4;
    if (int rc = enter(c, 2 + root * 16, sendbuff, bytes, stream)) return rc;
    const void* src = c->g->send[root];
    if (recvbuff != src && hipMemcpyAsync(recvbuff, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    return leave(c, stream);
}

// ---- test-side handles
void* loopback_group_create(int world, int timeout_ms) {
    Group* g = new Group();
    g->world = world; g->timeout_ms = timeout_ms;
    g->send.assign(world, nullptr); g->bytes.assign(world, 0); g->op.assign(world, 0);
    return g;
}
void* loopback_comm_create(void* group, int rank) { return new Comm{static_cast<Group*>(group), rank}; }
void loopback_comm_destroy(void* comm) { delete static_cast<Comm*>(comm); }
void loopback_group_stats(void* group, int* calls, int* timeouts, int* mismatches) {
    Group* g = static_cast<Group*>(group);
    std::lock_guard<std::mutex> lk(g->m);
    *calls = g->calls; *timeouts = g->timeouts; *mismatches = g->mismatches;
}
void loopback_group_destroy(void* group) { delete static_cast<Group*>(group); }

}  // extern "C"
