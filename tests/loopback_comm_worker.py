#!/usr/bin/env python3
"""Worker of tests/test_gpu_comm_world2.py: tdv_broadcast_model / tdv_gather_results with TWO ranks on one GPU.

Runs in its own process because csrc/comm.hip resolves the RCCL entry points once per process: here they resolve to the
loop-back stand-in tests/csrc/nccl_loopback.cpp (loaded RTLD_GLOBAL before the first call; neither torch nor the real RCCL is
ever loaded).  Each rank is a host thread with its own tdv_ctx on GPU 0.  Device memory comes from the HIP runtime through
ctypes.  Prints one JSON object: {case: {...}}; the pytest side asserts on it.

What this proves: the LOGIC of the C ABI's collectives at world size 2 - same status on every rank whatever one rank passes, no
rank left inside a collective (the stand-in turns a lone waiting rank into an error after its timeout, and every case joins its
threads under a timeout), payload and slot layout.  It is not a substitute for RCCL over xGMI."""
import ctypes as C
import importlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORLD = 2
JOIN_S = 30.0


def main():
    shim = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)          # before the product library makes its first collective call
    assert "torch" not in sys.modules
    shim.loopback_group_create.restype = C.c_void_p; shim.loopback_comm_create.restype = C.c_void_p
    shim.loopback_comm_create.argtypes = [C.c_void_p, C.c_int]; shim.loopback_group_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
    shim.loopback_comm_destroy.argtypes = [C.c_void_p]; shim.loopback_group_destroy.argtypes = [C.c_void_p]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    tdv = importlib.import_module("3dvision_amd")
    lib = tdv.lib()
    lib.tdv_broadcast_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.tdv_gather_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]

    def dev_array(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(a.nbytes, 4)) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        return p

    def to_host(p, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0
        return out

    group = shim.loopback_group_create(WORLD, 8000)
    comms = [shim.loopback_comm_create(group, r) for r in range(WORLD)]
    ctxs = [tdv.Context(0) for _ in range(WORLD)]

    def both(fn):
        """fn(rank) on two threads at once; returns the two results, or 'HUNG' where a thread did not come back."""
        res = [None] * WORLD

        def run(r):
            try:
                res[r] = fn(r)
            except Exception as e:  # noqa
                res[r] = "EXC %r" % (e,)
        th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(WORLD)]
        for t in th: t.start()
        for t in th: t.join(JOIN_S)
        return ["HUNG" if t.is_alive() else res[r] for r, t in enumerate(th)]

    rng = np.random.default_rng(5)
    n, cap = 3000, 4096
    xyz = rng.normal(size=(n, 3)).astype(np.float32); nrm = rng.normal(size=(n, 3)).astype(np.float32); fp = rng.random((n, 33)).astype(np.float32)
    SENT = np.float32(-7.5)                                   # what a receiving buffer holds before the call

    def buffers(rank, root, capacity=cap, normals=True, mandatory=True):
        if rank == root:
            pad = lambda a, w: np.concatenate([a, np.full((capacity - len(a), w), SENT, np.float32)]) if capacity > len(a) else a[:capacity]
            return dev_array(pad(xyz, 3)) if mandatory else None, dev_array(pad(nrm, 3)) if normals else None, dev_array(pad(fp, 33)) if mandatory else None
        return (dev_array(np.full((capacity, 3), SENT, np.float32)) if mandatory else None,
                dev_array(np.full((capacity, 3), SENT, np.float32)) if normals else None,
                dev_array(np.full((capacity, 33), SENT, np.float32)) if mandatory else None)

    def bcast_case(root=0, caps=(cap, cap), normals=(True, True), mandatory=(True, True), n_root=n):
        bufs = [buffers(r, root, caps[r], normals[r], mandatory[r]) for r in range(WORLD)]

        def call(r):
            cnt = C.c_int(n_root if r == root else -123)
            st = lib.tdv_broadcast_model(ctxs[r]._h, comms[r], root, bufs[r][0], bufs[r][1], bufs[r][2], caps[r], C.byref(cnt))
            return dict(status=st, n=cnt.value, err=lib.tdv_last_error(ctxs[r]._h).decode())
        out = both(call)
        other = 1 - root
        rec = dict(ranks=out)
        if all(isinstance(o, dict) for o in out):
            m = min(n_root, caps[other])
            if bufs[other][0] is not None:
                x = to_host(bufs[other][0], (caps[other], 3)); f = to_host(bufs[other][2], (caps[other], 33))
                rec["xyz_delivered"] = bool(m > 0 and x[:m].tobytes() == xyz[:m].tobytes()); rec["fpfh_delivered"] = bool(m > 0 and f[:m].tobytes() == fp[:m].tobytes())
                rec["xyz_untouched"] = bool((x == SENT).all()); rec["tail_untouched"] = bool((x[m:] == SENT).all() and (f[m:] == SENT).all())
            if bufs[other][1] is not None:
                g = to_host(bufs[other][1], (caps[other], 3))
                rec["normals_delivered"] = bool(m > 0 and g[:m].tobytes() == nrm[:m].tobytes()); rec["normals_untouched"] = bool((g == SENT).all())
        for b in bufs:
            for p in b:
                if p is not None: hip.hipFree(p)
        return rec

    def results(rank, k):
        arr = (tdv.InstanceResultC * max(k, 1))()
        for i in range(k):
            for j in range(16): arr[i].T[j] = float(1000 * rank + 16 * i + j)
            arr[i].fitness = 0.25 + rank + 0.01 * i; arr[i].rmse = 1e-3 * (i + 1); arr[i].coarse_inliers = 100 * rank + i
            arr[i].icp_iterations = i; arr[i].n_points = 5000 + i; arr[i].n_voxels = 400 + rank; arr[i].status = 0
        return arr

    def gather_case(n_local=(3, 2), slots=(4, 4)):
        def call(r):
            loc = results(r, n_local[r]); allr = (tdv.InstanceResultC * (WORLD * max(slots) + 1))()
            st = lib.tdv_gather_results(ctxs[r]._h, comms[r], loc, n_local[r], slots[r], allr)
            rows = []
            if st == 0:
                for q in range(WORLD * slots[r]):
                    a = allr[q]; rows.append([a.status, a.T[0], a.T[15], round(a.fitness, 4), a.coarse_inliers, a.n_voxels])
            return dict(status=st, rows=rows, err=lib.tdv_last_error(ctxs[r]._h).decode())
        return dict(ranks=both(call))

    out = {}
    out["happy"] = bcast_case()
    out["root1"] = bcast_case(root=1)
    out["capacity_small_on_one_rank"] = bcast_case(caps=(cap, n - 1))
    out["capacity_small_on_root"] = bcast_case(caps=(n - 1, cap))
    out["normals_null_on_one_rank"] = bcast_case(normals=(True, False))
    out["normals_null_on_root"] = bcast_case(normals=(False, True))
    out["empty_model"] = bcast_case(n_root=0)
    out["empty_model_no_buffers_on_receiver"] = bcast_case(n_root=0, mandatory=(True, False))
    out["buffers_null_on_receiver"] = bcast_case(mandatory=(True, False))
    out["gather"] = gather_case()
    out["gather_full_and_empty"] = gather_case(n_local=(4, 0))
    out["gather_slots_disagree"] = gather_case(slots=(4, 5))
    out["gather_too_many_on_one_rank"] = gather_case(n_local=(3, 5))
    out["gather_zero_slots"] = gather_case(n_local=(0, 0), slots=(0, 0))
    # after the error cases the pair still works (nobody is stuck, the collective sequence is aligned)
    out["happy_again"] = bcast_case()
    calls, touts, mism = C.c_int(), C.c_int(), C.c_int()
    shim.loopback_group_stats(group, C.byref(calls), C.byref(touts), C.byref(mism))
    out["shim"] = dict(calls=calls.value, timeouts=touts.value, mismatches=mism.value, rccl_loaded=any("rccl" in l for l in open("/proc/self/maps").read().splitlines()))
    print(json.dumps(out))
    sys.stdout.flush()
    os._exit(0)          # daemon threads (if any is stuck) must not keep the process


if __name__ == "__main__":
    main()
