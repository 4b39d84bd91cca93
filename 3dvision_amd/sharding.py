"""Multi-GPU instance sharding.

The reference fans independent instances out over host threads (src/pipeline.cpp:321-327); nothing is
exchanged between instances until the O(M^2) duplicate filter on 4x4 poses (:345).  Here the same axis
is sharded over ranks — one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests):

  * broadcast_model : ONE broadcast of the reference-model pack (points | normals | FPFH, 39 floats per
                      point) from rank 0 — the only payload every rank needs that is produced once;
  * shard_range     : contiguous block of instance ids per rank (mask locality), sizes differ by <= 1;
  * gather_results  : ONE gather of 19 floats per instance (16 T + fitness + rmse + inliers) to rank 0.

No collective sits on the data path of an instance; a single cloud pair is never split across GPUs.

The same two steps exist below Python, in the C ABI (tdv_broadcast_model / tdv_gather_results, csrc/comm.hip), for hosts
without torch: they take the caller's ncclComm_t.  `rccl_comm_from_process_group` makes one for a job that already has a
torch.distributed group (the id travels over that group), so that a Python job can move the model and the results through the
C ABI as well - what bench.py's N > 1 line does for config C5.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

MODEL_PACK_WIDTH = 3 + 3 + 33
RESULT_WIDTH = 19


def shard_range(n_items, world, rank):
    """[start, stop) of the contiguous shard of `rank`; the first n_items % world ranks get one extra."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_model(points, normals, fpfh):
    return torch.from_numpy(np.concatenate([points, normals, fpfh], 1).astype(np.float32))


def unpack_model(pack):
    a = pack.detach().cpu().numpy()
    return a[:, 0:3].copy(), a[:, 3:6].copy(), a[:, 6:39].copy()


def broadcast_model(pack, n_points, device, src=0):
    """Rank `src` passes its [n,39] pack; every other rank passes None and receives it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pack.to(device)
    if dist.get_rank() == src:
        buf = pack.to(device).contiguous()
    else:
        buf = torch.empty((n_points, MODEL_PACK_WIDTH), dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    return buf


def encode_result(T, fitness, rmse, inliers):
    r = np.empty(RESULT_WIDTH, np.float32)
    r[:16] = np.asarray(T, np.float32).reshape(16)
    r[16] = fitness; r[17] = rmse; r[18] = inliers
    return r


def gather_results(local, n_items, device, dst=0):
    """local: float32 [m, 19] results of this rank's shard, in shard order.  Returns the [n_items, 19]
    array in instance order on rank `dst`, None elsewhere."""
    local = torch.as_tensor(np.asarray(local, np.float32).reshape(-1, RESULT_WIDTH))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.numpy()
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = (n_items + world - 1) // world
    buf = torch.zeros((cap, RESULT_WIDTH), dtype=torch.float32, device=device)
    buf[: len(local)] = local.to(device)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        a, b = shard_range(n_items, world, r)
        parts.append(out[r][: b - a].cpu().numpy())
    return np.concatenate(parts, 0)


# ---- an ncclComm_t for the C ABI's collectives -------------------------------------------------------------------------------
class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]


_rccl = None


def rccl_library():
    """The RCCL this process should talk to: the copy torch already loaded (found in /proc/self/maps and re-opened RTLD_GLOBAL,
    so that csrc/comm.hip's dlsym(RTLD_DEFAULT, "ncclBroadcast") resolves to the SAME library that creates the communicator),
    else librccl.so.1."""
    global _rccl
    if _rccl is None:
        path = None
        try:
            for line in open("/proc/self/maps"):
                f = line.split()[-1]
                if os.path.basename(f).startswith("librccl.so"):
                    path = f
                    break
        except OSError:
            pass
        _rccl = C.CDLL(path or "librccl.so.1", mode=C.RTLD_GLOBAL)
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    return _rccl


def rccl_comm_from_process_group(device):
    """One ncclComm_t (as an int) spanning the ranks of the default torch.distributed group (backend nccl = RCCL), one rank per
    process: rank 0 draws the unique id, the group broadcasts its 128 bytes, every rank calls ncclCommInitRank.  The caller
    must have set the HIP device (torch.cuda.set_device).  Destroy with rccl_comm_destroy."""
    lib = rccl_library()
    world, rank = dist.get_world_size(), dist.get_rank()
    uid = _NcclUniqueId()
    if rank == 0:
        rc = lib.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError("ncclGetUniqueId failed: %d" % rc)
    t = torch.frombuffer(bytearray(C.string_at(C.byref(uid), 128)), dtype=torch.uint8).to(device)     # zeros except on rank 0
    dist.broadcast(t, src=0)
    C.memmove(C.byref(uid), t.cpu().numpy().tobytes(), 128)
    comm = C.c_void_p()
    rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError("ncclCommInitRank failed: %d" % rc)
    return comm.value


def rccl_comm_destroy(comm):
    if comm:
        rccl_library().ncclCommDestroy(C.c_void_p(comm))
