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
"""
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
