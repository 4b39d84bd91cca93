#!/usr/bin/env python3
"""Config C5 (BASELINE.json configs[4]): batched bin-picking sharded over the GPUs of one node — one process per GPU,
contiguous instance shards, ONE RCCL broadcast of the prepared model (points | normals | FPFH) from rank 0, no
collective on an instance's data path, ONE gather of 19 floats per instance.  Weak scaling: --instances-per-gpu each.

    python tools/bench_c5.py --instances-per-gpu 64                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/bench_c5.py --instances-per-gpu 1024                                    # the C5 shape

Rank 0 prints one JSON line (aggregate instances/s, broadcast and gather times, a result checksum)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances-per-gpu", type=int, default=64)
    ap.add_argument("--model-points", type=int, default=10000)
    ap.add_argument("--hyps", type=int, default=10000)
    ap.add_argument("--icp-iters", type=int, default=50)
    ap.add_argument("--voxel", type=float, default=0.0005)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    tdv = importlib.import_module("3dvision_amd")
    synth = importlib.import_module("3dvision_amd.synth")
    sharding = importlib.import_module("3dvision_amd.sharding")
    render = importlib.import_module("bench_batch").render
    ctx = tdv.Context(local_rank)

    # the frame (every rank renders the same scene here; in production each rank receives its frame / its masks)
    w, h, f = 1280, 720, 1500.0
    cx, cy = w / 2.0, h / 2.0
    dense, _ = synth.sample_object(3000000, 42)
    T = synth.make_transform([0.2, 1.0, 0.3], 35.0, (0.0, 0.0, 0.45))
    z = render(synth, dense, T, f, cx, cy, w, h)
    hit = np.isfinite(z)
    depth = np.zeros((h, w), np.uint16); depth[hit] = np.round(z[hit] * 1000.0).astype(np.uint16)
    label = np.where(hit, 1, 0).astype(np.uint8)          # one label image instead of B stacked masks
    n_total = args.instances_per_gpu * world
    a, b = sharding.shard_range(n_total, world, rank)
    B = b - a
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_masks = torch.from_numpy(np.repeat(np.where(hit, 255, 0).astype(np.uint8)[None], B, 0)).to(dev)

    # model: prepared on rank 0, broadcast once
    t0 = time.perf_counter()
    pack = None; nm = 0
    if rank == 0:
        raw, _ = synth.sample_object(args.model_points * 3, 7)
        d_raw = torch.from_numpy(raw).to(dev)
        d_mx = torch.empty_like(d_raw); d_mn = torch.empty_like(d_raw); d_mf = torch.empty((len(raw), 33), dtype=torch.float32, device=dev)
        nm = ctx.prepare_model_dev(d_raw.data_ptr(), len(raw), float(synth.mean_spacing(args.model_points)), 30, 5.0,
                                   d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr())
        pack = torch.cat([d_mx[:nm], d_mn[:nm], d_mf[:nm]], 1).contiguous()
    if world > 1:
        nmt = torch.tensor([nm], dtype=torch.int64, device=dev); dist.broadcast(nmt, src=0); nm = int(nmt.item())
    model = sharding.broadcast_model(pack, nm, dev)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    d_mx = model[:, 0:3].contiguous(); d_mn = model[:, 3:6].contiguous(); d_mf = model[:, 6:39].contiguous()

    prm = tdv.batch_params(width=w, height=h, fx=f, fy=f, cx=cx, cy=cy, zmax=1.5, voxel_size=args.voxel,
                           ransac_max_iterations=args.hyps, ransac_confidence=2.0, icp_max_iterations=args.icp_iters, icp_distance_factor=4.0)
    ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), min(B, 2), prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)  # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), B, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t1 = time.perf_counter()
    local = np.stack([sharding.encode_result(r["T"], r["fitness"], r["rmse"], r["coarse_inliers"]) for r in res]) if res else np.zeros((0, 19), np.float32)
    allres = sharding.gather_results(local, n_total, dev)
    gather_ms = (time.perf_counter() - t1) * 1e3
    if rank == 0:
        elapsed = float(tt.item())
        print(json.dumps(dict(config="C5-style: %d GPUs x %d instances (292k-px masks, %d-pt model), model broadcast once" % (world, args.instances_per_gpu, nm),
                              n_gpus=world, instances=n_total, wall_s=elapsed, instances_per_s=n_total / elapsed,
                              ms_per_instance_per_gpu=elapsed / args.instances_per_gpu * 1e3, model_bcast_ms=bcast_ms, gather_ms=gather_ms,
                              results_shape=list(allres.shape), identical_results=bool((allres == allres[0]).all()),
                              scaling="weak")))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
