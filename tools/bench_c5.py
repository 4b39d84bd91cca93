#!/usr/bin/env python3
"""Config C5 (BASELINE.json configs[4]): batched bin-picking sharded over the GPUs of one node — one process per GPU,
contiguous instance shards, ONE RCCL broadcast of the prepared model (points | normals | FPFH) from rank 0, no
collective on an instance's data path, ONE gather of 19 floats per instance.  Weak scaling: --instances-per-gpu each.

Default workload (--workload tray): every rank sees ITS OWN tray - one 1280x720 frame with --instances-per-gpu small parts (1,024:
a scene cloud of ~530k points), cut into instances by one uint16 label image, tools/c5_tray.py - of the same part, whose scan
rank 0 prepares and broadcasts.  Rank 0 reports the share of all gathered poses within c5_tray.MAX_ANGLE of their ground truth
(the reference's own algorithm loses a few of these 25-pixel parts) and fails under 88 %.
--workload relief: tools/bench_batch.py's large instances (own frame and ~190k-pixel mask each), where every pose must be right.

    python tools/bench_c5.py --instances-per-gpu 64                                   # one GPU
    python tools/bench_c5.py --gpus 8 --instances-per-gpu 1024                        # the C5 shape: starts its 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/bench_c5.py --instances-per-gpu 1024                                    # the same under an outer launcher

Rank 0 prints one JSON line (aggregate instances/s, broadcast and gather times, registration quality)."""
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


def tray_share(tdv, synth, sharding, ctx, torch, dist, dev, rank, world, order, cdev=None, instances_per_gpu=1024, hyps=10000, icp_iters=50, c_abi=None):
    """One rank's 1,024-instance tray through ONE tdv_register_batch_dev call, the model moved once from rank 0 and the results
    gathered once.  c_abi (default: whenever the job is a real RCCL job, i.e. a torch.distributed group with backend nccl exists and
    the collectives' tensors live on the GPU): the two steps go through the C ABI - tdv_broadcast_model / tdv_gather_results on an ncclComm_t made from the
    process group (sharding.rccl_comm_from_process_group) - else through torch.distributed (sharding.broadcast_model /
    gather_results: the one-GPU gloo rehearsal, where RCCL cannot hold two ranks on one device).  Returns (dict for rank 0 | None, ok)."""
    cdev = cdev or dev                                                 # where torch's collectives' tensors live (the CPU in a one-GPU rehearsal)
    if c_abi is None:
        c_abi = cdev.type == "cuda" and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"    # a real RCCL job (any size)
    c5 = importlib.import_module("c5_tray")
    B = instances_per_gpu
    wl = c5.build(tdv, synth, ctx, B, dev, order=order, hyps=hyps, icp_iters=icp_iters, pose_seed=1000 + rank)
    # the model: prepared on rank 0 (every rank's build made one of the same part; only rank 0's is used), broadcast once
    d_mx, d_mn, d_mf, nm = wl["model"]
    comm = sharding.rccl_comm_from_process_group(dev) if c_abi else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if c_abi:
        cap = int(d_mx.shape[0])                                       # every rank built the same part: same capacity everywhere
        if rank != 0:
            d_mx.zero_(); d_mn.zero_(); d_mf.zero_()                   # what arrives is rank 0's model, not this rank's own copy
            torch.cuda.synchronize()
        nm = ctx.broadcast_model(comm, 0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), cap, nm if rank == 0 else 0)
        wl["model"] = (d_mx, d_mn, d_mf, nm)
    else:
        pack = torch.cat([d_mx[:nm], d_mn[:nm], d_mf[:nm]], 1).contiguous() if rank == 0 else None
        if world > 1:
            nmt = torch.tensor([nm if rank == 0 else 0], dtype=torch.int64, device=cdev); dist.broadcast(nmt, src=0); nm = int(nmt.item())
        model = sharding.broadcast_model(pack.to(cdev) if pack is not None else None, nm, cdev).to(dev)
        wl["model"] = (model[:, 0:3].contiguous(), model[:, 3:6].contiguous(), model[:, 6:39].contiguous(), nm)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    c5.run(ctx, wl)                                                    # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = c5.run(ctx, wl)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    tt = torch.tensor([t_local], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ang = c5.angles(synth, wl, res).astype(np.float32)
    t1 = time.perf_counter()
    if c_abi:
        local = []
        for i, r in enumerate(res):
            c = tdv.InstanceResultC()
            tc = tdv.to_colmajor16(r["T"])
            for k in range(16): c.T[k] = float(tc[k])
            c.fitness = float(r["fitness"]); c.rmse = float(ang[i])   # the rmse slot carries the angle to the ground truth
            c.coarse_fitness = float(r["coarse_fitness"]); c.coarse_inliers = int(r["coarse_inliers"]); c.icp_iterations = int(r["icp_iterations"])
            c.n_points = int(r["n_points"]); c.n_voxels = int(r["n_voxels"]); c.status = int(r["status"])
            local.append(c)
        allc = ctx.gather_results(comm, local, B, world)
        gather_ms = (time.perf_counter() - t1) * 1e3
        assert len(allc) == B * world and all(a.status >= 0 for a in allc)
        angles = np.array([a.rmse for a in allc], np.float32); n_results = len(allc)
        sharding.rccl_comm_destroy(comm)
    else:
        local = np.stack([sharding.encode_result(r["T"], r["fitness"], ang[i], r["coarse_inliers"]) for i, r in enumerate(res)])
        allres = sharding.gather_results(local, B * world, cdev)      # slot 17 carries the angle to the ground truth instead of the rmse
        gather_ms = (time.perf_counter() - t1) * 1e3
        angles = allres[:, 17] if rank == 0 else None; n_results = len(allres) if rank == 0 else 0
    if rank != 0:
        return None, True
    elapsed = float(tt.item())
    share = float((angles <= c5.MAX_ANGLE).mean())
    out = dict(config="C5: %d GPU(s) x %d instances, each rank one %dx%d frame cut by a uint16 label image (scene cloud %d points) vs one %d-pt model broadcast once"
                      % (world, B, wl["sc"]["width"], wl["sc"]["height"], int(sum(r["n_points"] for r in res)), nm),
               n_gpus=world, instances=B * world, wall_s=elapsed, instances_per_s=B * world / elapsed, per_gpu_instances_per_s=B / t_local,
               model_bcast_ms=bcast_ms, gather_ms=gather_ms, results_gathered=n_results, registered_share=share,
               collectives="tdv_broadcast_model + tdv_gather_results (C ABI) on an ncclComm_t over RCCL" if c_abi else
                           ("torch.distributed (%s)" % dist.get_backend() if world > 1 else "none (one rank)"),
               max_angle_rad=c5.MAX_ANGLE, median_angle_to_gt_rad=float(np.median(angles)), scaling="weak")
    return out, share >= 0.88


def tray(args, tdv, synth, sharding, ctx, torch, dist, dev, rank, world, order, cdev=None):
    out, ok = tray_share(tdv, synth, sharding, ctx, torch, dist, dev, rank, world, order, cdev, args.instances_per_gpu, args.hyps, args.icp_iters,
                         c_abi=False if args.torch_collectives else (True if args.c_abi else None))
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks to start when no launcher is around this script (0: take WORLD_SIZE, else 1)")
    ap.add_argument("--workload", choices=["tray", "relief"], default="tray")
    ap.add_argument("--instances-per-gpu", type=int, default=1024)
    ap.add_argument("--frames-per-gpu", type=int, default=64, help="distinct poses rendered per rank; instances cycle through them (bounds the frame memory at 1024 instances)")
    ap.add_argument("--hyps", type=int, default=10000)
    ap.add_argument("--icp-iters", type=int, default=50)
    ap.add_argument("--voxel-px", type=float, default=1.2)
    ap.add_argument("--max-angle", type=float, default=1e-2)
    ap.add_argument("--torch-collectives", action="store_true", help="tray workload: move the model and the results with torch.distributed instead of the C ABI's collectives")
    ap.add_argument("--c-abi", action="store_true", help="tray workload: the C ABI's collectives even with ONE rank (a one-rank RCCL process group is created): what a one-GPU box can "
                                                         "check of the N > 1 path - library resolution, ncclCommInitRank from the group's id, tdv_broadcast_model, tdv_gather_results")
    args = ap.parse_args()
    launch = importlib.import_module("3dvision_amd.launch")
    if args.gpus > 1 and not launch.in_rendezvous():      # same self-launch as bench.py: the parent never touches HIP
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus and args.gpus != world:
        raise SystemExit("--gpus %d but %d rank(s) joined the job" % (args.gpus, world))
    # TDV_BENCH_REHEARSE=1 (as in bench.py): on a ONE-GPU box the ranks share GPU 0 and rendezvous over gloo with host tensors, so
    # that every line of the N > 1 path except RCCL itself has run before a multi-GPU node sees it.  Never the default.
    rehearse = os.environ.get("TDV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev
    if world > 1 or args.c_abi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        if rehearse: dist.init_process_group("gloo", rank=rank, world_size=world)
        else: dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    tdv = importlib.import_module("3dvision_amd")
    synth = importlib.import_module("3dvision_amd.synth")
    sharding = importlib.import_module("3dvision_amd.sharding")
    bb = importlib.import_module("bench_batch")
    ctx = tdv.Context(local_rank)
    order = tdv.TDV_VOXEL_ORDER_REFERENCE
    if args.workload == "tray":
        return tray(args, tdv, synth, sharding, ctx, torch, dist, dev, rank, world, order, cdev)

    n_total = args.instances_per_gpu * world
    a, b = sharding.shard_range(n_total, world, rank)
    B = b - a
    F = max(1, min(args.frames_per_gpu, B))
    # this rank's frames: poses a + 0 .. a + F - 1 (every rank renders its own; in production each rank receives its frames)
    wl = bb.build_workload(tdv, synth, ctx, F, args.voxel_px, 448, 3, order, dev, first_pose=a)
    frame_of = (np.arange(B) % F).astype(np.int32)
    d_masks = wl["masks"][torch.from_numpy(frame_of.astype(np.int64)).to(dev)].contiguous() if B != F else wl["masks"]

    # model: prepared on rank 0 (build_workload did it on every rank; only rank 0's copy is used), broadcast once
    t0 = time.perf_counter()
    d_mx, d_mn, d_mf, nm = wl["model"]
    pack = torch.cat([d_mx[:nm], d_mn[:nm], d_mf[:nm]], 1).contiguous() if rank == 0 else None
    if world > 1:
        nmt = torch.tensor([nm if rank == 0 else 0], dtype=torch.int64, device=dev); dist.broadcast(nmt, src=0); nm = int(nmt.item())
    model = sharding.broadcast_model(pack, nm, dev)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    d_mx = model[:, 0:3].contiguous(); d_mn = model[:, 3:6].contiguous(); d_mf = model[:, 6:39].contiguous()

    def params(n, fmap):
        return tdv.batch_params(width=bb.W, height=bb.H, scale_to_meters=bb.SCALE, fx=bb.F, fy=bb.F, cx=bb.CX, cy=bb.CY, zmax=bb.ZMAX,
                                voxel_size=wl["voxel"], ransac_max_iterations=args.hyps, icp_max_iterations=args.icp_iters, voxel_order=order,
                                n_frames=F, frame_of_instance=fmap)
    ctx.register_batch_dev(wl["depth"].data_ptr(), None, d_masks.data_ptr(), min(B, 2), params(min(B, 2), frame_of[:min(B, 2)]),
                           d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)  # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ctx.register_batch_dev(wl["depth"].data_ptr(), None, d_masks.data_ptr(), B, params(B, frame_of), d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t1 = time.perf_counter()
    ang = np.array([synth.pose_error(r["T"], wl["T_gt"][frame_of[i]])[0] for i, r in enumerate(res)], np.float32)
    local = np.stack([sharding.encode_result(r["T"], r["fitness"], ang[i], r["coarse_inliers"]) for i, r in enumerate(res)]) if res else np.zeros((0, 19), np.float32)
    allres = sharding.gather_results(local, n_total, dev)      # slot 17 carries the angle to the ground truth instead of the rmse
    gather_ms = (time.perf_counter() - t1) * 1e3
    bad = 0
    if rank == 0:
        elapsed = float(tt.item())
        angles = allres[:, 17]
        bad = int((~(angles <= args.max_angle)).sum())
        print(json.dumps(dict(config="C5: %d GPUs x %d instances (%d distinct poses per GPU, ~%d px masks, ~%d voxels) vs one %d-pt model broadcast once"
                                     % (world, args.instances_per_gpu, F, int(np.mean(wl["mask_px"])), res[0]["n_voxels"], nm),
                              n_gpus=world, instances=n_total, wall_s=elapsed, instances_per_s=n_total / elapsed,
                              ms_per_instance_per_gpu=elapsed / args.instances_per_gpu * 1e3, model_bcast_ms=bcast_ms, gather_ms=gather_ms,
                              results_shape=list(allres.shape), registered=n_total - bad, max_angle_to_gt_rad=float(angles.max()), scaling="weak")))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
