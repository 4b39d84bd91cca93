#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X:
RANSAC hypotheses/s + ICP iterations/s at 200k-point clouds, with the roofline of the dominant
kernel and a CPU baseline (the oracle) timed beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one pass of the hot path over one batch of synthetic input at the headline size
(N_s = N_t = 200,000, the reference's settings: ICP threshold 0.4 x voxel, RANSAC threshold 1.5 x voxel):   1 ICP iteration (nearest-neighbour correspondences + normal equations + 6x6 solve +
on-device transform update)  +  20,000 RANSAC hypotheses (3-point Kabsch SVD each, every
hypothesis scored against all 200,000 correspondences).  20,000 hyps per ICP iteration is the ratio
of BASELINE.json's two targets (1e6 hyps/s : 50 iters/s), so `value` >= 50 steps/s means both
targets are met at once.  At this size the ICP correspondences come from the exact box-pruned search (same
correspondences as the reference's brute-force scan, bit for bit: tests/test_gpu_fullsize.py); the brute-force scan
is timed after the K steps as a supplementary roofline entry.  The two rates are also reported separately (icp_iters_per_s,
ransac_hyps_per_s), each from its own synchronized sub-region of the same K steps.

Inputs are resident in HBM before the timed region (torch CUDA tensors; the C ABI's *_dev entry
points take their device pointers).  Multi-GPU: instances shard across ranks (one scene/model pair
per rank, weak scaling); the reference model is broadcast once from rank 0 over RCCL before the
timed region; no collective sits on the data path.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HYPS_PER_STEP = 20000
VALU_PEAK_TOPS = 78.6  # f32 VALU lane-ops/s without FMA: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (157.3 TFLOP/s counts FMA as 2)
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=200000, help="N_s = N_t (headline: 200000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-operators", action="store_true", help="skip the per-operator block measured after the timed region")
    ap.add_argument("--c5-timeout-s", type=float, default=300.0, help="N > 1: deadline of the C5 share; past it the line is printed without C5")
    ap.add_argument("--c5-instances", type=int, default=1024, help="N > 1: instances per rank of the C5 tray measured after the timed region")
    ap.add_argument("--cpu-budget-s", type=float, default=16.0)
    ap.add_argument("--min-region-ms", type=float, default=100.0,
                    help="the timed region is repeated (whole multiples of K steps) until it is at least this long; 0 = exactly K steps once")
    ap.add_argument("--launch-check", action="store_true",
                    help="CPU-only rehearsal of the N-rank launch (gloo, no GPU work): the same spawn / barrier / max-over-ranks / one-line skeleton")
    return ap.parse_args()


def launch_check(args):
    """The N > 1 skeleton of this script without a GPU: ranks rendezvous over gloo, rank 0 'broadcasts the model', every rank
    times K stand-in steps between barriers, the maximum over ranks is taken and rank 0 prints ONE line.  What
    tests/test_bench_launch.py runs here (no GPU in the build container) to prove that `python bench.py --gpus N` starts N ranks."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = dist.get_world_size() if world > 1 else 1
    if args.gpus != seen:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, seen))
    sharding = importlib.import_module("3dvision_amd.sharding")
    pack = torch.arange(64 * sharding.MODEL_PACK_WIDTH, dtype=torch.float32).reshape(64, -1) if rank == 0 else None
    t0 = time.perf_counter()
    model = sharding.broadcast_model(pack, 64, torch.device("cpu"))
    bcast_ms = (time.perf_counter() - t0) * 1e3
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    acc = 0.0
    for _ in range(args.steps):
        acc += float(model.sum())
    if world > 1:
        dist.barrier()
    times = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    mine = torch.tensor([args.steps / max(float(times[0]), 1e-9), float(rank)], dtype=torch.float64)
    per_rank = [torch.zeros_like(mine) for _ in range(seen)]
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    if rank == 0:
        print(json.dumps({"metric": "launch check (no GPU work)", "launch_check": True, "n_gpus": seen, "steps": args.steps, "warmup": args.warmup,
                          "value": args.steps * seen / max(float(times[0]), 1e-9), "unit": "stand-in steps/s", "scaling": "weak",
                          "model_bcast_ms": bcast_ms, "model_checksum": float(model.sum()),
                          "per_rank": [{"rank": int(x[1]), "steps_per_s": float(x[0])} for x in per_rank]}))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(orc, src, tgt, nrm, corr, T0, thr, voxel, budget_s):
    """The oracle (CPU restatement of registration.cpp, g++ -O3) on a bounded sample of the same workload: ICP
    NN+accumulate for a slice of the sources against ALL targets, and RANSAC for a few hypotheses against ALL points;
    converted to whole-iteration / per-hypothesis rates.  First on one thread (the reference's per-instance code path is
    single-threaded), then at the reference's own parallelism: one instance per pool thread, 8 threads
    (include/pipeline_config.hpp:62, src/pipeline.cpp:321-327) — the same sample on min(8, host cores) threads at once."""
    import threading
    ns = len(src)
    # calibrate with a small slice, then size the sample to about half the budget each
    t0 = time.perf_counter(); orc.icp_correspondences(src[:200], tgt, nrm, T0, thr); t_small = time.perf_counter() - t0
    m = int(min(ns, max(500, 200 * (budget_s * 0.5) / max(t_small, 1e-6))))
    t0 = time.perf_counter(); orc.icp_correspondences(src[:m], tgt, nrm, T0, thr); t_icp = time.perf_counter() - t0
    icp_iters_per_s = 1.0 / (t_icp * ns / m)
    t0 = time.perf_counter(); orc.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=20, confidence=2.0); t_small = time.perf_counter() - t0
    h = int(max(50, 20 * (budget_s * 0.5) / max(t_small, 1e-6)))
    t0 = time.perf_counter(); orc.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=h, confidence=2.0); t_r = time.perf_counter() - t0
    hyps_per_s = h / t_r
    step_s = 1.0 / icp_iters_per_s + HYPS_PER_STEP / hyps_per_s
    # the reference's thread pool: T instances at once, one per thread (ctypes releases the GIL inside the oracle)
    T = max(1, min(8, os.cpu_count() or 1))
    mt = max(200, m // 2); ht = max(20, h // 2)
    spent = [0.0] * T

    def work(k):
        a = time.perf_counter()
        orc.icp_correspondences(src[:mt], tgt, nrm, T0, thr)
        b = time.perf_counter()
        orc.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=ht, confidence=2.0)
        spent[k] = (b - a) * ns / mt + (time.perf_counter() - b) * HYPS_PER_STEP / ht      # seconds per step on this thread
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    t_par = time.perf_counter() - t0
    pool_steps_per_s = sum(1.0 / x for x in spent)
    return {
        "value": 1.0 / step_s, "unit": "steps/s", "cores": 1, "kind": "port",
        "icp_iters_per_s": icp_iters_per_s, "ransac_hyps_per_s": hyps_per_s,
        "sample": "ICP: %d of %d sources x all %d targets, 1 iteration (%.1f s); RANSAC: %d hypotheses x all %d points (%.1f s); "
                  "oracle/liboracle.so (g++ -O3, no -march=native), single thread" % (m, ns, len(tgt), t_icp, h, ns, t_r),
        "instance_parallel": {"value": pool_steps_per_s, "unit": "steps/s", "cores": T, "kind": "port",
                              "sample": "the reference's thread pool shape: %d threads, one instance each, every thread %d sources x all targets + %d hypotheses (%.1f s wall)"
                                        % (T, mt, ht, t_par)},
    }


def main():
    args = parse()
    launch = importlib.import_module("3dvision_amd.launch")
    if args.gpus > 1 and not launch.in_rendezvous():
        # `python bench.py --gpus N` without a launcher around it: start the N ranks ourselves.  Nothing in this process has
        # touched HIP yet (torch is not even imported), children are fresh interpreters, rank 0's line is relayed once.
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.launch_check:
        return launch_check(args)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    # TDV_BENCH_FORCE_DIST=1 (tests/test_gpu_bench_dist.py): take the N > 1 code path with the ONE rank a one-GPU box allows - a real RCCL
    # process group, its collectives, the C5 share through the C ABI's collectives - so that every line of it has run on hardware before
    # the driver's multi-GPU node sees it (RCCL across two devices is the one thing this cannot show).  Never set by the driver.
    distributed = env_world > 1 or os.environ.get("TDV_BENCH_FORCE_DIST") == "1"
    # Rehearsal on a ONE-GPU box (TDV_BENCH_REHEARSE=1): the ranks share GPU 0 and rendezvous over gloo with host tensors - RCCL
    # refuses two ranks on one device - so that every line of the N > 1 path below except RCCL itself runs before the driver's
    # multi-GPU node sees it.  Never the default: a real run is one rank per GPU over RCCL / xGMI.
    rehearse = os.environ.get("TDV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if env_world == 1:
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        if rehearse: dist.init_process_group("gloo", rank=rank, world_size=env_world)
        else: dist.init_process_group("nccl", rank=rank, world_size=env_world, device_id=torch.device("cuda", local_rank))
    world = dist.get_world_size() if distributed else 1      # the ranks RCCL actually joined: what n_gpus reports
    if args.gpus != world:
        raise SystemExit("--gpus %d but %d rank(s) joined the job" % (args.gpus, world))

    tdv = importlib.import_module("3dvision_amd")
    synth = importlib.import_module("3dvision_amd.synth")
    assert os.path.exists(tdv.LIB_PATH), "lib3dvision_hip.so missing — run __graft_entry__.build(); there is no CPU fallback"
    ctx = tdv.Context(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev      # where the (tiny) collectives' tensors live

    n = args.points
    voxel = float(np.float32(synth.mean_spacing(n)))
    icp_thr = voxel * 0.4          # registration.icp_distance_factor (include/pipeline_config.hpp:28, src/pipeline.cpp:104)
    # --- reference model: generated on rank 0, broadcast to every rank over RCCL (xGMI) ---------
    t_b0 = time.perf_counter()
    sharding = importlib.import_module("3dvision_amd.sharding")
    pack = None
    if rank == 0:
        tgt_np, nrm_np = synth.sample_object(n, 42)
        pack = sharding.pack_model(tgt_np, nrm_np, synth.random_features(n, 42))  # points | normals | FPFH = 156 B per point
    model = sharding.broadcast_model(pack, n, cdev).to(dev)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t_b0) * 1e3
    d_tgt = model[:, :3].contiguous(); d_nrm = model[:, 3:6].contiguous()
    # --- this rank's scene instance ---------------------------------------------------------------
    src_np, T_gt = synth.make_scene(n, 42 + rank)
    T0 = synth.perturb(T_gt, 42 + rank, angle_deg=0.3, trans=0.0005)   # a coarse pose as RANSAC leaves it: inside ICP's 0.4-voxel basin
    d_src = torch.from_numpy(src_np).to(dev)
    # correspondences for RANSAC: true nearest model point (found with the GPU NN scan under T_gt)
    # for half of the points, a random model point for the rest (FPFH-quality matches)
    nn = ctx.icp_correspondences(src_np, d_tgt.cpu().numpy(), T_gt, 1.0)["corr"]
    rng = np.random.Generator(np.random.PCG64(1234 + rank))
    corr_np = np.where(rng.random(n) < 0.5, nn, rng.integers(0, n, n)).astype(np.int32)
    d_corr = torch.from_numpy(corr_np).to(dev)
    torch.cuda.synchronize()

    def run(steps, timed):
        """steps ICP iterations + steps*HYPS_PER_STEP hypotheses; returns (t_icp, t_ransac) wall seconds."""
        torch.cuda.synchronize()
        # HIP-event timers bracket the DOMINANT kernel's dispatches only (the scoring kernel, below): two event records around each
        # of the ICP loop's 50-us launches would tax the very loop being timed; the search kernel's duration is measured right after
        # the timed region instead (icp_probe)
        if timed: ctx.timing_enable(False)
        a = time.perf_counter()
        r_icp = ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, icp_thr, steps, True, fixed_iterations=True)
        torch.cuda.synchronize()
        b = time.perf_counter()
        if timed: ctx.timing_enable(True)
        r_rs = ctx.ransac_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, None, None, d_corr.data_ptr(), voxel, steps * HYPS_PER_STEP, 2.0, 42)
        torch.cuda.synchronize()
        c = time.perf_counter()
        return b - a, c - b, r_icp, r_rs

    region_est = None
    if args.warmup > 0:
        run(args.warmup, False)
        w = run(args.warmup, False)                  # second pass: workspaces sized, clocks up - its wall time sizes the timed region
        region_est = (w[0] + w[1]) * args.steps / args.warmup
    # A timed region shorter than ~100 ms is dominated by box noise (and invisible to a 1 Hz utilisation sampler): the K-step
    # pass is repeated `regions` times inside ONE barrier/sync bracket; every figure below is per step over all K x regions steps.
    regions = 1
    if args.min_region_ms > 0 and region_est is not None:
        # (the estimate comes from W-step passes, whose per-call overheads weigh more than in a K-step pass: 25 % of margin, so that the
        #  region does not end up a few milliseconds under the floor)
        regions = max(1, int(np.ceil(1.25 * args.min_region_ms * 1e-3 / max(region_est, 1e-6))))
    rg = torch.tensor([regions], dtype=torch.int64, device=cdev)
    if distributed:
        dist.all_reduce(rg, op=dist.ReduceOp.MAX)      # same count on every rank
    regions = int(rg.item())
    ctx.timing_enable(True)
    ctx.timing_read(tdv.TIMER_ICP_NN); ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    t_icp = t_rs = 0.0
    for _ in range(regions):
        a_, b_, r_icp, r_rs = run(args.steps, True)
        t_icp += a_; t_rs += b_
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t_start
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    steps_timed = args.steps * regions
    sc_ms, sc_launches = ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
    icp_search_used = ctx.last_icp_search()   # before the supplementary scan below changes it
    # the search kernel of the timed region, timed on its own: one more ICP call of the same shape, outside `value`
    ctx.timing_read(tdv.TIMER_ICP_NN)
    ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, icp_thr, args.steps, True, fixed_iterations=True)
    nn_ms, nn_launches = ctx.timing_read(tdv.TIMER_ICP_NN)
    ctx.timing_enable(False)
    rescore_share = ctx.last_ransac_rescore()  # fast scoring pass: share of the tests scored a second time exactly (whole waves on a pair of points); -1: exact mode
    scored_share = ctx.last_ransac_scored()    # share of the (hypothesis, point) tests evaluated at all (exact bail-out: DESIGN.md 4)

    times = torch.tensor([elapsed, t_icp, t_rs], dtype=torch.float64, device=cdev)
    if distributed:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    elapsed, t_icp, t_rs = [float(x) for x in times.cpu()]
    # per-rank view (rank 0 prints it at N > 1): own steps/s, the dominant kernel's average dispatch and executed-op fraction
    ops_mine = ((16.6 + 28.0 * rescore_share) if rescore_share >= 0.0 else 28.0) * scored_share * n * float(steps_timed) * HYPS_PER_STEP
    mine = torch.tensor([rank, steps_timed / t_local, sc_ms / max(sc_launches, 1), ops_mine / max(sc_ms * 1e-3, 1e-12) / 1e12 / VALU_PEAK_TOPS,
                         bcast_ms], dtype=torch.float64, device=cdev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if distributed:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = [[float(v) for v in x.cpu()] for x in per_rank]

    # supplementary: the brute-force NN scan (the reference kernel's algorithm) on the same inputs, outside `value`
    ctx.timing_enable(True)
    ctx.set_icp_search("brute")
    ctx.timing_read(tdv.TIMER_ICP_NN)
    ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, icp_thr, 3, True, fixed_iterations=True)
    bf_ms, bf_launches = ctx.timing_read(tdv.TIMER_ICP_NN)
    ctx.timing_enable(False)
    BF_ITERS = 10                                     # whole iterations with the reference's scan (search + sums + solve), wall clock
    torch.cuda.synchronize(); t_bf0 = time.perf_counter()
    ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, icp_thr, BF_ITERS, True, fixed_iterations=True)
    torch.cuda.synchronize(); bf_iters_per_s = BF_ITERS / (time.perf_counter() - t_bf0)
    ctx.set_icp_search("auto")
    # supplementary: the same ICP call with the reference's accumulation order (bit-equal to the CPU path; a serial chain per iteration)
    REF_ITERS = 10
    ctx.set_icp_accumulation("reference")
    ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, icp_thr, 2, True, fixed_iterations=True)
    torch.cuda.synchronize(); t_r0 = time.perf_counter()
    ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, icp_thr, REF_ITERS, True, fixed_iterations=True)
    torch.cuda.synchronize(); ref_iters_per_s = REF_ITERS / (time.perf_counter() - t_r0)
    ctx.set_icp_accumulation("tree")

    # supplementary: RANSAC with EVERY (hypothesis, point) test evaluated (a traced call: the exact bail-out is off), outside `value`
    full_hyps = 4 * 65536
    ctx.timing_enable(True); ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
    torch.cuda.synchronize(); t_f0 = time.perf_counter()
    r_full = ctx.ransac_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, None, None, d_corr.data_ptr(), voxel, full_hyps, 2.0, 42, trace=True)
    torch.cuda.synchronize(); t_full = time.perf_counter() - t_f0
    full_ms, full_launches = ctx.timing_read(tdv.TIMER_RANSAC_SCORE); ctx.timing_enable(False)
    full_rescore = ctx.last_ransac_rescore(); full_scored = ctx.last_ransac_scored()

    # N > 1: config C5 (BASELINE.json configs[4]) rides in the same line - every rank registers its own 1,024-instance tray, the model
    # moved once by tdv_broadcast_model and the results gathered once by tdv_gather_results (C ABI, ncclComm_t over RCCL); in the one-GPU
    # rehearsal (gloo) the same two steps go through torch.distributed.  Outside `value`.
    # The share has never met more than one device (no multi-GPU box in the builder's reach), so it must not be able to take the headline
    # down with it: it runs on a thread of its own with a deadline.  An exception becomes `c5_error` in the line; a rank still inside a
    # collective at the deadline is abandoned - the line is printed without C5 and every rank leaves through os._exit (all ranks share the
    # deadline, so they leave together).
    c5_out = None; c5_error = None; c5_hung = False
    if distributed and not args.no_operators:
        import threading
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_c5
        box = {}

        def c5_work():
            try:
                torch.cuda.set_device(local_rank)                     # (the current device is per host thread)
                c5_ctx = tdv.Context(local_rank)
                box["out"], box["ok"] = bench_c5.tray_share(tdv, synth, sharding, c5_ctx, torch, dist, dev, rank, world, tdv.TDV_VOXEL_ORDER_REFERENCE, cdev,
                                                            instances_per_gpu=args.c5_instances)
                c5_ctx.close()
            except BaseException as e:                                 # noqa: BLE001 - reported, never fatal for the headline
                box["err"] = "%s: %s" % (type(e).__name__, e)
        th = threading.Thread(target=c5_work, name="c5-share", daemon=True)
        th.start(); th.join(timeout=args.c5_timeout_s)
        if th.is_alive():
            c5_hung = True; c5_error = "the C5 share did not finish within %.0f s (abandoned; the headline above is unaffected)" % args.c5_timeout_s
        elif "err" in box:
            c5_error = box["err"]
        else:
            c5_out = box.get("out")

    if rank == 0:
        steps_total = steps_timed * world
        pairs = float(n) * float(n)
        nn_avg_ms = nn_ms / max(nn_launches, 1)
        sc_avg_ms = sc_ms / max(sc_launches, 1)
        bf_avg_ms = bf_ms / max(bf_launches, 1)
        hyps_total = float(steps_timed) * HYPS_PER_STEP
        pruned_default = icp_search_used in ("pruned", "grid")
        # HBM traffic per launch from the PMC passes (rocprofv3 --pmc cannot run inside this process): taken from the
        # committed summary of the same command when it covers this workload, else null
        pm = None
        try:
            for pm_round in ("r4", "r3", "r2"):
                pm_path = os.path.join(ROOT, "profiles", pm_round, "pmc_summary.json")
                if os.path.exists(pm_path):
                    break
            pm = json.load(open(pm_path))
            if not (pm["workload"]["n_src"] == n and pm["workload"]["n_tgt"] == n):
                pm = None
        except Exception:
            pm = None

        def traffic(kernel):
            try:
                return pm["kernels"]["tdv::" + kernel]["hbm_bytes_per_launch"]
            except Exception:
                return None

        # Scoring kernel of the timed region.  Reference arithmetic (k_ransac_score): per hypothesis-point 18 ops transform +
        # 3 sub + 5 squared norm + 1 compare + 1 count = 28 VALU lane-ops (issued as packed f32 pairs).  Default
        # (k_ransac_score_fast): 16.6 lane-ops per hypothesis-point in the FMA pass (per 8 points 60 packed instructions -
        # 48 fma, 12 add - and 13 scalar-f32 ones) + 28 for every test scored again exactly: a wave re-scores the PAIR of points that
        # one of its 64 hypotheses has inside its rounding band (round 4; the whole 8-point chunk until then)
        # (rescore_share, counted by the kernel).  An FMA is ONE lane-op here, as in the peak (lane-instructions, not flops).
        # Algorithmic HBM bytes per launch: the packed pairs once (24 B per point) + 48 B per hypothesis in, 4 B out.
        # A "launch" is one dispatch of the scoring kernel (what rocprofv3's kernel statistics average over).  With the exact
        # bail-out a batch of hypotheses is two dispatches: every hypothesis over a prefix of the points, then the hypotheses that
        # can still beat the best count of the earlier batches over the rest - together they read the pair array once.
        # (batching rule of csrc/ransac.hip: batches of 65,536 hypotheses, the first one 8,192 when the bail-out is on)
        hyps_call = args.steps * HYPS_PER_STEP        # one RANSAC call per region
        bail = (os.environ.get("TDV_RANSAC_BAILOUT", "1") != "0") and hyps_call > 16384
        sc_batches = regions * ((1 + -(-(hyps_call - 8192) // 65536)) if bail else max(1, -(-hyps_call // 65536)))
        sc_hyps_per_launch = hyps_total / max(sc_launches, 1)
        fast = rescore_share >= 0.0
        ops_per_test = (16.6 + 28.0 * rescore_share) if fast else 28.0
        sc_tops = ops_per_test * scored_share * n * hyps_total / max(sc_ms * 1e-3, 1e-12) / 1e12
        sc_bytes = (24.0 * n * sc_batches + 56.0 * hyps_total) / max(sc_launches, 1)
        sc_kernel = "k_ransac_score_fast" if fast else "k_ransac_score"
        score = {
            "kernel": sc_kernel, "bound": "valu_f32",
            "achieved": sc_tops, "peak": VALU_PEAK_TOPS,
            "unit": ("T VALU lane-ops/s executed: %.1f per hypothesis-point = 16.6 in the FMA pass + 28 x the %.3f of the tests (whole waves on a pair of "
                     "points) scored again with the reference arithmetic; same inlier counts as the 28-op reference arithmetic" % (ops_per_test, rescore_share))
                    if fast else "Tops/s (f32 VALU, FMA contraction forbidden by parity; 28 ops per hypothesis-point)",
            "frac": sc_tops / VALU_PEAK_TOPS, "avg_launch_ms": sc_avg_ms, "launches": sc_launches,
            "hyps_per_launch": sc_hyps_per_launch, "batches": sc_batches, "total_ms": sc_ms,
            "hbm": {"algorithmic_bytes_per_launch": sc_bytes, "achieved": sc_bytes / max(sc_avg_ms * 1e-3, 1e-12) / 1e9,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": sc_bytes / max(sc_avg_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBPS},
            "traffic": traffic(sc_kernel),
        }
        full_ops = (16.6 + 28.0 * full_rescore) if full_rescore >= 0.0 else 28.0
        score["every_test_scored"] = {
            "note": "the same kernel with the bail-out off (a call that asks for the per-iteration trace): %d hypotheses x all %d points, one dispatch per batch" % (full_hyps, n),
            "hyps_per_s": full_hyps / t_full, "avg_launch_ms": full_ms / max(full_launches, 1), "launches": full_launches,
            "scored_share": full_scored, "achieved": full_ops * n * full_hyps / max(full_ms * 1e-3, 1e-12) / 1e12, "peak": VALU_PEAK_TOPS,
            "frac": full_ops * n * full_hyps / max(full_ms * 1e-3, 1e-12) / 1e12 / VALU_PEAK_TOPS,
            "best_inliers": int(r_full.inliers), "best_iteration": int(r_full.best_iteration)}
        if fast:
            score["rescore_share"] = rescore_share
            score["scored_share"] = scored_share
            score["note"] = ("%.3f of the hypothesis-point tests were evaluated: a hypothesis whose count over a prefix of the points plus all remaining points "
                             "cannot exceed the best count of the earlier batches is not scored further (exact: same best hypothesis, count, "
                             "fitness, rmse); `achieved` counts executed lane-ops only" % scored_share)
            score["reference_arithmetic_equivalent_tops"] = 28.0 * n * hyps_total / max(sc_ms * 1e-3, 1e-12) / 1e12
        # NN search of the timed region.  Algorithmic HBM bytes of one ICP iteration: 12*N_s + 24*N_t + 124
        icp_bytes = 12.0 * n + 24.0 * n + 124
        nn_kernel = {"grid": "k_icp_nn_grid", "pruned": "k_icp_nn_pruned"}.get(icp_search_used, "k_icp_nn_scan")
        nn_equiv = 8.0 * pairs / max(nn_avg_ms * 1e-3, 1e-12) / 1e12
        nn = {
            "kernel": nn_kernel, "avg_launch_ms": nn_avg_ms, "launches": nn_launches, "total_ms": nn_ms,
            "hbm": {"algorithmic_bytes_per_launch": icp_bytes, "achieved": icp_bytes / max(nn_avg_ms * 1e-3, 1e-12) / 1e9,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": icp_bytes / max(nn_avg_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBPS},
            "traffic": traffic(nn_kernel),
        }
        if icp_search_used == "grid":
            nn.update({"bound": "latency (hash-grid probes: 8 table entries + the points of the occupied cells per source point)",
                       "search": "grid", "bruteforce_equivalent_tops": nn_equiv,
                       "note": "exact hash-grid search (cells of 2.2 x the acceptance threshold, one lane per source point): identical "
                               "correspondences to the scan; no VALU roofline fraction is claimed for it"})
        elif pruned_default:
            nn.update({"bound": "latency (box walk: few pairs evaluated)", "search": "pruned", "bruteforce_equivalent_tops": nn_equiv,
                       "note": "exact pruned search: evaluates only boxes that can hold a neighbour within the bound, so the "
                               "brute-force pair count / time exceeds the VALU peak; no VALU roofline fraction is claimed for it"})
        else:
            nn.update({"bound": "valu_f32", "achieved": nn_equiv, "peak": VALU_PEAK_TOPS, "frac": nn_equiv / VALU_PEAK_TOPS})
        # brute-force scan (3 sub + 3 mul + 2 add per pair = 8 VALU ops), supplementary
        bf_tops = 8.0 * pairs / max(bf_avg_ms * 1e-3, 1e-12) / 1e12
        brute = {
            "kernel": "k_icp_nn_scan", "bound": "valu_f32", "achieved": bf_tops, "peak": VALU_PEAK_TOPS,
            "unit": "Tops/s (8 ops per pair)", "frac": bf_tops / VALU_PEAK_TOPS, "avg_launch_ms": bf_avg_ms, "launches": bf_launches,
            "traffic": traffic("k_icp_nn_scan"),
            "note": "reference-algorithm scan, timed after the K steps (TDV_ICP_SEARCH=brute makes it the default)",
        }
        dominant, other = (score, nn) if sc_ms >= nn_ms else (nn, score)
        roofline = dict(dominant)
        roofline["traffic_unit"] = "HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, mean over the dispatches of the same command)"
        roofline["traffic_source"] = ("FROM A COMMITTED PROFILE, not measured by this run: profiles/%s/pmc_summary.json (separate rocprofv3 --pmc FETCH_SIZE / "
                                      "WRITE_SIZE passes of this command on the builder's box; PMC counters cannot be read from inside the benchmarked process)" % pm_round) \
            if roofline.get("traffic") is not None else None
        roofline["second_kernel"] = other
        roofline["icp_nn_bruteforce_scan"] = brute
        # BASELINE.json's metric, flat, in the dict the driver keeps: hypotheses/s and iterations/s at 200k points (whole job), each from
        # its own synchronized sub-region of the timed steps; beside them the figures the docs promise next to every headline number
        hyps_per_s = steps_timed * HYPS_PER_STEP * world / t_rs
        iters_per_s = steps_timed * world / t_icp
        roofline.update({
            "ransac_hyps_per_s": hyps_per_s, "icp_iters_per_s": iters_per_s,
            "ransac_hyps_per_s_every_test_scored": full_hyps / t_full,
            "icp_iters_per_s_bruteforce_scan": bf_iters_per_s,
            "icp_iters_per_s_reference_order_sums": ref_iters_per_s,
            "icp_search": icp_search_used,
            "target_ransac_hyps_per_s": 1e6, "target_icp_iters_per_s": 50,
            "hbm_algorithmic_bytes_per_launch": score["hbm"]["algorithmic_bytes_per_launch"] if dominant is score else nn["hbm"]["algorithmic_bytes_per_launch"],
            "hbm_achieved_GBps": dominant["hbm"]["achieved"], "hbm_frac": dominant["hbm"]["frac"],
            "every_test_scored_frac": score["every_test_scored"]["frac"], "every_test_scored_avg_launch_ms": score["every_test_scored"]["avg_launch_ms"],
            "bruteforce_scan_frac": brute["frac"], "bruteforce_scan_avg_launch_ms": brute["avg_launch_ms"],
            "second_kernel_name": other["kernel"], "second_kernel_avg_launch_ms": other["avg_launch_ms"],
        })
        out = {
            "metric": "RANSAC hyps/s + ICP iters/s @ 200k-pt clouds",
            "value": steps_total / elapsed,
            "unit": "steps/s (1 step = 1 ICP iteration + %d RANSAC hypotheses, N_s=N_t=%d)" % (HYPS_PER_STEP, n),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_regions": regions, "steps_timed": steps_timed, "timed_region_ms": elapsed * 1e3,
            "ms_per_step": elapsed / steps_timed * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "icp_iters_per_s": iters_per_s,
            "ransac_hyps_per_s": hyps_per_s,
            "icp_iters_per_s_bruteforce_scan": bf_iters_per_s,
            "icp_iters_per_s_reference_order_sums": ref_iters_per_s,
            "targets": {"icp_iters_per_s": 50, "ransac_hyps_per_s": 1e6},
            "config": {"workload": "headline: N_s=N_t=%d, point-to-plane ICP (fixed %d iterations per call, %s correspondence search) + RANSAC scoring (%d hyps per call), "
                                   "%d call(s) of each in the timed region, one instance pair per GPU"
                                   % (n, args.steps, {"grid": "exact hash-grid", "pruned": "exact pruned"}.get(icp_search_used, "brute-force"), args.steps * HYPS_PER_STEP, regions),
                       "n_src": n, "n_tgt": n, "hyps_per_step": HYPS_PER_STEP, "parallelism": "instances sharded, %d rank(s)%s" % (world, " - REHEARSAL: all ranks on GPU 0, gloo" if rehearse else ""),
                       "icp_start": "0.3 deg / 0.5 mm from the ground truth (inside the basin of the reference's 0.4-voxel threshold; SURVEY 8d's 3 deg / 5 mm start "
                                    "lies outside it and every iteration count is fixed, so the start only decides how many correspondences are accepted)",
                       "model_bcast_ms": bcast_ms if distributed else None,
                       "ransac_hyps_per_s": hyps_per_s, "icp_iters_per_s": iters_per_s},
            "model_bcast_ms": bcast_ms if distributed else None,
            "per_rank": [{"rank": int(x[0]), "steps_per_s": x[1], "dominant_kernel_avg_ms": x[2], "dominant_kernel_frac": x[3], "model_bcast_ms": x[4] if distributed else None}
                         for x in per_rank],
            "roofline": roofline,
            "result_check": {"icp_fitness": float(r_icp.fitness), "icp_rmse": float(r_icp.rmse), "ransac_inliers": int(r_rs.inliers),
                             "ransac_iterations_run": int(r_rs.iterations_run)},
        }
        if c5_error is not None:
            out["config"]["c5_error"] = c5_error[:120]
            out["c5_error"] = c5_error
        if c5_out is not None:
            out["c5"] = c5_out
            out["config"].update({"c5_instances_per_s": c5_out["instances_per_s"], "c5_instances": c5_out["instances"], "c5_model_bcast_ms": c5_out["model_bcast_ms"],
                                  "c5_gather_ms": c5_out["gather_ms"], "c5_registered_share": c5_out["registered_share"], "c5_collectives": c5_out["collectives"]})
        if not args.no_operators and world == 1:
            # every other stage of the path, measured after the timed region (outside `value`): median of 3 repetitions each
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import opbench
            t_ops = time.perf_counter()
            out["operators"] = opbench.measure_all(ctx, tdv, synth, torch, dev)
            out["operators_wall_s"] = time.perf_counter() - t_ops
            # C4 / C5 flat, in a dict the driver keeps
            for row in out["operators"] if isinstance(out["operators"], list) else []:
                if not isinstance(row, dict): continue
                if row.get("op") == "register_batch_c5": out["config"]["c5_one_rank_instances_per_s"] = row.get("instances_per_s")
                if row.get("op", "").startswith("register_batch") and "instances_per_s" in row and "c4" in json.dumps(row.get("workload", "")).lower():
                    out["config"].setdefault("c4_instances_per_s", row.get("instances_per_s"))
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is a single-GPU-run item (rank 0, N = 1)
            from oracle import pyoracle as orc
            out["cpu_baseline"] = cpu_baseline(orc, src_np, model[:, :3].cpu().numpy(), model[:, 3:6].cpu().numpy(), corr_np, T0, icp_thr, voxel, args.cpu_budget_s)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        print(json.dumps(out))
        sys.stdout.flush()
    if c5_hung:
        os._exit(0)        # a thread of this process is still inside a collective: no orderly shutdown is possible, and the line is out
    ctx.close()
    if distributed:
        if c5_error is not None:
            os._exit(0)    # (a communicator of this process may be in an undefined state: leave without the orderly shutdown; the line is out)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
