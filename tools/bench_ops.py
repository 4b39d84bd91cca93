#!/usr/bin/env python3
"""Per-operator timings of the hot path on one MI355X, inputs resident in HBM (the *_dev ABI).
Fills the tables in DESIGN.md / BASELINE.md; bench.py stays the headline contract.

    python tools/bench_ops.py [--quick]      -> one JSON line per measurement
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--big", action="store_true", help="add the 500k-point size of BASELINE config C5 (normals, FPFH, ICP)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    tdv = importlib.import_module("3dvision_amd")
    synth = importlib.import_module("3dvision_amd.synth")
    dev = torch.device("cuda", 0)
    ctx = tdv.Context(0)
    ctx.timing_enable(True)

    def emit(**kw):
        print(json.dumps(kw), flush=True)

    def timed(fn, reps=3, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    def cloud(n, seed):
        pts, nrm = synth.sample_object(n, seed)
        T = synth.gt_transform(seed)
        Tinv = np.linalg.inv(T.astype(np.float64))
        cam = (pts.astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
        return cam, pts, nrm

    want = lambda name: (not args.only) or (args.only in name)

    # ---------------- R1+R2: depth -> cloud, 1280x720 frame with a 448x448 mask (C4's instance shape)
    if want("depth"):
        h, w = 720, 1280
        rng = np.random.default_rng(0)
        raw = (800 + rng.integers(0, 200, (h, w))).astype(np.uint16)
        mask = np.zeros((h, w), np.uint8); mask[136:584, 416:864] = 255
        bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_mask = torch.from_numpy(mask).to(dev); d_bgr = torch.from_numpy(bgr).to(dev)
        d_xyz = torch.empty((h * w, 3), dtype=torch.float32, device=dev); d_rgb = torch.empty_like(d_xyz)
        cnt = [0]

        def f():
            cnt[0] = ctx.depth_to_cloud_dev(d_raw.data_ptr(), d_mask.data_ptr(), d_bgr.data_ptr(), w, h, 1000.0, 900, 900, 640, 360, 1.5,
                                            d_xyz.data_ptr(), d_rgb.data_ptr(), h * w)
        ctx.timing_read(tdv.TIMER_DEPTH)
        t = timed(f, reps=20, warm=3)
        ms, launches = ctx.timing_read(tdv.TIMER_DEPTH)
        nbytes = 2 * (2 + 1) * h * w + 3 * cnt[0] + 24 * cnt[0]  # two passes over depth+mask, bgr of kept pixels, xyz+rgb out
        emit(op="depth_to_cloud", frame="1280x720", points=cnt[0], wall_ms=t * 1e3, kernels_ms=ms / max(launches, 1),
             algorithmic_bytes=nbytes, hbm_GBps=nbytes / (ms / max(launches, 1) * 1e-3) / 1e9)

    # ---------------- R1+R2 at C4 scale: 256 stacked masks (448x448 px each) of one frame, one count + one emit launch
    if want("depthbatch"):
        h, w, B = 720, 1280, 256   # config C4's instance count
        rng = np.random.default_rng(0)
        raw = (800 + rng.integers(0, 200, (h, w))).astype(np.uint16)
        masks = np.zeros((B, h, w), np.uint8)
        for b in range(B):
            y0 = (b * 37) % (h - 448); x0 = (b * 101) % (w - 448)
            masks[b, y0:y0 + 448, x0:x0 + 448] = 255
        d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_masks = torch.from_numpy(masks).to(dev)
        cap = B * 448 * 448
        d_xyz = torch.empty((cap, 3), dtype=torch.float32, device=dev)
        off = [None]

        def f():
            off[0] = ctx.depth_to_cloud_batch_dev(d_raw.data_ptr(), d_masks.data_ptr(), None, B, w, h, 1000.0, 900, 900, 640, 360, 1.5, d_xyz.data_ptr(), None, cap)
        ctx.timing_read(tdv.TIMER_DEPTH)
        t = timed(f, reps=10, warm=2)
        ms, launches = ctx.timing_read(tdv.TIMER_DEPTH)
        per_call_ms = ms / 12   # 2 warm-up + 10 timed calls were recorded
        npts = int(off[0][-1])
        # two passes read depth (L2/MALL-resident after the first instance) + mask; emit writes 12 B per point
        nbytes = 2 * B * h * w * 1 + 2 * h * w * 2 + 12 * npts
        emit(op="depth_to_cloud_batch", instances=B, frame="1280x720", points=npts, wall_ms=t * 1e3, kernels_ms=per_call_ms,
             algorithmic_bytes=nbytes, hbm_GBps=nbytes / (per_call_ms * 1e-3) / 1e9,
             bytes_incl_depth_rereads=2 * B * h * w * 3 + 12 * npts)

    sizes = [50000] if args.quick else [50000, 100000, 200000] + ([500000] if args.big else [])
    for n in sizes:
        cam, mdl, mnrm = cloud(n, 42)
        d_xyz = torch.from_numpy(cam).to(dev)
        # ---------------- R3 voxel
        if want("voxel"):
            voxel = float(synth.mean_spacing(n)) * 1.5
            d_out = torch.empty_like(d_xyz)
            m = [0]

            def f():
                m[0] = ctx.voxel_downsample_dev(d_xyz.data_ptr(), None, n, voxel, d_out.data_ptr(), None, n)
            ctx.timing_read(tdv.TIMER_VOXEL)
            t = timed(f, reps=5)
            ms, l = ctx.timing_read(tdv.TIMER_VOXEL)
            emit(op="voxel_downsample", n=n, voxels=m[0], wall_ms=t * 1e3, algorithmic_bytes=24 * n + 24 * m[0])
        # ---------------- R4a normals
        d_nrm = torch.empty_like(d_xyz)
        if want("normals") or want("fpfh") or want("match"):
            def f():
                ctx.estimate_normals_dev(d_xyz.data_ptr(), n, 30, d_nrm.data_ptr())
            ctx.timing_read(tdv.TIMER_KNN)
            t = timed(f, reps=2)
            ms, l = ctx.timing_read(tdv.TIMER_KNN)
            emit(op="estimate_normals_k30", n=n, wall_ms=t * 1e3, scan_kernel_ms=ms / max(l, 1), pairs=float(n) * n,
                 scan_Tops=9.0 * n * n / (ms / max(l, 1) * 1e-3) / 1e12)
        # ---------------- R4b FPFH
        d_desc = torch.empty((n, 33), dtype=torch.float32, device=dev)
        if want("fpfh") or want("match"):
            radius = float(synth.mean_spacing(n)) * 5.0
            cntt = torch.empty(n, dtype=torch.int32, device=dev)

            def f():
                ctx.compute_fpfh_dev(d_xyz.data_ptr(), d_nrm.data_ptr(), n, radius, d_desc.data_ptr(), None, cntt.data_ptr())
            ctx.timing_read(tdv.TIMER_RADIUS)
            t = timed(f, reps=2)
            ms, l = ctx.timing_read(tdv.TIMER_RADIUS)
            emit(op="compute_fpfh", n=n, radius=radius, mean_neighbors=float(cntt.float().mean()), wall_ms=t * 1e3,
                 scan_kernel_ms=ms / max(l, 1), scan_Tops=9.0 * n * n / (ms / max(l, 1) * 1e-3) / 1e12)
        # ---------------- R5(i) feature match
        if want("match") and n <= 200000:
            # model descriptors: FPFH of the model cloud (real, clustered) and random rows (unstructured: nothing to prune)
            d_mx = torch.from_numpy(mdl).to(dev); d_mn = torch.empty_like(d_mx)
            d_mdesc = torch.empty((n, 33), dtype=torch.float32, device=dev)
            ctx.estimate_normals_dev(d_mx.data_ptr(), n, 30, d_mn.data_ptr())
            ctx.compute_fpfh_dev(d_mx.data_ptr(), d_mn.data_ptr(), n, float(synth.mean_spacing(n)) * 5.0, d_mdesc.data_ptr(), None, None)
            d_rand = torch.from_numpy(synth.random_features(n, 9)).to(dev)
            d_corr = torch.empty(n, dtype=torch.int32, device=dev)
            for label, d_ft in (("fpfh_of_model", d_mdesc), ("random_rows", d_rand)):
                def f():
                    ctx.feature_match_dev(d_desc.data_ptr(), n, d_ft.data_ptr(), n, d_corr.data_ptr())
                ctx.timing_read(tdv.TIMER_FEATURE_MATCH)
                t = timed(f, reps=2)
                ms, l = ctx.timing_read(tdv.TIMER_FEATURE_MATCH)
                emit(op="feature_match", targets=label, ns=n, nt=n, wall_ms=t * 1e3, scan_kernel_ms=ms / max(l, 1),
                     bruteforce_equivalent_Tops=98.0 * n * n / (ms / max(l, 1) * 1e-3) / 1e12, algorithmic_bytes=132 * 2 * n + 4 * n)
    # ---------------- C2: ICP 50k x 10k
    if want("icp"):
        for (ns, nt) in [(50000, 10000)] + ([] if args.quick else [(200000, 200000)]) + ([(500000, 500000)] if args.big else []):
            tgt, nrm = synth.sample_object(nt, 42)
            src, T_gt = synth.make_scene(ns, 42)
            T0 = synth.perturb(T_gt)
            d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_n = torch.from_numpy(nrm).to(dev)
            thr = float(synth.mean_spacing(nt)) * 4
            iters = 50
            for search in (("pruned",) if ns > 200000 else ("brute", "pruned")):
              ctx.set_icp_search(search)
              for mode in (True, False):
                def f():
                    return ctx.icp_dev(d_s.data_ptr(), ns, d_t.data_ptr(), d_n.data_ptr(), nt, T0, thr, iters, mode, fixed_iterations=True)
                ctx.timing_read(tdv.TIMER_ICP_NN)
                t = timed(f, reps=2)
                ms, l = ctx.timing_read(tdv.TIMER_ICP_NN)
                r = f()
                emit(op="icp_fixed50", search=search, ns=ns, nt=nt, point_to_plane=mode, wall_ms=t * 1e3, iters_per_s=iters / t, nn_kernel_ms=ms / max(l, 1),
                     nn_Tops=8.0 * ns * nt / (ms / max(l, 1) * 1e-3) / 1e12, fitness=float(r.fitness),
                     ang_to_gt=synth.rotation_angle(T_gt[:3, :3], r.transformation[:3, :3]))
            ctx.set_icp_search("auto")
    # ---------------- C3: RANSAC 100k, 50k hypotheses
    if want("ransac"):
        for n, hyps in [(100000, 50000)] + ([] if args.quick else [(200000, 100000)]):
            tgt, _ = synth.sample_object(n, 42)
            src, T_gt = synth.make_scene(n, 42)
            nn = ctx.icp_correspondences(src, tgt, T_gt, 1.0)["corr"]
            rng = np.random.Generator(np.random.PCG64(5))
            corr = np.where(rng.random(n) < 0.5, nn, rng.integers(0, n, n)).astype(np.int32)
            d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_c = torch.from_numpy(corr).to(dev)
            voxel = float(synth.mean_spacing(n))

            def f():
                return ctx.ransac_dev(d_s.data_ptr(), n, d_t.data_ptr(), n, None, None, d_c.data_ptr(), voxel, hyps, 2.0, 42)
            ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
            t = timed(f, reps=2)
            ms, l = ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
            r = f()
            emit(op="ransac", n=n, hyps=hyps, wall_ms=t * 1e3, hyps_per_s=hyps / t, score_kernel_ms_total=ms / 2,
                 score_Tops=28.0 * n * hyps * 2 / (ms * 1e-3) / 1e12, inliers=int(r.inliers),
                 ang_to_gt=synth.rotation_angle(T_gt[:3, :3], r.transformation[:3, :3]))
    ctx.close()


if __name__ == "__main__":
    main()
