#!/usr/bin/env python3
"""Descriptor match on the C4 workload's real FPFH descriptors (one instance vs the model), timed per path:
packed-index search (leaf-major, the default; "walk" = round 2's k_fm_query; index build and query separately), round 1's
key-ordered pruned scan, the plain scan.
    python tools/bench_fm.py [--voxel-px 1.2] [--reps 5] [--stats]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_batch as bb  # noqa: E402


def descriptors(tdv, synth, ctx, voxel_px, dev):
    wl = bb.build_workload(tdv, synth, ctx, 1, voxel_px, 448, 3, tdv.TDV_VOXEL_ORDER_FIRST, dev)
    d_mx, d_mn, d_mf, nm = wl["model"]
    n_px = wl["mask_px"][0]
    d_xyz = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
    n = ctx.depth_to_cloud_dev(wl["depth"][0].data_ptr(), wl["masks"][0].data_ptr(), None, bb.W, bb.H, bb.SCALE, bb.F, bb.F, bb.CX, bb.CY, bb.ZMAX,
                               d_xyz.data_ptr(), None, n_px)
    d_v = torch.empty_like(d_xyz)
    v = ctx.voxel_downsample_dev(d_xyz.data_ptr(), None, n, wl["voxel"], d_v.data_ptr(), None, n)
    d_n = torch.empty((v, 3), dtype=torch.float32, device=dev); d_f = torch.empty((v, 33), dtype=torch.float32, device=dev)
    ctx.estimate_normals_dev(d_v.data_ptr(), v, 30, d_n.data_ptr())
    ctx.compute_fpfh_dev(d_v.data_ptr(), d_n.data_ptr(), v, wl["voxel"] * 5.0, d_f.data_ptr())
    return d_f, v, d_mf[:nm].contiguous(), nm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxel-px", type=float, default=1.2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--paths", default="index,walk,keyorder,brute")
    args = ap.parse_args()
    tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
    dev = torch.device("cuda", 0)
    ctx = tdv.Context(0)
    d_fs, ns, d_ft, nt = descriptors(tdv, synth, ctx, args.voxel_px, dev)
    d_corr = torch.empty(ns, dtype=torch.int32, device=dev)
    out = dict(ns=ns, nt=nt)
    ref = None
    for path in args.paths.split(","):
        for k in ("TDV_FM_KEYORDER", "TDV_FM_BRUTE", "TDV_FM_STATS", "TDV_FM_LEAFMAJOR"):
            os.environ.pop(k, None)
        if path == "keyorder": os.environ["TDV_FM_KEYORDER"] = "1"
        if path == "brute": os.environ["TDV_FM_BRUTE"] = "1"
        if path == "walk": os.environ["TDV_FM_LEAFMAJOR"] = "0"     # round 2's search (k_fm_query): a wave walks its two sources' leaves
        ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_corr.data_ptr())   # warm-up
        if path in ("index", "walk") and args.stats:
            os.environ["TDV_FM_STATS"] = "1"
            ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_corr.data_ptr())
            os.environ.pop("TDV_FM_STATS")
        ctx.timing_enable(True)
        ctx.timing_read(tdv.TIMER_FEATURE_MATCH); ctx.timing_read(7)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps):
            ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_corr.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.reps
        q = ctx.timing_read(tdv.TIMER_FEATURE_MATCH)[0] / args.reps; b = ctx.timing_read(7)[0] / args.reps
        ctx.timing_enable(False)
        c = d_corr.cpu().numpy()
        if ref is None: ref = c
        out[path] = dict(wall_ms=dt * 1e3, query_ms=q, index_build_ms=b, identical_to_first=bool(np.array_equal(c, ref)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
