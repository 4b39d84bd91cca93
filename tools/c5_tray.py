#!/usr/bin/env python3
"""Config C5's per-GPU workload (BASELINE.json configs[4]: "8x1024-instance batch ..., 500k-pt scene"): ONE 1280x720 depth
frame showing a tray of 1,024 small relief parts (3dvision_amd/synth.py: tray_scene), cut into instances by ONE uint16 label
image, registered against a scan of the part in one tdv_register_batch_dev call — depth -> cloud -> voxel (the reference's
container order) -> normals(30) -> FPFH -> descriptor match -> RANSAC -> ICP per instance, src/pipeline.cpp:25-150.
The scene cloud is ~530k points, ~520 per instance.

    python tools/c5_tray.py [--instances 1024] [--hyps 10000] [--icp-iters 50] [--order reference|first]

Prints one JSON line: instances/s, the share of instances within --max-angle of their ground truth (the reference's own
algorithm loses a few of these small parts: tests/test_gpu_c5.py holds sampled instances against the oracle's chain), the
workspace high-water mark.  Shared by tests/test_gpu_c5.py, tools/bench_c5.py and tools/opbench.py."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRAY = dict(feature_voxels=3.0)      # part geometry of the C5 tray (see synth.tray_scene for the rest)
MAX_ANGLE = 3e-2                     # rad: a 25-pixel part at 0.45 m; 3e-2 rad moves its rim by 0.3 pixel footprints


def build(tdv, synth, ctx, n_instances, dev, seed=7, order=None, hyps=10000, icp_iters=50, pose_seed=None):
    """The tray on the device: dict(sc = the numpy scene, depth / label device tensors, model = (xyz, normals, fpfh, n), params)."""
    import torch
    order = tdv.TDV_VOXEL_ORDER_REFERENCE if order is None else order
    sc = synth.tray_scene(n_instances, seed=seed, pose_seed=pose_seed, **TRAY)
    W, H = sc["width"], sc["height"]
    d_depth = torch.from_numpy(sc["depth"].view(np.int16)).to(dev)
    d_label = torch.from_numpy(sc["label"].view(np.int16)).to(dev)
    md = torch.from_numpy(sc["model_depth"].view(np.int16)).to(dev); mm = torch.from_numpy(sc["model_mask"]).to(dev)
    n_px = int((sc["model_mask"] > 0).sum())
    d_mraw = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
    n_raw = ctx.depth_to_cloud_dev(md.data_ptr(), mm.data_ptr(), None, W, H, sc["scale"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["zmax"],
                                   d_mraw.data_ptr(), None, n_px)
    d_mx = torch.empty_like(d_mraw); d_mn = torch.empty_like(d_mraw); d_mf = torch.empty((n_raw, 33), dtype=torch.float32, device=dev)
    nm = ctx.prepare_model_dev(d_mraw.data_ptr(), n_raw, sc["voxel"], 30, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), order=order)
    prm = tdv.batch_params(width=W, height=H, scale_to_meters=sc["scale"], fx=sc["fx"], fy=sc["fy"], cx=sc["cx"], cy=sc["cy"], zmax=sc["zmax"],
                           voxel_size=sc["voxel"], ransac_max_iterations=hyps, icp_max_iterations=icp_iters, voxel_order=order, mask_format=2)
    return dict(sc=sc, depth=d_depth, label=d_label, model=(d_mx, d_mn, d_mf, nm), params=prm, n_instances=n_instances)


def run(ctx, wl):
    d_mx, d_mn, d_mf, nm = wl["model"]
    return ctx.register_batch_dev(wl["depth"].data_ptr(), None, wl["label"].data_ptr(), wl["n_instances"], wl["params"],
                                  d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)


def angles(synth, wl, res):
    return np.array([synth.pose_error(r["T"], wl["sc"]["T_gt"][b])[0] if r["status"] == 0 else np.inf for b, r in enumerate(res)])


def measure(tdv, synth, ctx, torch, dev, n_instances=1024, order=None, hyps=10000, icp_iters=50, reps=1):
    wl = build(tdv, synth, ctx, n_instances, dev, order=order, hyps=hyps, icp_iters=icp_iters)
    run(ctx, wl)                                                       # warm-up: arena growth, lanes, code load
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = run(ctx, wl)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    ang = angles(synth, wl, res)
    nv = np.array([r["n_voxels"] for r in res]); npnt = np.array([r["n_points"] for r in res])
    return dict(op="register_batch_c5",
                workload="C5, one rank's share: %d instances cut by ONE uint16 label image from ONE %dx%d frame (scene cloud %d points, %.0f per instance, "
                         "%.0f voxels) vs a %d-point model, %d hypotheses + ICP each, %s voxel order"
                         % (n_instances, wl["sc"]["width"], wl["sc"]["height"], int(npnt.sum()), npnt.mean(), nv.mean(), wl["model"][3], hyps,
                            "reference" if wl["params"].voxel_order == tdv.TDV_VOXEL_ORDER_REFERENCE else "first-occurrence"),
                ms=dt * 1e3, instances_per_s=n_instances / dt, us_per_instance=dt / n_instances * 1e6,
                scene_points=int(npnt.sum()), status_ok=int(sum(r["status"] == 0 for r in res)),
                registered_share=float((ang <= MAX_ANGLE).mean()), max_angle_rad=MAX_ANGLE, median_angle_to_ground_truth_rad=float(np.median(ang)),
                icp_iterations_per_instance=float(np.mean([r["icp_iterations"] for r in res])),
                ransac_rescore_share=ctx.last_ransac_rescore(),
                workspace_high_water_MiB=ctx.workspace_high_water() / 2 ** 20 if hasattr(ctx, "workspace_high_water") else None,
                bound="chain of small per-instance launches (launch- and host-bound at ~500 points per instance)"), wl, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=1024)
    ap.add_argument("--hyps", type=int, default=10000)
    ap.add_argument("--icp-iters", type=int, default=50)
    ap.add_argument("--order", choices=["first", "reference"], default="reference")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    tdv = importlib.import_module("3dvision_amd")
    synth = importlib.import_module("3dvision_amd.synth")
    dev = torch.device("cuda", 0)
    ctx = tdv.Context(0)
    order = tdv.TDV_VOXEL_ORDER_REFERENCE if args.order == "reference" else tdv.TDV_VOXEL_ORDER_FIRST
    out, _, _ = measure(tdv, synth, ctx, torch, dev, args.instances, order, args.hyps, args.icp_iters, args.reps)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
