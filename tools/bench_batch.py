#!/usr/bin/env python3
"""Config C4 (BASELINE.json configs[3]) on one MI355X: batched bin-picking through tdv_register_batch_dev —
B instances, each a ~200k-pixel mask of one 1280x720 depth frame, registered against one shared model:
depth->cloud -> voxel -> normals(k=30) -> FPFH -> feature match -> RANSAC -> ICP, all device-resident.
Prints one JSON line (instances/s, per-stage kernel times, aggregate hyps/s and ICP iters/s).

    python tools/bench_batch.py [--instances 16] [--model-points 10000] [--hyps 10000] [--icp-iters 50]
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


def render(synth, model, T, f, cx, cy, w, h):
    p = model.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3]
    u = np.round(p[:, 0] / p[:, 2] * f + cx).astype(int); v = np.round(p[:, 1] / p[:, 2] * f + cy).astype(int)
    ok = (u >= 0) & (u < w) & (v >= 0) & (v < h) & (p[:, 2] > 0)
    z = np.full((h, w), np.inf)
    np.minimum.at(z, (v[ok], u[ok]), p[ok, 2])
    return z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=16)
    ap.add_argument("--model-points", type=int, default=10000)
    ap.add_argument("--hyps", type=int, default=10000)
    ap.add_argument("--icp-iters", type=int, default=50)
    ap.add_argument("--voxel", type=float, default=0.0005)
    ap.add_argument("--threads", type=int, default=1,
                    help="host threads, one tdv_ctx (stream + workspace) each, sharing the instances — the reference's thread-pool shape")
    ap.add_argument("--matched", action="store_true",
                    help="scene and model voxelised at the same 1 mm (as pipeline.cpp does with one voxel_size): smaller clouds, "
                         "but the registration is well-posed, so angle_to_gt is meaningful")
    args = ap.parse_args()
    if args.matched:
        args.voxel = 0.001
    import torch
    tdv = importlib.import_module("3dvision_amd")
    synth = importlib.import_module("3dvision_amd.synth")
    dev = torch.device("cuda", 0)
    ctx = tdv.Context(0)
    w, h, f = 1280, 720, 1500.0
    cx, cy = w / 2.0, h / 2.0
    # one object filling ~448x448 px: dense surface samples rendered through a z-buffer, holes closed by oversampling
    dense, _ = synth.sample_object(3000000, 42)
    T = synth.make_transform([0.2, 1.0, 0.3], 35.0, (0.0, 0.0, 0.45))
    z = render(synth, dense, T, f, cx, cy, w, h)
    hit = np.isfinite(z)
    depth = np.zeros((h, w), np.uint16); depth[hit] = np.round(z[hit] * 1000.0).astype(np.uint16)
    mask = np.where(hit, 255, 0).astype(np.uint8)
    B = args.instances
    masks = np.repeat(mask[None], B, 0)
    model_raw, _ = synth.sample_object(600000 if args.matched else args.model_points * 3, 7)
    d_raw = torch.from_numpy(model_raw).to(dev)
    d_mx = torch.empty_like(d_raw); d_mn = torch.empty_like(d_raw); d_mf = torch.empty((len(model_raw), 33), dtype=torch.float32, device=dev)
    mvox = args.voxel if args.matched else float(synth.mean_spacing(args.model_points))
    nm = ctx.prepare_model_dev(d_raw.data_ptr(), len(model_raw), mvox, 30, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr())
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev); d_masks = torch.from_numpy(masks).to(dev)
    prm = tdv.batch_params(width=w, height=h, fx=f, fy=f, cx=cx, cy=cy, zmax=1.5, voxel_size=args.voxel,
                           ransac_max_iterations=args.hyps, ransac_confidence=2.0, icp_max_iterations=args.icp_iters, icp_distance_factor=4.0)
    import threading
    nthr = max(1, args.threads)
    ctxs = [ctx] + [tdv.Context(0) for _ in range(nthr - 1)]
    share = [B // nthr + (1 if t < B % nthr else 0) for t in range(nthr)]
    offs = [sum(share[:t]) for t in range(nthr)]
    for c in ctxs:   # warm-up: arena growth, code load
        c.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), 1, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    ctx.timing_enable(nthr == 1)
    for s in range(7):
        ctx.timing_read(s)
    torch.cuda.synchronize()
    results = [None] * nthr

    def work(t):
        if share[t] == 0:
            results[t] = []
            return
        mptr = d_masks.data_ptr() + offs[t] * h * w
        results[t] = ctxs[t].register_batch_dev(d_depth.data_ptr(), None, mptr, share[t], prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)

    t0 = time.perf_counter()
    if nthr == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
        for x in th: x.start()
        for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = [r for part in results for r in part]
    names = ["icp_nn", "ransac_score", "feature_match", "knn_scan", "radius_scan", "depth", "voxel"]
    stage = {n: ctx.timing_read(i)[0] / B for i, n in enumerate(names)}
    Tinv = np.linalg.inv(T.astype(np.float64))
    ang = [synth.rotation_angle(Tinv[:3, :3], r["T"][:3, :3]) for r in res]
    print(json.dumps(dict(config=("matched-resolution batch (1 mm): " if args.matched else "C4-style batch: ") + "%d instances x %d-px mask of one 1280x720 frame vs %d-pt model" % (B, int(hit.sum()), nm),
                          instances=B, host_threads=nthr, pixels_per_instance=int(hit.sum()), voxels_per_instance=res[0]["n_voxels"], model_points=nm,
                          wall_s=dt, instances_per_s=B / dt, ms_per_instance=dt / B * 1e3,
                          ransac_hyps_per_s=B * args.hyps / dt, icp_iters_per_s=sum(r["icp_iterations"] for r in res) / dt,
                          kernel_ms_per_instance=stage, icp_fitness=[float(r["fitness"]) for r in res[:3]],
                          coarse_inliers=[r["coarse_inliers"] for r in res[:3]], angle_to_gt_rad=ang[:3])))
    ctx.close()


if __name__ == "__main__":
    main()
