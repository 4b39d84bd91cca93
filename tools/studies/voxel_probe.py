#!/usr/bin/env python3
"""Wall time and (under rocprofv3) kernel times of tdv_voxel_downsample_dev at 100k / 200k points, both orders."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
for n in (100000, 200000):
    pts, _ = synth.sample_object(n, 42)
    voxel = float(np.float32(synth.mean_spacing(n))) * 1.5
    d = torch.from_numpy(pts).to(dev); o = torch.empty_like(d)
    for order, name in ((tdv.TDV_VOXEL_ORDER_FIRST, "first"), (tdv.TDV_VOXEL_ORDER_REFERENCE, "reference")):
        for _ in range(3): v = ctx.voxel_downsample_dev(d.data_ptr(), None, n, voxel, o.data_ptr(), None, n, order)
        ts = []
        for _ in range(20):
            torch.cuda.synchronize(); t = time.perf_counter()
            v = ctx.voxel_downsample_dev(d.data_ptr(), None, n, voxel, o.data_ptr(), None, n, order)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e6)
        print("n %d -> %d voxels, %s order: median %.1f us, min %.1f us" % (n, v, name, np.median(ts), min(ts)))
