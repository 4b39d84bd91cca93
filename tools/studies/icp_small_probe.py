#!/usr/bin/env python3
"""Per-iteration latency of the one-launch ICP (k_icp_small) on a C5-sized problem: slope of the call time over fixed iteration counts."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
for ns, nt in ((400, 398), (1000, 1000), (2000, 2000)):
    tgt, nrm = synth.sample_object(nt, 42)
    src, T_gt = synth.make_scene(ns, 42)
    T0 = synth.perturb(T_gt, angle_deg=0.3, trans=0.0005)
    d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_n = torch.from_numpy(nrm).to(dev)
    thr = 0.4 * float(synth.mean_spacing(nt))
    res = {}
    for iters in (10, 110):
        f = lambda: ctx.icp_dev(d_s.data_ptr(), ns, d_t.data_ptr(), d_n.data_ptr(), nt, T0, thr, iters, True, fixed_iterations=True)
        for _ in range(3): f()
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res[iters] = float(np.median(ts)) * 1e6
    print("%d x %d: %.1f us per iteration (call of 10: %.0f us, of 110: %.0f us), search %s" % (ns, nt, (res[110] - res[10]) / 100.0, res[10], res[110], ctx.last_icp_search()))
