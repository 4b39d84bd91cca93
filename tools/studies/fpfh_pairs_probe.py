#!/usr/bin/env python3
"""Study: compute_fpfh with two points per wave (k_fpfh_pairs, the default) against one point per wave (TDV_FPFH_PAIRS=0):
time of the whole operator and equality of the descriptors, bit for bit."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
for n in (100000, 147001, 200000):
    pts, nrm = synth.sample_object(n, 42)
    d_x = torch.from_numpy(pts).to(dev); d_n = torch.from_numpy(nrm).to(dev)
    d_f = torch.empty((n, 33), dtype=torch.float32, device=dev)
    r = 5.0 * float(np.float32(synth.mean_spacing(n)))
    out = {}
    for mode in ("1", "0"):
        os.environ["TDV_FPFH_PAIRS"] = mode
        for _ in range(2): ctx.compute_fpfh_dev(d_x.data_ptr(), d_n.data_ptr(), n, r, d_f.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.compute_fpfh_dev(d_x.data_ptr(), d_n.data_ptr(), n, r, d_f.data_ptr())
        torch.cuda.synchronize(); out[mode] = (time.perf_counter() - t0) / 5 * 1e3
        out["f" + mode] = d_f.cpu().numpy().tobytes()
    print(n, "two points per wave %.3f ms, one %.3f ms, identical %s" % (out["1"], out["0"], out["f1"] == out["f0"]))
