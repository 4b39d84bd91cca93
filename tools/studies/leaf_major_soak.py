#!/usr/bin/env python3
"""Soak: the leaf-major descriptor search against the plain scan on the same device, many random clustered inputs of many sizes
(ns 4,096-90,000, nt 2,048-60,000, 8-2,000 clusters, noise 1e-4..0.1, duplicates, non-finite rows, far-away sources).  Prints the
number of calls and of mismatching calls (0 expected: both are exact searches with the reference's tie rule)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
tdv = importlib.import_module("3dvision_amd")
ctx = tdv.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
calls = bad = 0
t0 = time.time()
while calls < (int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    ns, nt = int(rng.integers(4096, 90000)), int(rng.integers(2048, 60000))
    ncl = int(rng.integers(8, 2000)); noise = 10.0 ** -float(rng.uniform(1, 4))
    centres = rng.random((ncl, 33)).astype(np.float32)
    ft = centres[rng.integers(0, ncl, nt)] + rng.normal(0, noise, (nt, 33)).astype(np.float32)
    fs = centres[rng.integers(0, ncl, ns)] + rng.normal(0, noise, (ns, 33)).astype(np.float32)
    k = int(rng.integers(0, 400))
    if k:
        ft[nt // 2: nt // 2 + k] = ft[:k]; fs[:min(k, ns)] = ft[nt // 2: nt // 2 + min(k, ns)]
    for _ in range(int(rng.integers(0, 6))):
        fs[int(rng.integers(0, ns))] = float(rng.choice([50.0, -9.0, np.nan, np.inf]))
        ft[int(rng.integers(0, nt)), int(rng.integers(0, 33))] = float(rng.choice([np.nan, np.inf, -np.inf]))
    os.environ.pop("TDV_FM_BRUTE", None)
    a = ctx.feature_match(fs, ft)
    os.environ["TDV_FM_BRUTE"] = "1"
    b = ctx.feature_match(fs, ft)
    os.environ.pop("TDV_FM_BRUTE", None)
    calls += 1
    if not np.array_equal(a, b):
        bad += 1
        print("MISMATCH", calls, ns, nt, ncl, noise, int((a != b).sum()))
print("calls %d, mismatching calls %d, %.0f s" % (calls, bad, time.time() - t0))
sys.exit(1 if bad else 0)
