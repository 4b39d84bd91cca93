#!/usr/bin/env python3
"""Stress run of the descriptor match (GPU box): many structured / unstructured / degenerate descriptor sets at sizes
that take the packed-index search, pass B and the fallback scan; every result is compared with the plain scan
(TDV_FM_BRUTE=1), which tests/ pin against the oracle.  Not part of the test-suite (minutes of GPU time)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tdv = importlib.import_module("3dvision_amd")
ctx = tdv.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def manifold(n, dim, noise, scale, offset):
    """points near a random smooth dim-D manifold of R^33"""
    t = rng.random((n, dim))
    A = rng.normal(size=(dim, 33)); B = rng.normal(size=(dim, 33)) * 0.3
    x = t @ A + np.sin(3 * t) @ B + rng.normal(0, noise, (n, 33))
    return ((x + offset) * scale).astype(np.float32)


bad = 0
t0 = time.time()
for trial in range(trials):
    hi = int(os.environ.get("FM_STRESS_MAX", "60000")); ns = int(rng.integers(4096, hi)); nt = int(rng.integers(2048, hi))
    kind = trial % 6
    if kind == 0:
        ft = manifold(nt, 2, 1e-3, 1.0, 0.0); fs = ft[rng.integers(0, nt, ns)] + rng.normal(0, 2e-3, (ns, 33)).astype(np.float32)
    elif kind == 1:
        ft = manifold(nt, 3, 1e-2, 0.01, 5.0); fs = manifold(ns, 3, 1e-2, 0.01, 5.0)
    elif kind == 2:                                   # plateau: most rows identical up to the last bits
        base = rng.random(33).astype(np.float32)
        ft = np.repeat(base[None], nt, 0) + (rng.random((nt, 33)) < 0.01) * np.float32(1e-7)
        fs = np.repeat(base[None], ns, 0); fs[::7] += rng.normal(0, 0.1, (len(fs[::7]), 33)).astype(np.float32)
        ft = ft.astype(np.float32); fs = fs.astype(np.float32)
    elif kind == 3:                                   # no structure at all
        ft = rng.random((nt, 33)).astype(np.float32); fs = rng.random((ns, 33)).astype(np.float32)
    elif kind == 4:                                   # histogram-like rows with outliers and non-finite values
        ft = manifold(nt, 3, 5e-3, 1.0, 2.0); ft = np.abs(ft); ft /= ft.sum(1, keepdims=True)
        fs = ft[rng.integers(0, nt, ns)] * (1 + rng.normal(0, 0.02, (ns, 33))).astype(np.float32)
        fs[:200] = rng.random((200, 33)); fs[300, 4] = np.nan; ft[17, 2] = np.inf; ft[18] = np.nan
        ft = ft.astype(np.float32); fs = fs.astype(np.float32)
    else:                                             # duplicates everywhere: the lowest index must win
        u = manifold(max(64, nt // 8), 2, 1e-3, 1.0, 0.0); ft = u[rng.integers(0, len(u), nt)]; fs = u[rng.integers(0, len(u), ns)]
    os.environ.pop("TDV_FM_BRUTE", None)
    got = ctx.feature_match(fs, ft)
    os.environ["TDV_FM_BRUTE"] = "1"
    ref = ctx.feature_match(fs, ft)
    os.environ.pop("TDV_FM_BRUTE", None)
    diff = int((got != ref).sum())
    bad += diff > 0
    print("trial %2d kind %d  %6d x %6d  %s" % (trial, kind, ns, nt, "ok" if diff == 0 else "MISMATCH in %d rows" % diff), flush=True)
print("%d trials, %d mismatching, %.0f s" % (trials, bad, time.time() - t0))
sys.exit(1 if bad else 0)
