"""Soak of round 4's new kernels against numpy on random shapes: the one-launch depth scan (k_depth_cloud_chain) and the radix sort
(k_rs_count / k_rs_scatter).  Usage: round4_soak.py [iterations] - prints one line per 20 iterations, exits 1 on the first mismatch."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
tdv = importlib.import_module("3dvision_amd")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)


def cloud_ref(raw, mask, scale, fx, fy, cx, cy, zmax):
    z = raw.astype(np.float32) * np.float32(1.0 / scale)
    z[mask <= 10] = 0
    keep = ~((z <= 0) | (z > np.float32(zmax)))
    v, u = np.nonzero(keep); zz = z[keep]
    return np.stack([(u.astype(np.float32) - np.float32(cx)) * zz / np.float32(fx), (v.astype(np.float32) - np.float32(cy)) * zz / np.float32(fy), zz], axis=1)


for it in range(iters):
    h = int(rng.integers(1, 2200)); w = int(rng.integers(1, 4000))
    if h * w > 6_000_000: h = 6_000_000 // w
    p0 = float(rng.uniform(0, 1))
    raw = rng.integers(0, 4000, (h, w)).astype(np.uint16); raw[rng.random((h, w)) < p0] = 0
    mask = rng.choice(np.array([0, 10, 11, 255], np.uint8), (h, w), p=[0.3, 0.1, 0.3, 0.3])
    ref = cloud_ref(raw, mask, 1000.0, 700.0, 710.0, w * 0.4, h * 0.6, 3.0)
    d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_mask = torch.from_numpy(mask).to(dev)
    d_xyz = torch.empty((h * w, 3), dtype=torch.float32, device=dev)
    n = ctx.depth_to_cloud_dev(d_raw.data_ptr(), d_mask.data_ptr(), None, w, h, 1000.0, 700.0, 710.0, w * 0.4, h * 0.6, 3.0, d_xyz.data_ptr(), None, h * w)
    if n != len(ref) or d_xyz[:n].cpu().numpy().tobytes() != ref.tobytes():
        print("DEPTH MISMATCH at", it, h, w, n, len(ref)); sys.exit(1)
    m = int(rng.integers(1, 700000)) if it % 5 else int(rng.integers(1, 5000))
    end_bit = int(rng.integers(1, 65))
    keys = rng.integers(0, 1 << 63, m, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, m, dtype=np.uint64)
    if it % 3 == 0: keys &= np.uint64(0xFF00FF)                       # few distinct digits
    vals = rng.integers(0, 1 << 32, m, dtype=np.uint64).astype(np.uint32)
    d_k = torch.from_numpy(keys.view(np.int64)).to(dev); d_v = torch.from_numpy(vals.view(np.int32)).to(dev)
    o_k = torch.empty_like(d_k); o_v = torch.empty_like(d_v)
    ctx.radix_sort_pairs_dev(d_k.data_ptr(), o_k.data_ptr(), d_v.data_ptr(), o_v.data_ptr(), m, end_bit)
    low = keys & np.uint64((1 << end_bit) - 1) if end_bit < 64 else keys
    order = np.argsort(low, kind="stable")
    if o_k.cpu().numpy().view(np.uint64).tobytes() != keys[order].tobytes() or o_v.cpu().numpy().view(np.uint32).tobytes() != vals[order].tobytes():
        print("SORT MISMATCH at", it, m, end_bit); sys.exit(1)
    if it % 20 == 19: print("ok", it + 1, flush=True)
print("soak passed:", iters, "iterations")
