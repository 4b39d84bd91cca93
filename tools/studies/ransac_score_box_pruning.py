"""Offline study of an exact box-pruned RANSAC scoring (pairs ordered along a 6-D Morton curve over (p, q), leaves of
16-64 pairs, interval bound of R*p + t against the q box).  Result: 10-28 % of leaves pass for random-triple hypotheses
(random pairs fill 6-D too sparsely), and the per-hypothesis L2 traffic would be ~7x the brute-force kernel's, so the
idea was NOT built: k_ransac_score stays a brute-force VALU-roofline kernel.  CPU only (needs scipy).
"""
import sys, importlib
sys.path.insert(0, '/root/repo')
import numpy as np
synth = importlib.import_module('3dvision_amd.synth')
n = 200000
tgt, _ = synth.sample_object(n, 42)
src, T_gt = synth.make_scene(n, 42)
vox = float(synth.mean_spacing(n)); thr = 1.5 * vox
# correspondences like bench.py: half true NN under T_gt (approx: use KD-tree), half random
from scipy.spatial import cKDTree
tree = cKDTree(tgt)
p_t = src @ T_gt[:3, :3].T + T_gt[:3, 3]
nn = tree.query(p_t)[1]
rng = np.random.default_rng(1234)
corr = np.where(rng.random(n) < 0.5, nn, rng.integers(0, n, n))
P = src.astype(np.float64); Q = tgt[corr].astype(np.float64)

def morton6(P, Q, bits):
    X = np.concatenate([P, Q], 1)
    lo, hi = X.min(0), X.max(0)
    q = np.clip(((X - lo) / (hi - lo + 1e-12) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(len(X), np.int64)
    for b in range(bits):
        for d in range(6):
            key |= ((q[:, d] >> b) & 1) << (6 * b + d)
    return key

def kabsch(ps, qs):
    cp, cq = ps.mean(0), qs.mean(0)
    H = (ps - cp).T @ (qs - cq)
    U, S, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[2] *= -1; R = Vt.T @ U.T
    return R, cq - R @ cp

for bits, LEAF in [(3, 64), (4, 64), (4, 32), (5, 16)]:
    order = np.argsort(morton6(P, Q, bits), kind='stable')
    Ps, Qs = P[order], Q[order]
    nl = n // LEAF
    Pl = Ps[:nl * LEAF].reshape(nl, LEAF, 3); Ql = Qs[:nl * LEAF].reshape(nl, LEAF, 3)
    Pmin, Pmax, Qmin, Qmax = Pl.min(1), Pl.max(1), Ql.min(1), Ql.max(1)
    fr = []; inl = []
    for h in range(40):
        tri = rng.integers(0, n, 3)
        R, t = kabsch(P[tri], Q[tri])
        # interval of R p + t over P box
        c = (Pmin + Pmax) / 2; e = (Pmax - Pmin) / 2
        xc = c @ R.T + t; xe = e @ np.abs(R).T
        gap = np.maximum(0, np.maximum((xc - xe) - Qmax, Qmin - (xc + xe)))
        lb = (gap ** 2).sum(1)
        fr.append((lb < thr * thr).mean())
        d = np.linalg.norm(P @ R.T + t - Q, axis=1); inl.append((d < thr).sum())
    # the true pose
    R, t = T_gt[:3, :3].astype(np.float64), T_gt[:3, 3].astype(np.float64)
    c = (Pmin + Pmax) / 2; e = (Pmax - Pmin) / 2
    xc = c @ R.T + t; xe = e @ np.abs(R).T
    gap = np.maximum(0, np.maximum((xc - xe) - Qmax, Qmin - (xc + xe)))
    f_true = ((gap ** 2).sum(1) < thr * thr).mean()
    print("bits %d leaf %d: leaves passing for random-triple hypotheses: mean %.3f median %.3f max %.3f (inliers median %d); true pose %.3f" %
          (bits, LEAF, np.mean(fr), np.median(fr), np.max(fr), int(np.median(inl)), f_true))
