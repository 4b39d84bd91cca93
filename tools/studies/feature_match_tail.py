#!/usr/bin/env python3
"""Offline study: distribution (not the mean) of the leaves an exact nearest-descriptor query must open under the STR
packing — the kernel's run time is set by its slowest waves."""
import sys
import numpy as np
sys.argv = [sys.argv[0]] + sys.argv[1:]
tag = sys.argv[1] if len(sys.argv) > 1 else "2p0"
ft = np.load("gpurun_out/desc_model_%s.npy" % tag); fs = np.load("gpurun_out/desc_inst_%s.npy" % tag)
nt, ns = len(ft), len(fs)
mu = ft.mean(0, dtype=np.float64)
w, V = np.linalg.eigh(np.cov((ft - mu).T.astype(np.float64))); V = V[:, ::-1]
P = (ft - mu) @ V[:, :3]
n = nt; nleaf = n // 64
ext = P.std(0)
g = (nleaf / np.prod(ext)) ** (1 / 3)
s0 = max(1, int(round(ext[0] * g))); s1 = max(1, int(round(ext[1] * g)))
r0 = np.argsort(P[:, 0], kind="stable"); slab = np.empty(n, np.int64); slab[r0] = np.arange(n) * s0 // n
order = np.lexsort((P[:, 1], slab)); col = np.empty(n, np.int64)
# equal-count columns inside each slab
pos = np.empty(n, np.int64); pos[order] = np.arange(n)
start = np.searchsorted(slab[order], np.arange(s0)); cnt = np.diff(np.r_[start, n])
col = slab * s1 + ((pos - start[slab]) * s1 // cnt[slab])
final = np.lexsort((P[:, 2], col))
T = ft[final]; tcol = col[final]
rows = []
starts = np.r_[0, np.nonzero(np.diff(tcol))[0] + 1, n]
for a, b in zip(starts[:-1], starts[1:]):
    blk = T[a:b]; pad = (-len(blk)) % 64
    rows.append(blk)
    if pad: rows.append(np.repeat(blk[-1:], pad, 0))
Tp = np.concatenate(rows); nbox = len(Tp) // 64
bmin = Tp.reshape(nbox, 64, 33).min(1).astype(np.float64); bmax = Tp.reshape(nbox, 64, 33).max(1).astype(np.float64)
rng = np.random.default_rng(1)
sel = rng.choice(ns, 3000, replace=False)
Td = ft.astype(np.float64); tn = (Td ** 2).sum(1)
need = []; bestd = []
for i0 in range(0, len(sel), 250):
    q = fs[sel[i0:i0 + 250]].astype(np.float64)
    d2 = np.maximum((q * q).sum(1)[:, None] + tn[None] - 2 * q @ Td.T, 0)
    best = d2.min(1)
    gap = np.maximum(np.maximum(bmin[None] - q[:, None, :], q[:, None, :] - bmax[None]), 0)
    lb = (gap ** 2).sum(-1)
    need.append((lb <= best[:, None] * (1 + 1e-6) + 1e-12).sum(1)); bestd.append(best)
need = np.concatenate(need); bestd = np.sqrt(np.concatenate(bestd))
print("slabs x columns", s0, s1, "leaves", nbox)
print("leaves needed per query: mean %.1f  median %d  p90 %d  p99 %d  max %d" % (need.mean(), np.median(need), *np.percentile(need, [90, 99]).astype(int), need.max()))
print("nearest distance:        mean %.4f median %.4f p90 %.4f p99 %.4f max %.4f" % (bestd.mean(), np.median(bestd), *np.percentile(bestd, [90, 99]), bestd.max()))
heavy = need > np.percentile(need, 95)
print("share of all leaf openings caused by the heaviest 5 %% of the queries: %.1f %%" % (100 * need[heavy].sum() / need.sum()))
np.save("/tmp/need_%s.npy" % tag, np.stack([sel, need]))

# ---- would a second box per leaf, aligned with the principal directions, cut the openings? (lower bound = max of the two)
for m in (3, 6, 10):
    Q = V[:, :m]
    Pp = ((Tp.astype(np.float64) - mu) @ Q).reshape(nbox, 64, m)
    pmin = Pp.min(1); pmax = Pp.max(1)
    need2 = []
    for i0 in range(0, len(sel), 250):
        q = fs[sel[i0:i0 + 250]].astype(np.float64)
        d2 = np.maximum((q * q).sum(1)[:, None] + tn[None] - 2 * q @ Td.T, 0)
        best = d2.min(1)
        gap = np.maximum(np.maximum(bmin[None] - q[:, None, :], q[:, None, :] - bmax[None]), 0)
        lb = (gap ** 2).sum(-1)
        pq = (q - mu) @ Q
        pg = np.maximum(np.maximum(pmin[None] - pq[:, None, :], pq[:, None, :] - pmax[None]) - 4e-5, 0)
        lb2 = (pg ** 2).sum(-1) * (1 - 1e-5)
        need2.append((np.maximum(lb, lb2) <= best[:, None] * (1 + 1e-6) + 1e-12).sum(1))
    need2 = np.concatenate(need2)
    print("with a PCA-%d box as well: mean %.1f  median %d  p90 %d  p99 %d  max %d" % (m, need2.mean(), np.median(need2), *np.percentile(need2, [90, 99]).astype(int), need2.max()))

# ---- and the principal-direction box ALONE (no 33-D box at all)?
for m in (3, 4, 6):
    Q = V[:, :m]
    Pp = ((Tp.astype(np.float64) - mu) @ Q).reshape(nbox, 64, m)
    pmin = Pp.min(1); pmax = Pp.max(1)
    need3 = []
    for i0 in range(0, len(sel), 250):
        q = fs[sel[i0:i0 + 250]].astype(np.float64)
        d2 = np.maximum((q * q).sum(1)[:, None] + tn[None] - 2 * q @ Td.T, 0)
        best = d2.min(1)
        pq = (q - mu) @ Q
        pg = np.maximum(np.maximum(pmin[None] - pq[:, None, :], pq[:, None, :] - pmax[None]) - 4e-5, 0)
        lb2 = (pg ** 2).sum(-1) * (1 - 1e-5)
        need3.append((lb2 <= best[:, None] * (1 + 1e-6) + 1e-12).sum(1))
    need3 = np.concatenate(need3)
    print("PCA-%d box alone: mean %.1f  median %d  p90 %d  p99 %d  max %d" % (m, need3.mean(), np.median(need3), *np.percentile(need3, [90, 99]).astype(int), need3.max()))

# ---- a full 33-D box in the ROTATED (principal) frame, alone and beside the axis-aligned one
Q = V
Pp = ((Tp.astype(np.float64) - mu) @ Q).reshape(nbox, 64, 33)
pmin = Pp.min(1); pmax = Pp.max(1)
needr, needb = [], []
for i0 in range(0, len(sel), 250):
    q = fs[sel[i0:i0 + 250]].astype(np.float64)
    d2 = np.maximum((q * q).sum(1)[:, None] + tn[None] - 2 * q @ Td.T, 0)
    best = d2.min(1)
    gap = np.maximum(np.maximum(bmin[None] - q[:, None, :], q[:, None, :] - bmax[None]), 0)
    lb = (gap ** 2).sum(-1)
    pq = (q - mu) @ Q
    pg = np.maximum(np.maximum(pmin[None] - pq[:, None, :], pq[:, None, :] - pmax[None]) - 4e-5, 0)
    lb2 = (pg ** 2).sum(-1) * (1 - 1e-5)
    needr.append((lb2 <= best[:, None] * (1 + 1e-6) + 1e-12).sum(1))
    needb.append((np.maximum(lb, lb2) <= best[:, None] * (1 + 1e-6) + 1e-12).sum(1))
needr = np.concatenate(needr); needb = np.concatenate(needb)
print("rotated 33-D box alone: mean %.1f median %d p90 %d p99 %d max %d" % (needr.mean(), np.median(needr), *np.percentile(needr, [90, 99]).astype(int), needr.max()))
print("rotated 33-D + axis-aligned: mean %.1f median %d p90 %d p99 %d max %d" % (needb.mean(), np.median(needb), *np.percentile(needb, [90, 99]).astype(int), needb.max()))
