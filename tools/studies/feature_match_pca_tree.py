#!/usr/bin/env python3
"""Offline study (numpy, real FPFH descriptors dumped by dump_descriptors.py): how many 64-target leaf boxes must an
exact nearest-descriptor search open, per query and per wave of 64 neighbouring queries, when both sides are ordered
(a) by round 1's scalar key (three centre bins) or (b) along a Morton curve over the top-m principal directions of the
target descriptors?  A box must be opened iff its lower bound <= the query's true nearest distance (perfect bound
order); a wave of 64 lane-queries opens the union."""
import sys
import numpy as np

tag = sys.argv[1] if len(sys.argv) > 1 else "1p2"
ft = np.load("gpurun_out/desc_model_%s.npy" % tag); fs = np.load("gpurun_out/desc_inst_%s.npy" % tag)
nt, ns = len(ft), len(fs)
print("targets", nt, "sources", ns)
mu = ft.mean(0, dtype=np.float64)
C = np.cov((ft - mu).T.astype(np.float64))
w, V = np.linalg.eigh(C); w = w[::-1]; V = V[:, ::-1]
print("variance share of the top components:", np.round(np.cumsum(w)[:8] / w.sum(), 4))


def morton(P, bits):
    """P: [n, m] in [0,1) -> interleaved code."""
    m = P.shape[1]
    q = np.clip((P * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(P), np.int64)
    for b in range(bits):
        for d in range(m):
            code |= ((q[:, d] >> b) & 1) << (b * m + d)
    return code


def order_pca(f, m, bits, lo, hi):
    P = ((f - mu) @ V[:, :m] - lo) / (hi - lo)
    return np.argsort(morton(P, bits), kind="stable")


def order_key(f):
    return np.argsort(f[:, 5] + f[:, 16] + f[:, 27], kind="stable")


rng = np.random.default_rng(0)


def study(name, tperm, sperm):
    T = ft[tperm]
    nbox = (nt + 63) // 64
    pad = nbox * 64 - nt
    Tp = np.concatenate([T, np.repeat(T[-1:], pad, 0)]) if pad else T
    bmin = Tp.reshape(nbox, 64, 33).min(1); bmax = Tp.reshape(nbox, 64, 33).max(1)
    # waves: 64 consecutive ordered sources; sample 24 waves
    S = fs[sperm]
    waves = rng.choice(ns // 64, 24, replace=False)
    per_q, per_w, per_g = [], [], []
    for wv in waves:
        q = S[wv * 64:(wv + 1) * 64].astype(np.float64)
        d2 = ((q[:, None, :] - T[None, :, :].astype(np.float64)) ** 2).sum(-1) if nt <= 70000 else None
        if d2 is None:
            d2 = (q * q).sum(1)[:, None] + (T.astype(np.float64) ** 2).sum(1)[None] - 2 * q @ T.astype(np.float64).T
        best = np.maximum(d2.min(1), 0)
        gap = np.maximum(np.maximum(bmin[None] - q[:, None, :], q[:, None, :] - bmax[None]), 0)
        lb = (gap ** 2).sum(-1)                       # [64, nbox]
        need = lb <= best[:, None] * (1 + 1e-6) + 1e-12
        per_q.append(need.sum(1).mean()); per_w.append(need.any(0).sum())
        # groups of 64 leaves: opened if any needed leaf inside
        ng = (nbox + 63) // 64
        gneed = np.zeros(ng, bool); gneed[np.nonzero(need.any(0))[0] // 64] = True
        per_g.append(gneed.sum())
    print("%-28s leaves/query %.1f   leaves/wave %.1f (of %d = %.2f %%)   groups/wave %.1f of %d" % (
        name, np.mean(per_q), np.mean(per_w), nbox, 100 * np.mean(per_w) / nbox, np.mean(per_g), (nbox + 63) // 64))


study("round-1 key (centre bins)", order_key(ft), order_key(fs))
for m, bits in [(2, 10), (3, 8), (4, 6), (5, 5), (6, 4)]:
    Pt = (ft - mu) @ V[:, :m]
    lo = Pt.min(0); hi = Pt.max(0) + 1e-9
    study("PCA-%d Morton (%d bits)" % (m, bits), order_pca(ft, m, bits, lo, hi), order_pca(fs, m, bits, lo, hi))


def order_kd(f, m, leaf=64):
    """k-d order: recursive median split along the widest of the top-m principal coordinates."""
    P = (f - mu) @ V[:, :m]
    idx = np.arange(len(f))
    out = []
    stack = [idx]
    while stack:
        a = stack.pop()
        if len(a) <= leaf:
            out.append(a); continue
        ext = P[a].max(0) - P[a].min(0)
        d = int(np.argmax(ext))
        # split at a multiple of `leaf` nearest the median so leaves stay full
        h = max(leaf, (len(a) // 2 + leaf - 1) // leaf * leaf)
        part = np.argpartition(P[a, d], h - 1)
        stack.append(a[part[h:]]); stack.append(a[part[:h]])
    return np.concatenate(out)


def order_kd_sources_like_targets(m):
    """Sources ordered by the same k-d cells is not available cheaply; order them by their own k-d tree."""
    return order_kd(fs, m)


for m in (3, 4, 5):
    study("k-d tree on PCA-%d" % m, order_kd(ft, m), order_kd(fs, m))


def order_str(f, m_dims, leaf=64, ref=None):
    """Sort-tile-recursive packing on the top principal coordinates: equal-count slabs along p0, equal-count columns
    along p1 inside every slab, rows sorted along p2 inside every column; slab counts proportional to the extents.
    ref = (bounds0, bounds1) of another set: cut with ITS boundaries (how sources are bucketed into target cells)."""
    P = (f - mu) @ V[:, :3]
    n = len(f)
    nleaf = max(1, n // leaf)
    if ref is None:
        ext = P.max(0) - P.min(0)
        # s0 * s1 * s2 = nleaf with s_d proportional to ext_d
        g = (nleaf / np.prod(ext)) ** (1 / 3)
        s0 = max(1, int(round(ext[0] * g))); s1 = max(1, int(round(ext[1] * g)))
        b0 = np.quantile(P[:, 0], np.linspace(0, 1, s0 + 1)[1:-1])
        c0 = np.searchsorted(b0, P[:, 0])
        b1 = [np.quantile(P[c0 == k, 1], np.linspace(0, 1, s1 + 1)[1:-1]) if (c0 == k).any() else np.zeros(s1 - 1) for k in range(s0)]
    else:
        b0, b1 = ref
        s0 = len(b0) + 1; s1 = len(b1[0]) + 1
        c0 = np.searchsorted(b0, P[:, 0])
    c1 = np.empty(n, np.int64)
    for k in range(s0):
        mk = c0 == k
        c1[mk] = np.searchsorted(b1[k], P[mk, 1])
    col = c0 * s1 + c1
    order = np.lexsort((P[:, 2], col))
    return order, col[order], (b0, b1), (s0, s1)


def study_padded(name, tperm, tcol, sperm):
    """like study(), but every column of targets is padded to a multiple of 64 rows (leaves never straddle columns)."""
    rows = []
    T = ft[tperm]
    starts = np.r_[0, np.nonzero(np.diff(tcol))[0] + 1, len(tcol)]
    for a, b in zip(starts[:-1], starts[1:]):
        blk = T[a:b]
        pad = (-len(blk)) % 64
        rows.append(blk)
        if pad: rows.append(np.repeat(blk[-1:], pad, 0))
    Tp = np.concatenate(rows)
    nbox = len(Tp) // 64
    bmin = Tp.reshape(nbox, 64, 33).min(1); bmax = Tp.reshape(nbox, 64, 33).max(1)
    S = fs[sperm]
    waves = rng.choice(ns // 64, 24, replace=False)
    per_q, per_w, per_g = [], [], []
    Td = T.astype(np.float64); tn = (Td ** 2).sum(1)
    for wv in waves:
        q = S[wv * 64:(wv + 1) * 64].astype(np.float64)
        d2 = (q * q).sum(1)[:, None] + tn[None] - 2 * q @ Td.T
        best = np.maximum(d2.min(1), 0)
        gap = np.maximum(np.maximum(bmin[None] - q[:, None, :], q[:, None, :] - bmax[None]), 0)
        lb = (gap ** 2).sum(-1)
        need = lb <= best[:, None] * (1 + 1e-6) + 1e-12
        per_q.append(need.sum(1).mean()); per_w.append(need.any(0).sum())
        ng = (nbox + 63) // 64
        gneed = np.zeros(ng, bool); gneed[np.nonzero(need.any(0))[0] // 64] = True
        per_g.append(gneed.sum())
    print("%-34s leaves/query %.1f   leaves/wave %.1f (of %d = %.2f %%)   groups/wave %.1f of %d" % (
        name, np.mean(per_q), np.mean(per_w), nbox, 100 * np.mean(per_w) / nbox, np.mean(per_g), (nbox + 63) // 64))


tperm, tcol, bounds, shape = order_str(ft, 3)
print("STR slabs x columns:", shape)
sperm, _, _, _ = order_str(fs, 3, ref=bounds)
study_padded("STR targets / sources in target cells", tperm, tcol, sperm)
sperm2, _, _, _ = order_str(fs, 3)
study_padded("STR targets / STR sources (own)", tperm, tcol, sperm2)
