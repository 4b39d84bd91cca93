"""Offline study behind k_feature_match_pruned (3dvision_amd/csrc/ransac.hip): fraction of 64-target boxes a wave of
key-ordered sources must evaluate (33-D box lower bound, inside-out visiting, per-split bounds) for wave sizes and split
counts.  Result on the synthetic part: 26-35 %.  Run fpfh_descriptors.py first.  CPU only.
"""
import numpy as np
fs = np.load('build/fs.npy').astype(np.float64); ft = np.load('build/ft.npy').astype(np.float64)
ns, nt = len(fs), len(ft)
D = ((fs[:, None, :] - ft[None, :, :]) ** 2).sum(-1)
fn = lambda f: f[:, 5] + f[:, 16] + f[:, 27]
kt_key = fn(ft); ks_key = fn(fs)
# counting-sort like: 4096 buckets
bt = np.minimum(4095, (kt_key * 4096).astype(int)); bs = np.minimum(4095, (ks_key * 4096).astype(int))
kt = np.argsort(bt, kind='stable'); ks = np.argsort(bs, kind='stable')
T = ft[kt]; Tb = bt[kt]
BOX = 64
nb = (nt + BOX - 1) // BOX
bmin = np.stack([T[i*BOX:(i+1)*BOX].min(0) for i in range(nb)]); bmax = np.stack([T[i*BOX:(i+1)*BOX].max(0) for i in range(nb)])
Dk = D[ks][:, kt]; Sx = fs[ks]; Sb = bs[ks]
tstart = np.searchsorted(Tb, np.arange(4097))
for W, NS in [(128, 1), (128, 17), (64, 1), (64, 4), (64, 8), (64, 16)]:
    evaluated = 0; total = 0
    for w in range(0, ns - W + 1, W * 4):
        q = Sx[w:w+W]
        gap = np.maximum(0, np.maximum(bmin[None] - q[:, None, :], q[:, None, :] - bmax[None]))
        lb = (gap ** 2).sum(-1)
        c = min(nb - 1, tstart[Sb[w]] // BOX)
        order = [c]
        for d in range(1, nb):
            if c + d < nb: order.append(c + d)
            if c - d >= 0: order.append(c - d)
        for s in range(NS):
            best = np.full(W, np.inf)
            for b in order[s::NS]:
                total += 1
                if (lb[:, b] <= best).any():
                    evaluated += 1
                    best = np.minimum(best, Dk[w:w+W, b*BOX:(b+1)*BOX].min(1))
    print("W=%3d nsplit=%2d: fraction evaluated %.3f" % (W, NS, evaluated / total))
