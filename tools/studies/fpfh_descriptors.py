"""Offline study input: real FPFH descriptors of the synthetic part (scene 12k points, model 6k) computed by the CPU oracle,
saved under build/ for the pruning studies in this directory.  CPU only.

    python tools/studies/fpfh_descriptors.py
"""
import sys, importlib, time
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import pyoracle as orc
synth = importlib.import_module('3dvision_amd.synth')
ns, nt = 12000, 6000
tgt, _ = synth.sample_object(nt, 7)
src, T = synth.make_scene(ns, 42)
vox = float(synth.mean_spacing(nt))
def feats(x):
    n = orc.estimate_normals(x, 30)
    return orc.compute_fpfh(x, n, 5 * vox)
t0 = time.time()
ft = feats(tgt); fs = feats(src)
print("fpfh done", time.time() - t0, fs.shape, ft.shape)
import os
os.makedirs('build', exist_ok=True)
np.save('build/fs.npy', fs); np.save('build/ft.npy', ft)
