"""Round-3 additions to the golden fixtures (tests/golden/vectors_r3.npz, meta_r3.json), generated from the CPU oracle in the build
container like make_golden.py's - they pin the ORACLE and the synthetic generators against regressions and host / compiler drift,
not the reference (which holds no vectors for this path):
  * cv::resize(INTER_NEAREST) restatement (src/pipeline.cpp:38-41): a seeded 37 x 53 mask resized to 211 x 97 and to 16 x 20;
  * the C5 tray (3dvision_amd/synth.py: tray_scene, seed 7): scene size, per-instance pixel counts, and the oracle's whole
    processInstance chain on instances 0 and 511 (voxel count, RANSAC winner, refined transform, ICP iterations).

    python tests/golden/make_golden_r3.py
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import pyoracle as orc  # noqa: E402

synth = importlib.import_module("3dvision_amd.synth")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tray_chain(sc, model, b, hyps=10000, iters=50):
    S, F, CX, CY, V, Z = sc["scale"], sc["fx"], sc["cx"], sc["cy"], sc["voxel"], sc["zmax"]
    m = np.where(sc["label"] == b + 1, 255, 0).astype(np.uint8)
    p, _ = orc.unproject(orc.depth_preprocess(sc["depth"], m, S), None, F, F, CX, CY, Z)
    s, _, _ = orc.voxel_downsample(p, None, V); n = orc.estimate_normals(s, 30); f = orc.compute_fpfh(s, n, V * 5.0)
    co = orc.ransac(s, model[0], fs=f, ft=model[2], voxel=V, max_iterations=hyps, confidence=0.999)
    fi = orc.icp(s, model[0], model[1], co["T"], V * 0.4, iters, True)
    return dict(points=len(p), voxels=len(s), best_iter=int(co["best_iter"]), coarse_fitness=float(co["fitness"]), iterations=int(fi["iterations"]),
                fitness=float(fi["fitness"])), co["T"], fi["T"]


def main():
    vec, meta = {}, {}
    rng = np.random.default_rng(2026)
    mask = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    up = orc.mask_resize_nearest(mask, 97, 211); down = orc.mask_resize_nearest(mask, 20, 16)
    vec["resize_down_16x20"] = down
    meta["resize"] = dict(src_sha=sha(mask), up_211x97_sha=sha(up), down_16x20_sha=sha(down))
    c5 = importlib.import_module("c5_tray")
    sc = synth.tray_scene(1024, seed=7, **c5.TRAY)
    per = np.bincount(sc["label"].ravel(), minlength=1025)[1:]
    S, F, CX, CY, V, Z = sc["scale"], sc["fx"], sc["cx"], sc["cy"], sc["voxel"], sc["zmax"]
    x, _ = orc.unproject(orc.depth_preprocess(sc["model_depth"], sc["model_mask"], S), None, F, F, CX, CY, Z)
    mx, _, _ = orc.voxel_downsample(x, None, V); mn = orc.estimate_normals(mx, 30); mf = orc.compute_fpfh(mx, mn, V * 5.0)
    meta["tray"] = dict(scene_points=int(per.sum()), per_instance_sha=sha(per.astype(np.int32)), depth_sha=sha(sc["depth"]), label_sha=sha(sc["label"]),
                        voxel=float(sc["voxel"]), model_points=len(x), model_voxels=len(mx), model_fpfh_sha=sha(mf))
    vec["tray_per_instance"] = per.astype(np.int32)
    for b in (0, 511):
        info, Tc, Tf = tray_chain(sc, (mx, mn, mf), b)
        meta["tray"]["instance_%d" % b] = info
        vec["tray_%d_coarse_T" % b] = Tc; vec["tray_%d_fine_T" % b] = Tf
    np.savez_compressed(os.path.join(HERE, "vectors_r3.npz"), **vec)
    json.dump(meta, open(os.path.join(HERE, "meta_r3.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True)[:1500])


if __name__ == "__main__":
    main()
