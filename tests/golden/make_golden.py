"""Generate the golden fixtures under tests/golden/ from the CPU oracle (run in the build container).

The reference holds no golden vectors, known-answer tests or fixtures for this path and cannot be
built or run here (C++ needing Eigen/OpenCV), so these vectors pin the ORACLE
(oracle/oracle.cpp) — against regressions and against host/compiler drift on the GPU box — not the
reference itself.  Two groups are pinned independently of the oracle:
  * demo-scene invariants (40,401 / 32,129 / 1,476 / 1,600) were computed for SURVEY.md 3.1 with
    numpy straight from src/pipeline.cpp:211-257,275-282;
  * the mt19937 / Lemire index stream is checked against libstdc++'s own std::mt19937 +
    std::uniform_int_distribution<size_t> (what registration.cpp:235-236 instantiates).
Fixtures are data only: inputs are regenerated from seeds by 3dvision_amd/synth.py, outputs are stored.

    python tests/golden/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402

synth = importlib.import_module("3dvision_amd.synth")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cloud(n, seed=42):
    pts, _ = synth.sample_object(n, seed)
    T = synth.gt_transform(seed)
    Tinv = np.linalg.inv(T.astype(np.float64))
    return (pts.astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)


def main():
    meta = {}
    # ---- demo scene (config C1 inputs)
    depth, bgr = orc.demo_scene(); mask = orc.demo_mask(); model, mnrm = orc.demo_model()
    d = orc.depth_preprocess(depth, mask, 1000.0)
    xyz, rgb = orc.unproject(d, bgr, 900, 900, 640, 360, 1.5)
    v1, c1, f1 = orc.voxel_downsample(xyz, rgb, 0.001)
    v5, _, _ = orc.voxel_downsample(xyz, None, 0.005)
    vm, _, _ = orc.voxel_downsample(model, None, 0.001)
    meta["demo"] = dict(points=len(xyz), z_values=[float(z) for z in np.unique(xyz[:, 2])],
                        z_counts=[int(c) for c in np.unique(xyz[:, 2], return_counts=True)[1]],
                        voxels_1mm=len(v1), voxels_5mm=len(v5), model_points=len(model), model_voxels_1mm=len(vm),
                        nonzero_depth=int(orc.count_nonzero(d)),
                        sha_depth=sha(d), sha_xyz=sha(xyz), sha_rgb=sha(rgb), sha_voxel_1mm_xyz=sha(v1), sha_voxel_1mm_rgb=sha(c1),
                        sha_model=sha(model))
    # ---- index stream
    tri = {str(n): orc.sample_triples(n, 64).astype(np.int64) for n in (1600, 32129, 200000)}
    # ---- per-op vectors at small N
    n = 640
    pts = cloud(n)
    nrm, knn = orc.estimate_normals(pts, 30, want_knn=True)
    radius = 0.02
    desc, nb, cnt = orc.compute_fpfh(pts, nrm, radius, want_neighbors=True)
    vx, _, vfirst = orc.voxel_downsample(pts, None, 0.01)
    tgt, tnrm = synth.sample_object(512, 7)
    src, T_gt = synth.make_scene(n, 7)
    ft = synth.random_features(512, 3); fs = synth.random_features(n, 4)
    corr = orc.feature_match(fs, ft)
    nncorr = orc.icp_correspondences(src, tgt, None, T_gt, 1.0, point_to_plane=False)["corr"]
    rs = orc.ransac(src, tgt, corr=nncorr, voxel=0.006, max_iterations=400, confidence=2.0, trace=True)
    T0 = synth.perturb(T_gt, 7)
    icp_pl = orc.icp(src, tgt, tnrm, T0, 0.008, 40, True, trace=True)
    icp_pt = orc.icp(src, tgt, tnrm, T0, 0.008, 40, False, trace=True)
    c0 = orc.icp_correspondences(src, tgt, tnrm, T0, 0.008)
    # ---- rank-deficient point-to-plane system (planar model: SURVEY H4)
    rng = np.random.Generator(np.random.PCG64(5))
    p = (rng.random((64, 3)) - 0.5).astype(np.float32)
    J = np.zeros((64, 6), np.float32); J[:, 0] = p[:, 1]; J[:, 1] = -p[:, 0]; J[:, 5] = 1
    A = (J.T @ J).astype(np.float32); b = (J.T @ (rng.random(64).astype(np.float32) - 0.5)).astype(np.float32)
    x = orc.ldlt6_solve(A, b)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"),
                        triples_1600=tri["1600"], triples_32129=tri["32129"], triples_200000=tri["200000"],
                        demo_voxel_first64=v1[:64], demo_voxel_first_index64=f1[:64],
                        normals=nrm, knn=knn, fpfh=desc, fpfh_nbr_cnt=cnt, fpfh_nbr=nb.astype(np.int16),
                        voxel_xyz=vx, voxel_first=vfirst,
                        feature_corr=corr, nn_corr=nncorr,
                        ransac_inliers=rs["inliers"], ransac_T=rs["T"], ransac_best=np.array([rs["best_iter"], rs["iters_run"]]),
                        ransac_fit_rmse=np.array([rs["fitness"], rs["rmse"]], np.float32),
                        icp_pl_trace=icp_pl["trace"], icp_pl_T=icp_pl["T"], icp_pt_trace=icp_pt["trace"], icp_pt_T=icp_pt["T"],
                        icp_c0_corr=c0["corr"], icp_c0_d2=c0["d2"], icp_c0_acc=c0["accepted"], icp_c0_ATA=c0["ATA"], icp_c0_ATb=c0["ATb"],
                        ldlt_A=A, ldlt_b=b, ldlt_x=x)
    meta["params"] = dict(n=n, radius=radius, voxel=0.01, ransac_voxel=0.006, icp_thr=0.008, seeds=dict(cloud=42, pair=7))
    meta["icp_c0_n_corr"] = int(c0["n_corr"])
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta["demo"], indent=1))
    print("wrote", os.path.getsize(os.path.join(HERE, "vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
