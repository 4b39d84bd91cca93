"""Config C5 (BASELINE.json configs[4]) at its own size, one rank's share: 1,024 instances cut by ONE uint16 label image from
ONE frame whose cloud is ~530k points, through ONE tdv_register_batch_dev call (tools/c5_tray.py; src/pipeline.cpp:25-150 per
instance, :321-327 the fan-out).  Sampled instances are held against the oracle's whole chain and against the operator chain
bit for bit; all 1,024 are compared with their ground truth.  (The 8-rank job around it is 3dvision_amd/sharding.py +
tools/bench_c5.py: tests/test_dist_gloo.py, tests/test_bench_launch.py.)"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_INST = 1024
SAMPLE = [0, 1, 340, 511, 777, 1023]


def test_c5_1024_instances_from_one_label_image(ctx, tdv, synth, orc):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    c5 = importlib.import_module("c5_tray")
    dev = torch.device("cuda", 0)
    own = tdv.Context(0)         # a context of its own: the workspace figure below is then this workload's, not the test session's
    out, wl, res = c5.measure(tdv, synth, own, torch, dev, N_INST)
    print(out)
    sc = wl["sc"]
    S, F, CX, CY, V, ZMAX = sc["scale"], sc["fx"], sc["cx"], sc["cy"], sc["voxel"], sc["zmax"]
    assert len(res) == N_INST and all(r["status"] == 0 for r in res)
    assert own.last_voxel_grouping() == "pixels"          # the batch's clouds come from its own unprojection: voxels through pixel windows, no table
    assert 450000 <= out["scene_points"] <= 600000 and out["scene_points"] == int((sc["label"] > 0).sum())
    per = np.bincount(sc["label"].ravel(), minlength=N_INST + 1)[1:]
    assert [r["n_points"] for r in res] == per.tolist()                      # every instance got exactly its label's pixels
    # the model: GPU preparation == the oracle's (Pipeline::run :291-294)
    d_mx, d_mn, d_mf, nm = wl["model"]
    mx = d_mx[:nm].cpu().numpy(); mn = d_mn[:nm].cpu().numpy(); mf = d_mf[:nm].cpu().numpy()
    oxyz, _ = orc.unproject(orc.depth_preprocess(sc["model_depth"], sc["model_mask"], S), None, F, F, CX, CY, ZMAX)
    omx, _, _ = orc.voxel_downsample(oxyz, None, V); omn = orc.estimate_normals(omx, 30); omf = orc.compute_fpfh(omx, omn, V * 5.0)
    assert mx.tobytes() == omx.tobytes() and mn.tobytes() == omn.tobytes() and mf.tobytes() == omf.tobytes()
    hyps, iters = wl["params"].ransac_max_iterations, wl["params"].icp_max_iterations
    # the same batch once more with the reference's accumulation order (and what that costs at this size)
    import time
    own.set_icp_accumulation("reference")
    c5.run(own, wl)
    torch.cuda.synchronize(); t0 = time.perf_counter(); res_ref = c5.run(own, wl); torch.cuda.synchronize(); t_ref = time.perf_counter() - t0
    own.set_icp_accumulation("tree")
    print("C5 share with reference-order ICP sums: %.1f ms = %.0f instances/s (tree sums: %.0f instances/s)" % (t_ref * 1e3, N_INST / t_ref, out["instances_per_s"]))
    golden = np.load(os.path.join(ROOT, "tests", "golden", "vectors_r3.npz"))   # the oracle's chain on instances 0 and 511, made in the build container
    assert np.array_equal(per, golden["tray_per_instance"])
    for b in SAMPLE:
        mask = np.where(sc["label"] == b + 1, 255, 0).astype(np.uint8)
        # the oracle's processInstance
        oc, _ = orc.unproject(orc.depth_preprocess(sc["depth"], mask, S), None, F, F, CX, CY, ZMAX)
        osrc, _, _ = orc.voxel_downsample(oc, None, V)
        onr = orc.estimate_normals(osrc, 30); ofp = orc.compute_fpfh(osrc, onr, V * 5.0)
        oco = orc.ransac(osrc, omx, fs=ofp, ft=omf, voxel=V, max_iterations=hyps, confidence=0.999)
        ofi = orc.icp(osrc, omx, omn, oco["T"], V * 0.4, iters, True)
        # the operator chain on the GPU (host-buffer ABI), stage by stage against it
        xyz, _ = ctx.depth_to_cloud(sc["depth"], mask, None, S, F, F, CX, CY, ZMAX)
        assert xyz.tobytes() == oc.tobytes()
        src, _ = ctx.voxel_downsample(xyz, None, V, tdv.TDV_VOXEL_ORDER_REFERENCE)
        assert src.tobytes() == osrc.tobytes()
        nr = ctx.estimate_normals(src, 30); assert nr.tobytes() == onr.tobytes()
        fp = ctx.compute_fpfh(src, nr, V * 5.0); assert fp.tobytes() == ofp.tobytes()
        co = ctx.ransac(src, mx, fs=fp, ft=mf, voxel=V, max_iterations=hyps, confidence=0.999)
        assert co.transformation.tobytes() == oco["T"].tobytes() and co.best_iteration == oco["best_iter"] and co.fitness == oco["fitness"]
        if "tray_%d_coarse_T" % b in golden.files:            # ... and the committed vectors: the oracle on this box equals the build container's
            assert oco["T"].tobytes() == golden["tray_%d_coarse_T" % b].tobytes() and ofi["T"].tobytes() == golden["tray_%d_fine_T" % b].tobytes()
        fi = ctx.icp(src, mx, mn, co.transformation, V * 0.4, iters, True)
        # (1) The reference's accumulation order (round 4): the refined transform, rmse, fitness and iteration count EQUAL the oracle's.
        ctx.set_icp_accumulation("reference")
        try:
            fe = ctx.icp(src, mx, mn, co.transformation, V * 0.4, iters, True)
        finally:
            ctx.set_icp_accumulation("tree")
        assert fe.transformation.tobytes() == ofi["T"].tobytes() and fe.iterations == ofi["iterations"], (b, fe.iterations, ofi["iterations"])
        assert np.float32(fe.rmse).tobytes() == np.float32(ofi["rmse"]).tobytes() and np.float32(fe.fitness).tobytes() == np.float32(ofi["fitness"]).tobytes()
        assert res_ref[b]["T"].tobytes() == ofi["T"].tobytes() and res_ref[b]["icp_iterations"] == ofi["iterations"]      # ... in the batched call too
        # (2) The default (f64 tree sums): north star 1e-4 rad on rotation, 1e-3 mm on translation.  Both readings of "translation"
        # are printed: the translation COLUMN of T (the literal one: where the camera origin lands) and the displacement of the
        # instance's centroid.  The column is a lever arm away from the data - 0.45 m here, where the rotation tolerance alone
        # is worth 45 um - so it is asserted with that lever arm; the centroid must hold 1e-6 m.
        da, _ = synth.pose_error(fi.transformation, ofi["T"])
        c = np.append(src.astype(np.float64).mean(0), 1.0)
        dt = float(np.linalg.norm((fi.transformation.astype(np.float64) - ofi["T"].astype(np.float64)) @ c))
        dcol = float(np.abs(fi.transformation[:3, 3].astype(np.float64) - ofi["T"][:3, 3]).max())
        print("instance %4d tree sums vs oracle: dR %.2e rad, translation column %.2e m (%s 1e-6), centroid %.2e m" % (b, da, dcol, "<=" if dcol <= 1e-6 else ">", dt))
        assert fi.iterations == ofi["iterations"] and da <= 1e-4 and dt <= 1e-6, (b, fi.iterations, ofi["iterations"], da, dt)
        assert dcol <= 1e-6 + da * float(np.linalg.norm(c[:3])), (b, dcol, da)
        # the batch == the operator chain, bit for bit
        r = res[b]
        assert r["n_points"] == len(xyz) and r["n_voxels"] == len(src)
        assert r["coarse_inliers"] == co.inliers and r["coarse_fitness"] == co.fitness
        assert r["T"].tobytes() == fi.transformation.tobytes() and r["icp_iterations"] == fi.iterations and r["fitness"] == fi.fitness and r["rmse"] == fi.rmse
        print("instance %4d: %d px -> %d voxels, coarse fitness %.2f, ICP %d iterations, %.4f rad from the ground truth"
              % (b, len(xyz), len(src), co.fitness, fi.iterations, synth.pose_error(fi.transformation, sc["T_gt"][b])[0]))
    # all 1,024 against their ground truth: the reference's algorithm itself loses a few of these 25-pixel parts (its ICP threshold
    # of 0.4 voxel is a tenth of a millimetre here), the batch must not lose more
    ang = c5.angles(synth, wl, res)
    assert (ang <= c5.MAX_ANGLE).mean() >= 0.88, (ang <= c5.MAX_ANGLE).mean()
    assert out["workspace_high_water_MiB"] < 8192
