"""BASELINE.json's configurations at their own sizes (C1 is tests/test_gpu_demo_chain.py, the 200k headline is
tests/test_gpu_fullsize.py; C5 needs 8 GPUs and is the driver's run — its sharding logic is tests/test_dist_gloo.py):

  C2  single-pair ICP, 50,000-point scene vs 10,000-point model
  C3  FPFH + RANSAC coarse alignment on 100,000-point clouds, 50,000 hypotheses
  C4  batched bin-picking, 256 instances x ~200k-pixel masks, full chain per instance

The O(N^2) oracle cannot run these in seconds, so each is pinned by exact comparison with the oracle on sampled rows,
by brute-force == pruned on all rows, by a float32 numpy evaluation of the reference's expression for sampled
hypotheses, and by size-independent properties (ground truth recovered, batch == operator chain on sampled instances)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------------ C2
def test_c2_icp_50k_vs_10k(ctx, orc, synth):
    ns, nt = 50000, 10000
    tgt, nrm = synth.sample_object(nt, 42)
    src, T_gt = synth.make_scene(ns, 42)
    voxel = float(synth.mean_spacing(nt))
    thr = 0.4 * voxel                                     # registration.icp_distance_factor, include/pipeline_config.hpp:28
    T0 = synth.perturb(T_gt, angle_deg=0.3, trans=0.0005)
    try:
        ctx.set_icp_search("brute")
        cb = ctx.icp_correspondences(src, tgt, T0, thr)
        a = ctx.icp(src, tgt, nrm, T0, thr, 50, True)
        ctx.set_icp_search("pruned")
        cp = ctx.icp_correspondences(src, tgt, T0, thr)
        b = ctx.icp(src, tgt, nrm, T0, thr, 50, True)
        ctx.set_icp_search("grid")
        cg = ctx.icp_correspondences(src, tgt, T0, thr)
        g2 = ctx.icp(src, tgt, nrm, T0, thr, 50, True)
        assert ctx.last_icp_search() == "grid"            # the reference's threshold, 0.4 spacings: the grid's regime
    finally:
        ctx.set_icp_search("auto")
    # sampled rows against the oracle (the scan reports every row's nearest target, accepted or not)
    sel = np.arange(0, ns, 25)
    ref = orc.icp_correspondences(src[sel], tgt, None, T0, thr, point_to_plane=False)
    assert np.array_equal(cb["corr"][sel], ref["corr"]) and cb["d2"][sel].tobytes() == ref["d2"].tobytes()
    assert np.array_equal(cb["accepted"][sel], ref["accepted"])
    # the pruned search agrees with the scan on ALL rows it accepts, and on which rows those are
    acc = cb["accepted"].astype(bool)
    assert acc.sum() > ns // 10
    assert np.array_equal(cp["accepted"], cb["accepted"]) and cp["n_corr"] == cb["n_corr"]
    assert np.array_equal(cp["corr"][acc], cb["corr"][acc]) and cp["d2"][acc].tobytes() == cb["d2"][acc].tobytes()
    assert np.array_equal(cg["accepted"], cb["accepted"]) and cg["n_corr"] == cb["n_corr"]
    assert np.array_equal(cg["corr"][acc], cb["corr"][acc]) and cg["d2"][acc].tobytes() == cb["d2"][acc].tobytes()
    # identical ICP runs, and the refinement works at the reference's own threshold
    for o in (b, g2):
        assert a.transformation.tobytes() == o.transformation.tobytes()
        assert (a.iterations, a.n_corr, a.rmse, a.fitness) == (o.iterations, o.n_corr, o.rmse, o.fitness)
    ang0, _ = synth.pose_error(T0, T_gt); ang, tr = synth.pose_error(a.transformation, T_gt)
    print("C2: %d iterations, fitness %.3f, angle %.2e -> %.2e rad, translation %.2e m" % (a.iterations, a.fitness, ang0, ang, tr))
    assert ang < ang0 and ang < 2e-3 and tr < 2e-4
    # the oracle's ICP on a 5,000-row subsample from the same start (its scan is O(ns * nt) per iteration)
    sub = np.arange(0, ns, 10)
    g = ctx.icp(src[sub], tgt, nrm, T0, thr, 50, True)
    o = orc.icp(src[sub], tgt, nrm, T0, thr, 50, True)
    da, dt = synth.pose_error(g.transformation, o["T"])
    dcol = float(np.abs(g.transformation[:3, 3].astype(np.float64) - o["T"][:3, 3]).max())
    print("C2 subsample, tree sums vs the oracle: dR %.2e rad, translation column %.2e m, %d iterations" % (da, dcol, g.iterations))
    assert g.iterations == o["iterations"] and da <= 1e-4 and dt <= 1e-6 and dcol <= 1e-6, (g.iterations, o["iterations"], da, dt, dcol)
    # the reference's accumulation order: equal, and what it costs at C2's size (the oracle cannot run 50k x 10k x 50 in a test)
    import time
    ctx.set_icp_accumulation("reference")
    try:
        e = ctx.icp(src[sub], tgt, nrm, T0, thr, 50, True)
        ctx.icp(src, tgt, nrm, T0, thr, 50, True)
        t0 = time.perf_counter(); f = ctx.icp(src, tgt, nrm, T0, thr, 50, True); t_ref = time.perf_counter() - t0
    finally:
        ctx.set_icp_accumulation("tree")
    t0 = time.perf_counter(); a2 = ctx.icp(src, tgt, nrm, T0, thr, 50, True); t_tree = time.perf_counter() - t0
    assert e.transformation.tobytes() == o["T"].tobytes() and e.iterations == o["iterations"]
    assert np.float32(e.rmse).tobytes() == np.float32(o["rmse"]).tobytes() and np.float32(e.fitness).tobytes() == np.float32(o["fitness"]).tobytes()
    da, dt = synth.pose_error(f.transformation, a2.transformation)
    print("C2 50k x 10k whole call (host buffers): reference order %d iterations %.2f ms, tree %d iterations %.2f ms; the two within %.1e rad / %.1e m"
          % (f.iterations, t_ref * 1e3, a2.iterations, t_tree * 1e3, da, dt))
    assert f.iterations == a2.iterations and da <= 1e-4 and dt <= 1e-6


# ------------------------------------------------------------------------------------------------------------ C3
def _score_f32(src, q, T, thr):
    """Inlier count of one hypothesis exactly as the oracle (oracle/oracle.cpp orc_ransac, following registration.cpp:270-279) and
    the kernels (csrc/ransac.hip) evaluate it, in float32 numpy with the same expression tree: every 3-term sum as
    c0 + (c1 + c2) (Eigen's fixed-size redux; oracle/small_linalg.hpp sum3), then + t, the squared norm of the difference
    as d0*d0 + (d1*d1 + d2*d2), its float32 sqrt, strict < threshold.  numpy float32 arithmetic is IEEE round-to-nearest per
    operation, so this is an independent evaluation of the same roundings: the counts must be EQUAL, not close."""
    R = T[:3, :3].astype(np.float32); t = T[:3, 3].astype(np.float32)
    x, y, z = src[:, 0], src[:, 1], src[:, 2]
    out = np.empty((len(src), 3), np.float32)
    for r in range(3):
        out[:, r] = (R[r, 0] * x + (R[r, 1] * y + R[r, 2] * z)) + t[r]
    assert out.dtype == np.float32
    d = out - q
    n2 = d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
    err = np.sqrt(n2)
    assert err.dtype == np.float32
    return int((err < np.float32(thr)).sum())


def test_c3_ransac_50k_hypotheses_at_100k(ctx, orc, synth):
    ns = nt = 100000
    tgt, _ = synth.sample_object(nt, 42)
    src, T_gt = synth.make_scene(ns, 42, outlier_frac=0.05)
    # correspondences as a descriptor match would deliver them: half exact nearest points, half arbitrary
    rng = np.random.default_rng(5)
    corr = rng.integers(0, nt, ns).astype(np.int32)
    good = np.nonzero(rng.random(ns) < 0.5)[0]
    nn = ctx.icp_correspondences(src[good], tgt, T_gt, 1.0)["corr"]
    corr[good] = nn
    voxel = 0.002
    a = ctx.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=50000, confidence=2.0, trace=True)
    b = ctx.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=50000, confidence=2.0, trace=True)
    assert a.iterations_run == 50000 and np.array_equal(a.trace_inliers, b.trace_inliers)
    assert a.transformation.tobytes() == b.transformation.tobytes() and a.best_iteration == b.best_iteration
    tr = a.trace_inliers
    # the index stream is the reference's (mt19937(42) + Lemire), skipped iterations included
    tri = orc.sample_triples(ns, 50000)
    skipped = (tri[:, 0] == tri[:, 1]) | (tri[:, 1] == tri[:, 2]) | (tri[:, 0] == tri[:, 2])
    assert np.array_equal(tr == -1, skipped)
    # the winner is the first iteration with the highest count (strict >, registration.cpp:284)
    assert a.best_iteration == int(np.argmax(tr)) and a.inliers == int(tr.max()) and a.inliers > 0.3 * ns
    # sampled hypotheses: transform from the oracle's 3-point Kabsch, count from the float32 numpy evaluation in the oracle's
    # expression order - equality, no slack
    thr = np.float32(voxel * 1.5)
    for it in [int(a.best_iteration)] + [int(i) for i in np.nonzero(~skipped)[0][[0, 17, 4242, 31337, -1]]]:
        T = orc.hypothesis_from_pairs(src[tri[it].astype(np.int64)], tgt[corr[tri[it].astype(np.int64)]])
        assert _score_f32(src, tgt[corr], T, thr) == int(tr[it]), it
    if a.best_iteration >= 0:
        T = orc.hypothesis_from_pairs(src[tri[a.best_iteration].astype(np.int64)], tgt[corr[tri[a.best_iteration].astype(np.int64)]])
        assert T.tobytes() == a.transformation.tobytes()                 # device SVD == CPU restatement, bit for bit
    ang, _ = synth.pose_error(a.transformation, T_gt)
    assert ang < 2e-2


def test_headline_size_counts_equal_numpy_float32(ctx, orc, synth):
    """The same independent check at the headline size: 200,000 points, 5 hypotheses of a traced call (every test evaluated)
    and the winner of an untraced call (FMA pass + exact bail-out): inlier counts equal the float32 numpy evaluation."""
    ns = nt = 200000
    tgt, _ = synth.sample_object(nt, 42)
    src, T_gt = synth.make_scene(ns, 42)
    rng = np.random.default_rng(9)
    corr = rng.integers(0, nt, ns).astype(np.int32)
    good = np.nonzero(rng.random(ns) < 0.5)[0]
    corr[good] = ctx.icp_correspondences(src[good], tgt, T_gt, 1.0)["corr"]
    voxel = float(np.float32(synth.mean_spacing(ns)))
    thr = np.float32(np.float32(voxel) * np.float32(1.5))
    a = ctx.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=2000, confidence=2.0, trace=True)
    tri = orc.sample_triples(ns, 2000)
    ok = np.nonzero(a.trace_inliers >= 0)[0]
    for it in [int(a.best_iteration)] + [int(i) for i in ok[[0, 3, 777, -1]]]:
        T = orc.hypothesis_from_pairs(src[tri[it].astype(np.int64)], tgt[corr[tri[it].astype(np.int64)]])
        assert _score_f32(src, tgt[corr], T, thr) == int(a.trace_inliers[it]), it
    b = ctx.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=40000, confidence=2.0)       # bail-out path, no trace
    trib = orc.sample_triples(ns, 40000)
    T = orc.hypothesis_from_pairs(src[trib[b.best_iteration].astype(np.int64)], tgt[corr[trib[b.best_iteration].astype(np.int64)]])
    assert T.tobytes() == b.transformation.tobytes()
    assert _score_f32(src, tgt[corr], T, thr) == int(b.inliers)


def test_c3_features_at_100k_register(ctx, tdv, synth):
    """The feature half of C3 on the relief part at ~100k points per side: normals(30) -> FPFH(5 voxels) -> descriptor
    match -> 50,000 hypotheses -> the pose is the ground truth's; the three descriptor-match paths agree."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    bb = importlib.import_module("bench_batch")
    dev = torch.device("cuda", 0)
    wl = bb.build_workload(tdv, synth, ctx, 1, 1.45, 448, 3, tdv.TDV_VOXEL_ORDER_REFERENCE, dev)
    d_mx, d_mn, d_mf, nm = wl["model"]
    mx = d_mx[:nm].cpu().numpy(); mf = d_mf[:nm].cpu().numpy()
    n_px = wl["mask_px"][0]
    d_xyz = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
    n = ctx.depth_to_cloud_dev(wl["depth"][0].data_ptr(), wl["masks"][0].data_ptr(), None, bb.W, bb.H, bb.SCALE, bb.F, bb.F, bb.CX, bb.CY, bb.ZMAX,
                               d_xyz.data_ptr(), None, n_px)
    src, _ = ctx.voxel_downsample(d_xyz[:n].cpu().numpy(), None, wl["voxel"])
    assert 80000 < len(src) < 130000 and 80000 < nm < 130000, (len(src), nm)
    nrm = ctx.estimate_normals(src, 30)
    fp = ctx.compute_fpfh(src, nrm, wl["voxel"] * 5.0)
    corr = ctx.feature_match(fp, mf)
    if tdv.STUDY_BUILD:                       # round 1's key-ordered pruned scan lives in the study library only
        os.environ["TDV_FM_KEYORDER"] = "1"
        try:
            assert np.array_equal(ctx.feature_match(fp, mf), corr)
        finally:
            del os.environ["TDV_FM_KEYORDER"]
    T = wl["T_gt"][0]
    moved = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    right = np.linalg.norm(moved - mx[corr], axis=1) < 1.5 * wl["voxel"]
    coarse = ctx.ransac(src, mx, corr=corr, voxel=wl["voxel"], max_iterations=50000, confidence=0.999)
    ang, tr = synth.pose_error(coarse.transformation, T)
    print("C3 features: %d x %d points, %.1f %% right correspondences, coarse fitness %.3f, angle %.2e rad" % (len(src), nm, 100 * right.mean(), coarse.fitness, ang))
    assert right.mean() > 0.1 and coarse.inliers >= 0.9 * right.sum() and ang < 1e-2


# ------------------------------------------------------------------------------------------------------------ C4
def test_c4_256_instances_full_chain(ctx, tdv, synth):
    """256 distinct instances (own pose, own frame, own ~190k-pixel mask) through tdv_register_batch_dev in the
    reference's voxel order: every instance registers to its ground truth; per-instance point counts equal the
    single-instance operator's; three sampled instances equal the operator-by-operator chain bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    bb = importlib.import_module("bench_batch")
    dev = torch.device("cuda", 0)
    B = 256
    order = tdv.TDV_VOXEL_ORDER_REFERENCE
    wl = bb.build_workload(tdv, synth, ctx, B, 2.0, 448, 3, order, dev)      # voxel = 2 px: ~65k voxels per instance
    d_mx, d_mn, d_mf, nm = wl["model"]
    assert min(wl["mask_px"]) > 150000
    hyps, iters = 10000, 50
    prm = tdv.batch_params(width=bb.W, height=bb.H, scale_to_meters=bb.SCALE, fx=bb.F, fy=bb.F, cx=bb.CX, cy=bb.CY, zmax=bb.ZMAX,
                           voxel_size=wl["voxel"], ransac_max_iterations=hyps, icp_max_iterations=iters, voxel_order=order, n_frames=B)
    res = ctx.register_batch_dev(wl["depth"].data_ptr(), None, wl["masks"].data_ptr(), B, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    assert len(res) == B
    ang = np.array([synth.pose_error(r["T"], T)[0] for r, T in zip(res, wl["T_gt"])])
    tr = np.array([synth.pose_error(r["T"], T)[1] for r, T in zip(res, wl["T_gt"])])
    print("C4: 256 instances, %d-pt model, voxels/instance %d..%d, angle to ground truth max %.2e rad, translation max %.2e m, ICP iterations %d..%d"
          % (nm, min(r["n_voxels"] for r in res), max(r["n_voxels"] for r in res), ang.max(), tr.max(),
             min(r["icp_iterations"] for r in res), max(r["icp_iterations"] for r in res)))
    assert all(r["status"] == 0 for r in res)
    assert [r["n_points"] for r in res] == wl["mask_px"]           # every masked pixel has a valid depth here
    assert ang.max() < 1e-2 and tr.max() < 1e-3
    mx = d_mx[:nm].cpu().numpy(); mn = d_mn[:nm].cpu().numpy(); mf = d_mf[:nm].cpu().numpy()
    for b in (0, 101, 255):
        r = res[b]
        depth = wl["depth"][b].cpu().numpy().view(np.uint16); mask = wl["masks"][b].cpu().numpy()
        xyz, _ = ctx.depth_to_cloud(depth, mask, None, bb.SCALE, bb.F, bb.F, bb.CX, bb.CY, bb.ZMAX)
        src, _ = ctx.voxel_downsample(xyz, None, wl["voxel"], order)
        assert r["n_points"] == len(xyz) and r["n_voxels"] == len(src)
        nrm = ctx.estimate_normals(src, 30)
        fp = ctx.compute_fpfh(src, nrm, wl["voxel"] * 5.0)
        coarse = ctx.ransac(src, mx, fs=fp, ft=mf, voxel=wl["voxel"], max_iterations=hyps, confidence=0.999)
        assert r["coarse_inliers"] == coarse.inliers and r["coarse_fitness"] == coarse.fitness
        fine = ctx.icp(src, mx, mn, coarse.transformation, wl["voxel"] * 0.4, iters, True)
        assert r["icp_iterations"] == fine.iterations and r["T"].tobytes() == fine.transformation.tobytes()
        assert r["fitness"] == fine.fitness and r["rmse"] == fine.rmse
