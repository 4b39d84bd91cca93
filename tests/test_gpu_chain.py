"""End-to-end parity of Pipeline::processInstance (src/pipeline.cpp:25-150) on a workload the reference's algorithm
solves (tests/test_oracle_chain.py): depth -> cloud -> voxelDownsample (the reference's container order) ->
estimateNormals(30) -> computeFPFH(5 voxels) -> descriptor match -> RANSAC -> ICP(0.4 voxel), on the GPU

  (a) operator by operator through the host-buffer C ABI, and
  (b) in one call of the batched, device-resident entry point (tdv_register_batch_dev, one depth frame per instance),

against the CPU oracle's chain on the same frames: every intermediate exact (cloud, voxels and their order, normals,
descriptor correspondences, the per-iteration inlier counts, the RANSAC transform), the refined transform within the
north star's tolerance (1e-4 rad, 1e-3 mm) and within 1e-2 rad of the ground truth."""
import numpy as np
import pytest
import torch

import chain_scene as cs

pytestmark = pytest.mark.gpu
N_INST = 3


@pytest.fixture(scope="module")
def scene(orc, synth):
    sc = cs.build(synth, n_instances=N_INST)
    model = cs.oracle_model(orc, sc)
    inst = [cs.oracle_instance(orc, sc, b, model) for b in range(N_INST)]
    return sc, model, inst


def _intr():
    return dict(fx=cs.F, fy=cs.F, cx=cs.CX, cy=cs.CY)


def _gpu_model(ctx, tdv, sc):
    xyz, _ = ctx.depth_to_cloud(sc["model_depth"], sc["model_mask"], None, cs.SCALE, cs.F, cs.F, cs.CX, cs.CY, cs.ZMAX)
    mx, _ = ctx.voxel_downsample(xyz, None, cs.VOXEL, tdv.TDV_VOXEL_ORDER_REFERENCE)
    mn = ctx.estimate_normals(mx, 30)
    mf = ctx.compute_fpfh(mx, mn, cs.VOXEL * 5.0)
    return xyz, mx, mn, mf


def test_stagewise_chain_equals_oracle_chain(ctx, tdv, synth, scene):
    sc, model, inst = scene
    raw, mx, mn, mf = _gpu_model(ctx, tdv, sc)
    assert raw.tobytes() == model["raw"].tobytes()
    assert mx.tobytes() == model["xyz"].tobytes() and mn.tobytes() == model["normals"].tobytes()
    assert mf.tobytes() == model["fpfh"].tobytes()
    for b in range(N_INST):
        o = inst[b]
        xyz, _ = ctx.depth_to_cloud(sc["depth"][b], sc["masks"][b], None, cs.SCALE, cs.F, cs.F, cs.CX, cs.CY, cs.ZMAX)
        assert xyz.tobytes() == o["xyz"].tobytes()
        src, _ = ctx.voxel_downsample(xyz, None, cs.VOXEL, tdv.TDV_VOXEL_ORDER_REFERENCE)
        assert src.tobytes() == o["src"].tobytes()                       # values AND unordered_map order
        nrm = ctx.estimate_normals(src, 30)
        assert nrm.tobytes() == o["normals"].tobytes()
        fp = ctx.compute_fpfh(src, nrm, cs.VOXEL * 5.0)
        same = (fp.view(np.uint32) == o["fpfh"].view(np.uint32)).all(1)
        assert same.all(), (~same).sum()                                 # theta from glibc's atan2f restated on the device: every bin the reference's
        corr = ctx.feature_match(fp, mf)
        assert np.array_equal(corr, o["coarse"]["corr"])
        coarse = ctx.ransac(src, mx, fs=fp, ft=mf, voxel=cs.VOXEL, max_iterations=cs.HYPS, confidence=0.999, trace=True)
        assert np.array_equal(coarse.trace_inliers, o["coarse"]["inliers"])
        assert coarse.best_iteration == o["coarse"]["best_iter"] and coarse.iterations_run == o["coarse"]["iters_run"]
        assert coarse.transformation.tobytes() == o["coarse"]["T"].tobytes() and coarse.fitness == o["coarse"]["fitness"]
        fine = ctx.icp(src, mx, mn, coarse.transformation, cs.VOXEL * 0.4, cs.ICP_ITERS, True)
        ang, tr = synth.pose_error(fine.transformation, o["fine"]["T"])
        assert fine.iterations == o["fine"]["iterations"] and ang <= 1e-4 and tr <= 1e-6, (fine.iterations, o["fine"]["iterations"], ang, tr)
        assert abs(float(fine.fitness) - float(o["fine"]["fitness"])) < 1e-6
        ctx.set_icp_accumulation("reference")                         # the reference's accumulation order: the oracle's bits
        try:
            fine_ref = ctx.icp(src, mx, mn, coarse.transformation, cs.VOXEL * 0.4, cs.ICP_ITERS, True)
        finally:
            ctx.set_icp_accumulation("tree")
        assert fine_ref.transformation.tobytes() == o["fine"]["T"].tobytes() and fine_ref.iterations == o["fine"]["iterations"]
        assert np.float32(fine_ref.rmse).tobytes() == np.float32(o["fine"]["rmse"]).tobytes()
        assert np.float32(fine_ref.fitness).tobytes() == np.float32(o["fine"]["fitness"]).tobytes()
        ang_gt, tr_gt = synth.pose_error(fine.transformation, sc["T_gt"][b])
        print("instance %d: %d voxels, FPFH rows bit-equal %.4f, coarse inliers %d @%d, ICP %d iterations; vs oracle %.1e rad %.1e m; "
              "vs ground truth %.1e rad %.1e m" % (b, len(src), same.mean(), coarse.inliers, coarse.best_iteration, fine.iterations, ang, tr, ang_gt, tr_gt))
        assert ang_gt < 1e-2 and tr_gt < 1e-3


def _batch(ctx, tdv, sc, order, d_model):
    dev = torch.device("cuda", 0)
    d_depth = torch.from_numpy(sc["depth"].view(np.int16)).to(dev)     # one frame per instance
    d_masks = torch.from_numpy(sc["masks"]).to(dev)
    prm = tdv.batch_params(width=cs.W, height=cs.H, scale_to_meters=cs.SCALE, zmax=cs.ZMAX, voxel_size=cs.VOXEL,
                           ransac_max_iterations=cs.HYPS, icp_max_iterations=cs.ICP_ITERS, voxel_order=order, n_frames=N_INST, **_intr())
    d_mx, d_mn, d_mf, nm = d_model
    return ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), N_INST, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)


def _device_model(ctx, tdv, sc, order):
    dev = torch.device("cuda", 0)
    raw, _ = ctx.depth_to_cloud(sc["model_depth"], sc["model_mask"], None, cs.SCALE, cs.F, cs.F, cs.CX, cs.CY, cs.ZMAX)
    d_raw = torch.from_numpy(raw).to(dev)
    d_mx = torch.empty_like(d_raw); d_mn = torch.empty_like(d_raw); d_mf = torch.empty((len(raw), 33), dtype=torch.float32, device=dev)
    nm = ctx.prepare_model_dev(d_raw.data_ptr(), len(raw), cs.VOXEL, 30, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), order=order)
    return d_mx, d_mn, d_mf, nm


def test_batched_chain_equals_oracle_chain(ctx, tdv, synth, scene):
    sc, model, inst = scene
    d_model = _device_model(ctx, tdv, sc, tdv.TDV_VOXEL_ORDER_REFERENCE)
    d_mx, d_mn, d_mf, nm = d_model
    assert nm == len(model["xyz"])
    assert d_mx[:nm].cpu().numpy().tobytes() == model["xyz"].tobytes()
    assert d_mn[:nm].cpu().numpy().tobytes() == model["normals"].tobytes()
    assert d_mf[:nm].cpu().numpy().tobytes() == model["fpfh"].tobytes()
    res = _batch(ctx, tdv, sc, tdv.TDV_VOXEL_ORDER_REFERENCE, d_model)
    for b, r in enumerate(res):
        o = inst[b]
        assert r["status"] == 0 and r["n_points"] == len(o["xyz"]) and r["n_voxels"] == len(o["src"])
        assert r["coarse_inliers"] == int(o["coarse"]["inliers"][o["coarse"]["best_iter"]]) and r["coarse_fitness"] == o["coarse"]["fitness"]
        ang, tr = synth.pose_error(r["T"], o["fine"]["T"])
        assert r["icp_iterations"] == o["fine"]["iterations"] and ang <= 1e-4 and tr <= 1e-6, (ang, tr)
        # and bit for bit what the operator-by-operator chain returns
        fine = ctx.icp(o["src"], model["xyz"], model["normals"], o["coarse"]["T"], cs.VOXEL * 0.4, cs.ICP_ITERS, True)
        assert r["T"].tobytes() == fine.transformation.tobytes() and r["fitness"] == fine.fitness and r["rmse"] == fine.rmse
        ang_gt, tr_gt = synth.pose_error(r["T"], sc["T_gt"][b])
        assert ang_gt < 1e-2 and tr_gt < 1e-3
    # the whole batched chain with the reference's accumulation order: every instance's refined transform IS the oracle's
    ctx.set_icp_accumulation("reference")
    try:
        res_ref = _batch(ctx, tdv, sc, tdv.TDV_VOXEL_ORDER_REFERENCE, d_model)
    finally:
        ctx.set_icp_accumulation("tree")
    for b, r in enumerate(res_ref):
        o = inst[b]["fine"]
        assert r["T"].tobytes() == o["T"].tobytes() and r["icp_iterations"] == o["iterations"]
        assert np.float32(r["rmse"]).tobytes() == np.float32(o["rmse"]).tobytes() and np.float32(r["fitness"]).tobytes() == np.float32(o["fitness"]).tobytes()


def test_batched_chain_first_occurrence_order_registers_too(ctx, tdv, synth, scene):
    """TDV_VOXEL_ORDER_FIRST (no host replay of the reference's container) permutes the downsampled cloud, so RANSAC's
    index stream draws other triples: a different coarse pose, the same registration."""
    sc, model, inst = scene
    res = _batch(ctx, tdv, sc, tdv.TDV_VOXEL_ORDER_FIRST, _device_model(ctx, tdv, sc, tdv.TDV_VOXEL_ORDER_FIRST))
    for b, r in enumerate(res):
        assert r["status"] == 0 and r["n_voxels"] == len(inst[b]["src"])
        ang_gt, tr_gt = synth.pose_error(r["T"], sc["T_gt"][b])
        assert r["coarse_fitness"] > 0.3 and r["fitness"] > 0.4 and ang_gt < 1e-2 and tr_gt < 1e-3, (ang_gt, tr_gt)


def test_instances_sharing_frames(ctx, tdv, synth, scene):
    """The instance -> frame map: explicit array, in any order, equals one frame per instance."""
    sc, model, inst = scene
    dev = torch.device("cuda", 0)
    perm = [2, 0, 1]
    d_depth = torch.from_numpy(sc["depth"][perm].view(np.int16)).to(dev)      # frames stored in another order
    d_masks = torch.from_numpy(sc["masks"]).to(dev)
    fmap = np.argsort(perm).astype(np.int32)                                   # instance b reads frame fmap[b]
    cap = int((sc["masks"] > 0).sum())
    d_xyz = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    off = ctx.depth_to_cloud_batch_dev(d_depth.data_ptr(), d_masks.data_ptr(), None, N_INST, cs.W, cs.H, cs.SCALE, cs.F, cs.F, cs.CX, cs.CY, cs.ZMAX,
                                       d_xyz.data_ptr(), None, cap, n_frames=N_INST, frame_of_instance=fmap)
    xyz = d_xyz.cpu().numpy()
    for b in range(N_INST):
        assert xyz[off[b]:off[b + 1]].tobytes() == inst[b]["xyz"].tobytes()
    with pytest.raises(tdv.TdvError):
        ctx.depth_to_cloud_batch_dev(d_depth.data_ptr(), d_masks.data_ptr(), None, N_INST, cs.W, cs.H, cs.SCALE, cs.F, cs.F, cs.CX, cs.CY, cs.ZMAX,
                                     d_xyz.data_ptr(), None, cap, n_frames=N_INST, frame_of_instance=np.array([0, 1, 3], np.int32))


@pytest.mark.parametrize("k", [30, 120])     # 120 > the radius lists' cap of 100: the batch then runs its stages on the reference-ordered cloud itself
def test_batched_chain_tie_heavy_terraces(ctx, tdv, synth, k):
    """Terraces of constant depth seen head-on: pixels and voxel means sit on lattices, so nearly every neighbour list holds
    runs of exactly equal distances, whose order is decided by the members' positions in the reference's container order.
    The batch computes normals, descriptors and matches on the first-occurrence order with those positions as tie-break ids
    (csrc/batch.hip); it has to return what the operators return on the reference-ordered cloud, bit for bit."""
    dev = torch.device("cuda", 0)
    w, h, f = 256, 192, 500.0
    intr = dict(fx=f, fy=f, cx=w / 2.0, cy=h / 2.0)
    voxel, scale, zmax = 0.004, 1000.0, 1.5
    v, u = np.mgrid[0:h, 0:w]

    def frame(shift, seed):
        steps = ((u + shift) // 24 + 2 * ((v + 2 * shift) // 16)) % 5
        bumps = np.random.default_rng(seed).integers(0, 2, (h // 16 + 2, w // 24 + 2))[(v + 2 * shift) // 16, (u + shift) // 24]
        d = (500 + 6 * steps + 3 * bumps).astype(np.uint16)
        m = np.zeros((h, w), np.uint8); m[8:h - 8, 8:w - 8] = 255
        return d, m

    md, mm = frame(0, 5)
    depth = np.stack([frame(3, 5)[0], frame(7, 5)[0]]); masks = np.stack([mm, mm])
    raw, _ = ctx.depth_to_cloud(md, mm, None, scale, f, f, w / 2.0, h / 2.0, zmax)
    d_raw = torch.from_numpy(raw).to(dev)
    d_mx = torch.empty_like(d_raw); d_mn = torch.empty_like(d_raw); d_mf = torch.empty((len(raw), 33), dtype=torch.float32, device=dev)
    nm = ctx.prepare_model_dev(d_raw.data_ptr(), len(raw), voxel, k, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), order=tdv.TDV_VOXEL_ORDER_REFERENCE)
    mx = d_mx[:nm].cpu().numpy(); mn = d_mn[:nm].cpu().numpy(); mf = d_mf[:nm].cpu().numpy()
    prm = tdv.batch_params(width=w, height=h, scale_to_meters=scale, zmax=zmax, voxel_size=voxel, normals_k=k, ransac_max_iterations=1500,
                           icp_max_iterations=10, voxel_order=tdv.TDV_VOXEL_ORDER_REFERENCE, n_frames=2, **intr)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev); d_masks = torch.from_numpy(masks).to(dev)
    res = ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), 2, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    tied = 0
    for b, r in enumerate(res):
        xyz, _ = ctx.depth_to_cloud(depth[b], masks[b], None, scale, f, f, w / 2.0, h / 2.0, zmax)
        src, _ = ctx.voxel_downsample(xyz, None, voxel, tdv.TDV_VOXEL_ORDER_REFERENCE)
        nrm, knn = ctx.estimate_normals(src, k, want_knn=True)
        d2 = ((src[knn] - src[:, None, :]) ** 2).sum(2)
        tied += int((np.diff(d2, axis=1) == 0).sum())
        fp = ctx.compute_fpfh(src, nrm, voxel * 5.0)
        coarse = ctx.ransac(src, mx, fs=fp, ft=mf, voxel=voxel, max_iterations=1500, confidence=0.999)
        fine = ctx.icp(src, mx, mn, coarse.transformation, voxel * 0.4, 10, True)
        assert r["status"] == 0 and r["n_voxels"] == len(src)
        assert r["coarse_inliers"] == coarse.inliers and r["coarse_fitness"] == coarse.fitness
        assert r["T"].tobytes() == fine.transformation.tobytes() and r["fitness"] == fine.fitness and r["rmse"] == fine.rmse
    assert tied > 1000, tied      # the scene does what it was built for
