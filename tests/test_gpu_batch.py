"""the batched, device-resident Pipeline::processInstance (tdv_register_batch_dev) must return
exactly what the same chain gives when each operator is called on its own through the host-buffer ABI
(which the other test modules pin against the oracle), instance by instance — here in first-occurrence voxel order on
the cuboid; tests/test_gpu_chain.py compares the batch in the reference's order with the oracle's whole chain."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(synth, orc, n_inst=3, w=640, h=480):
    """A depth frame showing n_inst copies of the synthetic object at different poses, one mask each."""
    f = 600.0
    cx, cy = w / 2.0, h / 2.0
    depth = np.zeros((h, w), np.uint16)
    masks = np.zeros((n_inst, h, w), np.uint8)
    model, _ = synth.sample_object(60000, 42)
    for b in range(n_inst):
        T = synth.make_transform([0.3 + 0.2 * b, 1.0, 0.4 - 0.3 * b], 25.0 + 10 * b, (-0.15 + 0.15 * b, 0.02 * b, 0.55 + 0.03 * b))
        p = model.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3]  # object pose in the camera frame
        u = np.round(p[:, 0] / p[:, 2] * f + cx).astype(int); v = np.round(p[:, 1] / p[:, 2] * f + cy).astype(int)
        ok = (u >= 0) & (u < w) & (v >= 0) & (v < h) & (p[:, 2] > 0)
        z = np.full((h, w), np.inf)
        np.minimum.at(z, (v[ok], u[ok]), p[ok, 2])  # z-buffer: nearest surface per pixel
        hit = np.isfinite(z)
        depth[hit] = np.round(z[hit] * 1000.0).astype(np.uint16)
        masks[b][hit] = 255
    return depth, masks, dict(fx=f, fy=f, cx=cx, cy=cy, width=w, height=h)


def test_batch_equals_stagewise_chain(ctx, tdv, synth, orc):
    depth, masks, intr = _scene(synth, orc)
    voxel = 0.004
    prm = tdv.batch_params(voxel_size=voxel, zmax=1.5, ransac_max_iterations=4000, icp_max_iterations=30, voxel_order=tdv.TDV_VOXEL_ORDER_FIRST, **intr)
    dev = torch.device("cuda", 0)
    # model: prepared once on the device (voxel -> normals -> FPFH), as Pipeline::run does
    model_raw, _ = synth.sample_object(20000, 7)
    d_model_raw = torch.from_numpy(model_raw).to(dev)
    d_mx = torch.empty_like(d_model_raw); d_mn = torch.empty_like(d_model_raw)
    d_mf = torch.empty((len(model_raw), 33), dtype=torch.float32, device=dev)
    nm = ctx.prepare_model_dev(d_model_raw.data_ptr(), len(model_raw), voxel, 30, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), order=tdv.TDV_VOXEL_ORDER_FIRST)
    mx = d_mx[:nm].cpu().numpy(); mn = d_mn[:nm].cpu().numpy(); mf = d_mf[:nm].cpu().numpy()
    # the model prep itself equals the stagewise ops
    ex, _ = ctx.voxel_downsample(model_raw, None, voxel, tdv.TDV_VOXEL_ORDER_FIRST)
    assert nm == len(ex) and mx.tobytes() == ex.tobytes()
    en = ctx.estimate_normals(ex, 30)
    assert mn.tobytes() == en.tobytes()
    assert mf.tobytes() == ctx.compute_fpfh(ex, en, voxel * 5.0).tobytes()

    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_masks = torch.from_numpy(masks).to(dev)
    res = ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), len(masks), prm,
                                 d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    assert len(res) == len(masks)
    for b, r in enumerate(res):
        xyz, _ = ctx.depth_to_cloud(depth, masks[b], None, 1000.0, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 1.5)
        assert r["status"] == 0 and r["n_points"] == len(xyz) > 1000
        src, _ = ctx.voxel_downsample(xyz, None, voxel, tdv.TDV_VOXEL_ORDER_FIRST)
        assert r["n_voxels"] == len(src)
        nrm = ctx.estimate_normals(src, 30)
        fp = ctx.compute_fpfh(src, nrm, voxel * 5.0)
        coarse = ctx.ransac(src, mx, fs=fp, ft=mf, voxel=voxel, max_iterations=4000, confidence=0.999)
        assert r["coarse_inliers"] == coarse.inliers and r["coarse_fitness"] == coarse.fitness
        fine = ctx.icp(src, mx, mn, coarse.transformation, voxel * 0.4, 30, True)
        assert r["icp_iterations"] == fine.iterations
        assert r["T"].tobytes() == fine.transformation.tobytes() and r["fitness"] == fine.fitness and r["rmse"] == fine.rmse
        print("instance %d: %d px -> %d voxels, coarse inliers %d, icp fitness %.3f" % (b, r["n_points"], r["n_voxels"], r["coarse_inliers"], r["fitness"]))


def test_batch_empty_mask(ctx, tdv, synth, orc):
    depth, masks, intr = _scene(synth, orc, n_inst=1)
    masks = np.concatenate([np.zeros_like(masks), masks], 0)  # first instance: empty mask
    dev = torch.device("cuda", 0)
    model, nrm = synth.sample_object(3000, 7)
    d_mx = torch.from_numpy(model).to(dev); d_mn = torch.from_numpy(nrm).to(dev)
    d_mf = torch.from_numpy(synth.random_features(3000, 1)).to(dev)
    prm = tdv.batch_params(voxel_size=0.006, ransac_max_iterations=500, icp_max_iterations=5, **intr)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)  # keep alive across the call
    d_masks = torch.from_numpy(masks).to(dev)
    res = ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), 2, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), 3000)
    assert res[0]["status"] == 1 and res[0]["n_points"] == 0 and np.array_equal(res[0]["T"], np.eye(4, dtype=np.float32))   # empty depth after masking (pipeline.cpp:57-60)
    assert res[1]["status"] == 0 and res[1]["n_voxels"] > 100


def test_all_instance_clouds_in_one_pass(ctx, tdv, synth, orc):
    """tdv_depth_to_cloud_batch_dev: stacked masks and the label-image format give, per instance, exactly the
    cloud (values and row-major order) of the single-instance entry point."""
    depth, masks, intr = _scene(synth, orc, n_inst=3)
    h, w = depth.shape
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev); d_masks = torch.from_numpy(masks).to(dev); d_bgr = torch.from_numpy(bgr).to(dev)
    cap = int(sum((m > 0).sum() for m in masks))
    d_xyz = torch.zeros((cap, 3), dtype=torch.float32, device=dev); d_rgb = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    off = ctx.depth_to_cloud_batch_dev(d_depth.data_ptr(), d_masks.data_ptr(), d_bgr.data_ptr(), 3, w, h, 1000.0, intr["fx"], intr["fy"],
                                       intr["cx"], intr["cy"], 1.5, d_xyz.data_ptr(), d_rgb.data_ptr(), cap)
    xyz = d_xyz.cpu().numpy(); rgb = d_rgb.cpu().numpy()
    label = np.zeros((h, w), np.uint8)
    for b in range(3):
        ref_xyz, ref_rgb = ctx.depth_to_cloud(depth, masks[b], bgr, 1000.0, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 1.5)
        assert off[b + 1] - off[b] == len(ref_xyz) > 0
        assert xyz[off[b]:off[b + 1]].tobytes() == ref_xyz.tobytes() and rgb[off[b]:off[b + 1]].tobytes() == ref_rgb.tobytes()
        label[(masks[b] > 0) & (label == 0)] = b + 1
    # label image: instance b = pixels with value b + 1 (overlaps resolved towards the lower label above)
    d_label = torch.from_numpy(label).to(dev)
    off2 = ctx.depth_to_cloud_batch_dev(d_depth.data_ptr(), d_label.data_ptr(), None, 3, w, h, 1000.0, intr["fx"], intr["fy"],
                                        intr["cx"], intr["cy"], 1.5, d_xyz.data_ptr(), None, cap, mask_format=1)
    xyz2 = d_xyz.cpu().numpy()
    for b in range(3):
        ref_xyz, _ = ctx.depth_to_cloud(depth, np.where(label == b + 1, 255, 0).astype(np.uint8), None, 1000.0, intr["fx"], intr["fy"],
                                        intr["cx"], intr["cy"], 1.5)
        assert off2[b + 1] - off2[b] == len(ref_xyz) and xyz2[off2[b]:off2[b + 1]].tobytes() == ref_xyz.tobytes()
    with pytest.raises(tdv.TdvError):
        ctx.depth_to_cloud_batch_dev(d_depth.data_ptr(), d_masks.data_ptr(), None, 3, w, h, 1000.0, intr["fx"], intr["fy"], intr["cx"], intr["cy"],
                                     1.5, d_xyz.data_ptr(), None, 10)


def test_all_instance_clouds_odd_frame_scalar_path(ctx, orc):
    """A frame whose pixel count is not a multiple of 16 takes the non-vectorised kernels; same results."""
    rng = np.random.default_rng(4)
    h, w, B = 97, 131, 4
    raw = rng.integers(300, 1400, (h, w)).astype(np.uint16)
    masks = (rng.random((B, h, w)) < 0.4).astype(np.uint8) * 255
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_masks = torch.from_numpy(masks).to(dev); d_bgr = torch.from_numpy(bgr).to(dev)
    cap = B * h * w
    d_xyz = torch.zeros((cap, 3), dtype=torch.float32, device=dev); d_rgb = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    off = ctx.depth_to_cloud_batch_dev(d_raw.data_ptr(), d_masks.data_ptr(), d_bgr.data_ptr(), B, w, h, 1000.0, 500, 500, 65, 48, 1.2,
                                       d_xyz.data_ptr(), d_rgb.data_ptr(), cap)
    xyz = d_xyz.cpu().numpy(); rgb = d_rgb.cpu().numpy()
    for b in range(B):
        ref_xyz, ref_rgb = orc.unproject(orc.depth_preprocess(raw, masks[b], 1000.0), bgr, 500, 500, 65, 48, 1.2)
        assert xyz[off[b]:off[b + 1]].tobytes() == ref_xyz.tobytes() and rgb[off[b]:off[b + 1]].tobytes() == ref_rgb.tobytes()


@pytest.mark.parametrize("seed", range(6))
def test_all_instance_clouds_camera_parameter_fuzz(ctx, orc, seed):
    """The tiled kernels (pixel count a multiple of 16) against the oracle, bit for bit, over camera parameters: the emit
    pass divides by the wave-uniform focal lengths with a hoisted reciprocal when the parameters allow it and with `/`
    otherwise (seeds 4, 5: focal length resp. principal point outside the allowed ranges); the depth test is a raw-value
    range found on the host; masks go through the four-bytes-at-once rule in all three mask modes' byte ranges."""
    rng = np.random.default_rng(100 + seed)
    h, w, B = 48, 176, 9                      # 8,448 pixels = 8.25 tiles: a ragged last tile and a ragged last instance group
    raw = rng.integers(0, 3000, (h, w)).astype(np.uint16)
    raw[rng.random((h, w)) < 0.05] = 0
    masks = rng.integers(0, 256, (B, h, w)).astype(np.uint8)
    masks[rng.random((B, h, w)) < 0.5] = 0
    masks[:, :, :32][:, 10:20] = 11           # around the threshold of the CPU rule (> 10)
    masks[:, :, 32:64][:, 10:20] = 10
    masks[B - 1] = 0                          # an instance without pixels
    scale = float(rng.choice([1000.0, 4000.0, 0.25, 65535.0]))
    fx = float(np.float32(rng.uniform(0.3, 3000.0))); fy = float(np.float32(rng.uniform(0.3, 3000.0)))
    cx = float(np.float32(rng.uniform(-50.0, w + 50.0))); cy = float(np.float32(rng.choice([0.0, 23.5, h / 2 - 0.123])))
    if seed % 2: fx = -fx
    if seed == 4: fx = 3.0e-5
    if seed == 5: cx = 1.0e-6
    zmax = float(np.float32(rng.uniform(0.2, 2.5) * 1000.0 / scale))
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_masks = torch.from_numpy(masks).to(dev)
    cap = B * h * w
    d_xyz = torch.zeros((cap + 1, 3), dtype=torch.float32, device=dev)
    for shift in (0, 1):                      # output base 16-B aligned and not
        out_ptr = d_xyz.data_ptr() + 4 * shift
        off = ctx.depth_to_cloud_batch_dev(d_raw.data_ptr(), d_masks.data_ptr(), None, B, w, h, scale, fx, fy, cx, cy, zmax, out_ptr, None, cap)
        flat = d_xyz.cpu().numpy().reshape(-1)[shift:]
        assert off[B] - off[B - 1] == 0
        for b in range(B):
            ref_xyz, _ = orc.unproject(orc.depth_preprocess(raw, masks[b], scale), None, fx, fy, cx, cy, zmax)
            assert off[b + 1] - off[b] == len(ref_xyz)
            assert flat[3 * off[b]:3 * off[b + 1]].tobytes() == ref_xyz.tobytes(), (seed, b, shift)


def test_all_instance_clouds_mask_rules_every_byte_value(ctx, orc, tdv):
    """The four-bytes-at-once mask rules of the tiled pass against per-pixel numpy: `> 10`, `!= 0` and `== label` over
    every byte value (0..255 side by side in every row), labels 1..255 (label = instance + 1: 255 instances of one
    label image), and one depth frame per instance chosen through frame_of_instance."""
    rng = np.random.default_rng(11)
    h, w = 32, 256                                     # 8,192 pixels: 8 tiles
    raw = rng.integers(1, 2000, (3, h, w)).astype(np.uint16)     # three frames
    raw[:, ::7, ::5] = 0
    ramp = np.tile(np.arange(256, dtype=np.uint8), (h, 1))       # every byte value in every row
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw.view(np.int16)).to(dev)
    fx, fy, cx, cy, zmax, scale = 300.0, 310.0, 128.5, 15.5, 1.7, 1000.0

    def run(masks_np, n_inst, mask_format, mask_mode, n_frames, fmap):
        d_masks = torch.from_numpy(masks_np).to(dev)
        cap = n_inst * h * w
        d_xyz = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
        off = ctx.depth_to_cloud_batch_dev(d_raw.data_ptr(), d_masks.data_ptr(), None, n_inst, w, h, scale, fx, fy, cx, cy, zmax, d_xyz.data_ptr(), None, cap,
                                           mask_format=mask_format, mask_mode=mask_mode, n_frames=n_frames, frame_of_instance=fmap)
        return off, d_xyz.cpu().numpy()

    def expect(frame, keep):
        depth = raw[frame].astype(np.float32) * np.float32(1.0 / scale)
        depth = np.where(keep, depth, np.float32(0)).astype(np.float32)
        return orc.unproject(depth, None, fx, fy, cx, cy, zmax)[0]

    # stacked masks, both threshold rules, frames 0 / 1 / 2 / 1 / 0
    stacked = np.stack([np.roll(ramp, 17 * b, axis=1) for b in range(5)])
    fmap = np.array([0, 1, 2, 1, 0], np.int32)
    for mode, rule in ((tdv.TDV_MASK_THRESHOLD10, lambda m: m > 10), (tdv.TDV_MASK_NONZERO, lambda m: m != 0)):
        off, xyz = run(stacked, 5, 0, mode, 3, fmap)
        for b in range(5):
            ref = expect(fmap[b], rule(stacked[b]))
            assert off[b + 1] - off[b] == len(ref) and xyz[off[b]:off[b + 1]].tobytes() == ref.tobytes(), (mode, b)
    # one label image, 255 instances: instance b keeps the pixels whose value is b + 1 (value 0 belongs to nobody)
    off, xyz = run(ramp, 255, 1, tdv.TDV_MASK_THRESHOLD10, 1, None)
    for b in (0, 1, 9, 10, 126, 127, 128, 199, 253, 254):
        ref = expect(0, ramp == b + 1)
        assert off[b + 1] - off[b] == len(ref) > 0 and xyz[off[b]:off[b + 1]].tobytes() == ref.tobytes(), b
    assert off[255] == sum(len(expect(0, ramp == v)) for v in range(1, 256))


@pytest.mark.parametrize("n,voxel,k", [(3, 0.01, 30), (40, 0.002, 30), (200, 0.004, 30), (3000, 0.002, 8), (5000, 0.01, 64)])
def test_model_prep_small_clouds(ctx, tdv, synth, n, voxel, k):
    """tdv_prepare_model_dev (one radius search shared by normals and FPFH, kNN only for the deficient points) equals
    the three stagewise operators on clouds smaller than a leaf, smaller than k, and with k above 32."""
    dev = torch.device("cuda", 0)
    raw, _ = synth.sample_object(n, 13)
    d_raw = torch.from_numpy(raw).to(dev)
    d_mx = torch.empty_like(d_raw); d_mn = torch.empty_like(d_raw); d_mf = torch.empty((n, 33), dtype=torch.float32, device=dev)
    nm = ctx.prepare_model_dev(d_raw.data_ptr(), n, voxel, k, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), order=tdv.TDV_VOXEL_ORDER_FIRST)
    ex, _ = ctx.voxel_downsample(raw, None, voxel, tdv.TDV_VOXEL_ORDER_FIRST)
    assert nm == len(ex) and d_mx[:nm].cpu().numpy().tobytes() == ex.tobytes()
    en = ctx.estimate_normals(ex, k)
    assert d_mn[:nm].cpu().numpy().tobytes() == en.tobytes()
    assert d_mf[:nm].cpu().numpy().tobytes() == ctx.compute_fpfh(ex, en, voxel * 5.0).tobytes()


def test_batch_status_1_vs_2(ctx, tdv, synth, orc):
    """src/pipeline.cpp:57-60 vs :86-89: an instance whose masked depth holds no non-zero value ends with status 1 ("empty
    depth after masking"); one that has depth under its mask but nothing inside 0 < z <= zmax ends with status 2 ("empty
    point cloud").  The oracle's countNonZero / unproject on the same inputs draw the same line."""
    depth, masks, intr = _scene(synth, orc, n_inst=1)
    h, w = depth.shape
    far = depth.copy(); far[:40, :40] = 60000                        # 60 m: beyond zmax = 1.5
    m_empty = np.zeros((h, w), np.uint8)                             # masks everything away
    m_zero_depth = np.zeros((h, w), np.uint8); m_zero_depth[h - 8:, w - 8:] = 255   # keeps only pixels whose raw depth is 0
    assert (far[h - 8:, w - 8:] == 0).all()
    m_far = np.zeros((h, w), np.uint8); m_far[:40, :40] = 255         # keeps only pixels beyond the clip
    m_weak = np.where(masks[0] > 0, 10, 0).astype(np.uint8)          # mask value 10 is not > 10: everything rejected
    allm = np.stack([m_empty, m_zero_depth, masks[0], m_far, m_weak])    # empty instances before AND after a real one (offsets of empty clouds)
    want = []
    for m in allm:
        sd = orc.depth_preprocess(far, m, 1000.0)
        if orc.count_nonzero(sd) == 0: want.append(1)
        elif len(orc.unproject(sd, None, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 1.5)[0]) == 0: want.append(2)
        else: want.append(0)
    assert want == [1, 1, 0, 2, 1]
    dev = torch.device("cuda", 0)
    model, nrm = synth.sample_object(3000, 7)
    d_mx = torch.from_numpy(model).to(dev); d_mn = torch.from_numpy(nrm).to(dev)
    d_mf = torch.from_numpy(synth.random_features(3000, 1)).to(dev)
    prm = tdv.batch_params(voxel_size=0.006, ransac_max_iterations=500, icp_max_iterations=5, **intr)
    d_depth = torch.from_numpy(far.view(np.int16)).to(dev)
    d_masks = torch.from_numpy(allm).to(dev)
    res = ctx.register_batch_dev(d_depth.data_ptr(), None, d_masks.data_ptr(), len(allm), prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), 3000)
    assert [r["status"] for r in res] == want
    assert all(r["n_points"] == 0 and np.array_equal(r["T"], np.eye(4, dtype=np.float32)) for i, r in enumerate(res) if i != 2)
    assert res[2]["n_voxels"] > 100


def test_mask_resize_nearest_matches_oracle(ctx, orc):
    """cv::resize(mask, ..., INTER_NEAREST) of src/pipeline.cpp:38-41: tdv_mask_resize_nearest vs the oracle's restatement of
    OpenCV's resizeNN, up- and down-scaling, non-integer ratios, 1-pixel sources, several masks per call."""
    rng = np.random.default_rng(11)
    for (sh, sw), (dh, dw) in [((480, 640), (720, 1280)), ((720, 1280), (480, 640)), ((37, 53), (720, 1280)), ((1, 1), (9, 7)),
                               ((200, 300), (200, 300)), ((719, 1279), (720, 1280)), ((3, 1000), (97, 131))]:
        m = rng.integers(0, 256, (3, sh, sw)).astype(np.uint8)
        got = ctx.mask_resize_nearest(m, dw, dh)
        for b in range(3):
            assert got[b].tobytes() == orc.mask_resize_nearest(m[b], dw, dh).tobytes(), ((sh, sw), (dh, dw))


def test_batch_resizes_mismatched_masks_and_takes_u16_labels(ctx, tdv, synth, orc):
    """Masks of another size than the frame are resized first (pipeline.cpp:38-41): the batch on half-size masks equals the
    batch on the oracle-resized full-size masks; a u16 label image (mask_format 2) equals the stacked masks it encodes."""
    depth, masks, intr = _scene(synth, orc, n_inst=3)
    h, w = depth.shape
    small = masks[:, ::2, ::2].copy()                                  # 240 x 320 masks for a 480 x 640 frame
    full = np.stack([orc.mask_resize_nearest(m, w, h) for m in small])
    label = np.zeros((h, w), np.uint16)
    for b in range(3):
        label[(full[b] > 10) & (label == 0)] = b + 1
    stacked_from_label = np.stack([np.where(label == b + 1, 255, 0).astype(np.uint8) for b in range(3)])
    dev = torch.device("cuda", 0)
    model, nrm = synth.sample_object(3000, 7)
    d_mx = torch.from_numpy(model).to(dev); d_mn = torch.from_numpy(nrm).to(dev)
    d_mf = torch.from_numpy(synth.random_features(3000, 1)).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    common = dict(voxel_size=0.006, ransac_max_iterations=800, icp_max_iterations=8, **intr)

    def run(m, **kw):
        d_m = torch.from_numpy(m).to(dev)
        return ctx.register_batch_dev(d_depth.data_ptr(), None, d_m.data_ptr(), 3, tdv.batch_params(**common, **kw), d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), 3000)

    def same(a, b):
        return all(x["status"] == y["status"] == 0 and x["n_points"] == y["n_points"] > 500 and x["n_voxels"] == y["n_voxels"] and
                   x["T"].tobytes() == y["T"].tobytes() and x["coarse_inliers"] == y["coarse_inliers"] for x, y in zip(a, b))
    assert same(run(small, mask_width=w // 2, mask_height=h // 2), run(full))
    assert same(run(label.view(np.int16), mask_format=2), run(stacked_from_label))
    with pytest.raises(tdv.TdvError):
        run(label.view(np.int16), mask_format=2, mask_width=w // 2, mask_height=h // 2)
