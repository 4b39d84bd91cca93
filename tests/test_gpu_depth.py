"""R1/R2 parity: depth scale+mask and unprojection through the C ABI vs the CPU oracle
(reference src/pipeline.cpp:46-54, :61-84).  Bar: bit-exact values AND row-major order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_frame(seed, h=97, w=131):
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 3000, (h, w)).astype(np.uint16)
    raw[rng.random((h, w)) < 0.2] = 0
    mask = rng.choice(np.array([0, 1, 5, 10, 11, 12, 128, 255], np.uint8), (h, w))
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    return raw, mask, bgr


@pytest.mark.parametrize("seed", [0, 1])
def test_depth_preprocess_bit_exact(ctx, orc, tdv, seed):
    raw, mask, _ = _random_frame(seed)
    for scale in (1000.0, 999.0, 4000.0):
        got = ctx.depth_preprocess(raw, mask, scale)
        ref = orc.depth_preprocess(raw, mask, scale)
        assert got.tobytes() == ref.tobytes()
        got = ctx.depth_preprocess(raw, None, scale)
        assert got.tobytes() == orc.depth_preprocess(raw, None, scale).tobytes()
    # reference CUDA semantics (mask != 0) are selectable and differ only for mask values 1..10
    got = ctx.depth_preprocess(raw, mask, 1000.0, tdv.TDV_MASK_NONZERO)
    ref = raw.astype(np.float32) * np.float32(1.0 / 1000.0)
    ref[mask == 0] = 0
    assert got.tobytes() == ref.tobytes()


def test_depth_preprocess_odd_sizes(ctx, orc):
    for (h, w) in [(1, 1), (1, 3), (7, 5), (33, 1023), (720, 1280)]:
        rng = np.random.default_rng(h * 10007 + w)
        raw = rng.integers(0, 65536, (h, w)).astype(np.uint16)
        mask = rng.integers(0, 256, (h, w)).astype(np.uint8)
        assert ctx.depth_preprocess(raw, mask, 1000.0).tobytes() == orc.depth_preprocess(raw, mask, 1000.0).tobytes()


def test_demo_scene_cloud(ctx, orc):
    """Config C1 (demo scene, src/pipeline.cpp:211-257): 40,401 points in row-major order."""
    depth, bgr = orc.demo_scene()
    mask = orc.demo_mask()
    d_ref = orc.depth_preprocess(depth, mask, 1000.0)
    xyz_ref, rgb_ref = orc.unproject(d_ref, bgr, 900, 900, 640, 360, 1.5)
    assert len(xyz_ref) == 40401
    d = ctx.depth_preprocess(depth, mask, 1000.0)
    assert d.tobytes() == d_ref.tobytes()
    xyz, rgb = ctx.deproject(d, bgr, 900, 900, 640, 360, 1.5)
    assert xyz.tobytes() == xyz_ref.tobytes() and rgb.tobytes() == rgb_ref.tobytes()
    xyz2, rgb2 = ctx.depth_to_cloud(depth, mask, bgr, 1000.0, 900, 900, 640, 360, 1.5)
    assert xyz2.tobytes() == xyz_ref.tobytes() and rgb2.tobytes() == rgb_ref.tobytes()


@pytest.mark.parametrize("seed", [3, 4])
def test_deproject_random(ctx, orc, seed):
    raw, mask, bgr = _random_frame(seed, 211, 307)
    d = orc.depth_preprocess(raw, mask, 1000.0)
    for zmax in (1.5, 10.0, 0.5):
        ref_xyz, ref_rgb = orc.unproject(d, bgr, 611.5, 609.25, 153.2, 101.7, zmax)
        xyz, rgb = ctx.deproject(d, bgr, 611.5, 609.25, 153.2, 101.7, zmax)
        assert xyz.tobytes() == ref_xyz.tobytes() and rgb.tobytes() == ref_rgb.tobytes()
        xyz, rgb = ctx.depth_to_cloud(raw, mask, bgr, 1000.0, 611.5, 609.25, 153.2, 101.7, zmax)
        assert xyz.tobytes() == ref_xyz.tobytes() and rgb.tobytes() == ref_rgb.tobytes()
    # no colour image -> no colours (CPU branch semantics)
    xyz, rgb = ctx.deproject(d, None, 611.5, 609.25, 153.2, 101.7, 1.5)
    ref_xyz, _ = orc.unproject(d, None, 611.5, 609.25, 153.2, 101.7, 1.5)
    assert rgb is None and xyz.tobytes() == ref_xyz.tobytes()


def test_deproject_empty_and_capacity(ctx, orc, tdv):
    d = np.zeros((16, 16), np.float32)
    xyz, _ = ctx.deproject(d, None, 100, 100, 8, 8, 1.5)
    assert len(xyz) == 0
    d[:] = 1.0
    with pytest.raises(tdv.TdvError):
        ctx.deproject(d, None, 100, 100, 8, 8, 1.5, capacity=10)
    # NaN depth is kept by the reference's test (z<=0 || z>max is false for NaN)
    d[3, 4] = np.nan
    xyz, _ = ctx.deproject(d, None, 100, 100, 8, 8, 1.5)
    ref, _ = orc.unproject(d, None, 100, 100, 8, 8, 1.5)
    assert len(xyz) == 256 and np.array_equal(np.isnan(xyz), np.isnan(ref))


def test_bilateral_filter(ctx, orc):
    """cuda/depth_processing.cu:62-155 restated; only expf may differ in the last bits."""
    rng = np.random.default_rng(5)
    d = (0.8 + 0.05 * rng.random((123, 211))).astype(np.float32)
    d[rng.random(d.shape) < 0.15] = 0.0
    for ss, sr in ((1.5, 0.02), (0.6, 0.01), (4.0, 0.05)):  # radius 3, 1, clamped 5
        got = ctx.bilateral_filter(d, ss, sr)
        ref = orc.bilateral_filter(d, ss, sr)
        assert np.array_equal(got == 0, ref == 0) and np.array_equal(got == 0, d == 0)
        assert np.abs(got - ref).max() <= 2e-6
    # known answers that need no oracle: a constant image is a fixed point (the weights cancel); an isolated pixel keeps its value
    flat = np.full((70, 300), 0.75, np.float32)
    assert np.abs(ctx.bilateral_filter(flat, 1.5, 0.02) - 0.75).max() <= 1e-6
    lone = np.zeros((40, 130), np.float32); lone[20, 64] = 0.9; lone[0, 0] = 0.5; lone[39, 129] = 0.6
    out = ctx.bilateral_filter(lone, 2.0, 0.01)
    assert out[20, 64] == np.float32(0.9) and out[0, 0] == np.float32(0.5) and out[39, 129] == np.float32(0.6) and np.count_nonzero(out) == 3


def test_label_image_masks(ctx, orc, tdv):
    """one u8 label image instead of one full-frame mask per instance."""
    raw, _, bgr = _random_frame(9, 97, 131)
    labels = np.random.default_rng(1).integers(0, 4, raw.shape).astype(np.uint8)
    for lab in (1, 2, 3):
        binmask = np.where(labels == lab, 255, 0).astype(np.uint8)
        ref_xyz, ref_rgb = ctx.depth_to_cloud(raw, binmask, bgr, 1000.0, 600, 600, 65, 48, 10.0)
        xyz, rgb = ctx.depth_to_cloud(raw, labels, bgr, 1000.0, 600, 600, 65, 48, 10.0, tdv.TDV_MASK_LABEL_BASE + lab)
        assert xyz.tobytes() == ref_xyz.tobytes() and rgb.tobytes() == ref_rgb.tobytes()
        assert ctx.depth_preprocess(raw, labels, 1000.0, tdv.TDV_MASK_LABEL_BASE + lab).tobytes() == orc.depth_preprocess(raw, binmask, 1000.0).tobytes()


def _numpy_cloud(raw, mask, scale, fx, fy, cx, cy, zmax):
    """src/pipeline.cpp:46-54,61-84 in numpy float32 (the oracle's loop is slow at megapixels): same operations, same order."""
    z = raw.astype(np.float32) * np.float32(1.0 / scale)
    if mask is not None:
        z[mask <= 10] = 0
    keep = ~((z <= 0) | (z > np.float32(zmax)))
    v, u = np.nonzero(keep)
    zz = z[keep]
    x = (u.astype(np.float32) - np.float32(cx)) * zz / np.float32(fx)
    y = (v.astype(np.float32) - np.float32(cy)) * zz / np.float32(fy)
    return np.stack([x, y, zz], axis=1)


def test_one_launch_chain_many_tiles_and_repeated_calls(ctx, orc):
    """k_depth_cloud_chain (round 4): tiles of 4,096 pixels chained through persistent status words.  Frames of 1 tile, 225 tiles
    (one look-back round), ~2,000 tiles (several rounds, inclusive prefixes of finished tiles cut the walk), a ragged last tile; the same
    ctx alternates sizes, so words of earlier calls (older epochs) lie in the array; results equal numpy's and, at the small size, the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    sizes = [(16, 16), (720, 1280), (2160, 3840), (97, 131), (2161, 3839), (720, 1280), (1, 1), (64, 64), (2160, 3840)]
    for rep, (h, w) in enumerate(sizes):
        rng = np.random.default_rng(1000 + rep)
        raw = rng.integers(0, 3000, (h, w)).astype(np.uint16)
        raw[rng.random((h, w)) < (0.1 + 0.2 * (rep % 4))] = 0
        mask = rng.choice(np.array([0, 5, 10, 11, 200], np.uint8), (h, w), p=[0.2, 0.1, 0.1, 0.3, 0.3])
        if rep == 2:
            mask[: h // 2] = 0                                                   # hundreds of tiles without a single point
        ref = _numpy_cloud(raw, mask, 1000.0, 611.5, 609.25, w / 2.0, h / 2.0, 2.5)
        d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_mask = torch.from_numpy(mask).to(dev)
        d_xyz = torch.full((h * w, 3), -7.0, dtype=torch.float32, device=dev)
        n = ctx.depth_to_cloud_dev(d_raw.data_ptr(), d_mask.data_ptr(), None, w, h, 1000.0, 611.5, 609.25, w / 2.0, h / 2.0, 2.5, d_xyz.data_ptr(), None, h * w)
        assert n == len(ref), (h, w, n, len(ref))
        assert d_xyz[:n].cpu().numpy().tobytes() == ref.astype(np.float32).tobytes(), (h, w)
        assert bool((d_xyz[n:] == -7.0).all())                                   # nothing written past the count
        if h * w < 20000:
            oxyz, _ = orc.unproject(orc.depth_preprocess(raw, mask, 1000.0), None, 611.5, 609.25, w / 2.0, h / 2.0, 2.5)
            assert oxyz.tobytes() == ref.tobytes()


def test_one_launch_chain_unaligned_buffers_and_small_capacity(ctx, tdv):
    """Pointers that do not allow the 8-byte depth / 4-byte mask loads take the scalar loads; a capacity below the count returns the count needed."""
    import torch
    dev = torch.device("cuda", 0)
    h, w = 301, 517
    rng = np.random.default_rng(77)
    raw = rng.integers(1, 2000, (h, w)).astype(np.uint16); mask = rng.choice(np.array([0, 255], np.uint8), (h, w))
    ref = _numpy_cloud(raw, mask, 1000.0, 500.0, 500.0, 250.0, 150.0, 1.5)
    buf_r = torch.zeros(h * w + 8, dtype=torch.int16, device=dev); buf_m = torch.zeros(h * w + 8, dtype=torch.uint8, device=dev)
    for shift_r, shift_m in ((1, 0), (0, 1), (3, 3), (0, 0)):
        buf_r[shift_r:shift_r + h * w] = torch.from_numpy(raw.view(np.int16).ravel()).to(dev); buf_m[shift_m:shift_m + h * w] = torch.from_numpy(mask.ravel()).to(dev)
        d_xyz = torch.empty((h * w, 3), dtype=torch.float32, device=dev)
        n = ctx.depth_to_cloud_dev(buf_r.data_ptr() + 2 * shift_r, buf_m.data_ptr() + shift_m, None, w, h, 1000.0, 500.0, 500.0, 250.0, 150.0, 1.5, d_xyz.data_ptr(), None, h * w)
        assert n == len(ref) and d_xyz[:n].cpu().numpy().tobytes() == ref.tobytes(), (shift_r, shift_m)
    with pytest.raises(tdv.TdvError):
        ctx.depth_to_cloud_dev(buf_r.data_ptr(), buf_m.data_ptr(), None, w, h, 1000.0, 500.0, 500.0, 250.0, 150.0, 1.5, d_xyz.data_ptr(), None, len(ref) - 1)
    n = ctx.depth_to_cloud_dev(buf_r.data_ptr(), buf_m.data_ptr(), None, w, h, 1000.0, 500.0, 500.0, 250.0, 150.0, 1.5, d_xyz.data_ptr(), None, h * w)      # the ctx goes on working
    assert n == len(ref)


@pytest.mark.study
def test_three_launch_path_equals_the_chain(ctx):
    """Rounds 1-3 (count per block, scan, emit: TDV_DEPTH_THREE_PASS=1, study library) against the one-launch chain: same points, same order."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    out = {}
    for (h, w) in ((97, 131), (720, 1280)):
        raw, mask, bgr = _random_frame(h + w, h, w)
        d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_mask = torch.from_numpy(mask).to(dev); d_bgr = torch.from_numpy(bgr).to(dev)
        try:
            for mode in ("0", "1"):
                os.environ["TDV_DEPTH_THREE_PASS"] = mode
                d_xyz = torch.zeros((h * w, 3), dtype=torch.float32, device=dev); d_rgb = torch.zeros_like(d_xyz)
                n = ctx.depth_to_cloud_dev(d_raw.data_ptr(), d_mask.data_ptr(), d_bgr.data_ptr(), w, h, 1000.0, 611.5, 609.25, 60.0, 50.0, 2.0, d_xyz.data_ptr(), d_rgb.data_ptr(), h * w)
                out[mode] = (n, d_xyz.cpu().numpy().tobytes(), d_rgb.cpu().numpy().tobytes())
        finally:
            os.environ.pop("TDV_DEPTH_THREE_PASS", None)
        assert out["0"] == out["1"] and out["0"][0] > 0
