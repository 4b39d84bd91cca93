"""R3 parity: voxel downsample through the C ABI vs the CPU oracle (reference
src/registration.cpp:29-60).  Bar: voxel count exact; per-voxel means bit-exact; in
TDV_VOXEL_ORDER_REFERENCE the output ORDER equals the reference's unordered_map iteration order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_demo_cloud_reference_order(ctx, orc, tdv):
    """Config C1: 40,401 masked pixels -> 32,129 voxels at 1 mm (1,476 at 5 mm), with colours."""
    depth, bgr = orc.demo_scene()
    d = orc.depth_preprocess(depth, orc.demo_mask(), 1000.0)
    xyz, rgb = orc.unproject(d, bgr, 900, 900, 640, 360, 1.5)
    for voxel, expect in ((0.001, 32129), (0.005, 1476)):
        ref_xyz, ref_rgb, _ = orc.voxel_downsample(xyz, rgb, voxel)
        assert len(ref_xyz) == expect
        got_xyz, got_rgb = ctx.voxel_downsample(xyz, rgb, voxel, tdv.TDV_VOXEL_ORDER_REFERENCE)
        assert got_xyz.tobytes() == ref_xyz.tobytes()
        assert got_rgb.tobytes() == ref_rgb.tobytes()


@pytest.mark.parametrize("n,voxel", [(5000, 0.004), (1, 0.01), (2049, 0.5), (4097, 0.0005), (30000, 0.02)])
def test_first_occurrence_order(ctx, orc, synth, tdv, n, voxel):
    """TDV_VOXEL_ORDER_FIRST: voxels sorted by their smallest member index; same means."""
    pts, _ = synth.sample_object(n, 3)
    pts = pts + np.float32(0.3)  # mixed-sign keys after the shift below
    pts[: n // 2] -= np.float32(0.45)
    ref_xyz, _, first = orc.voxel_downsample(pts, None, voxel)
    order = np.argsort(first, kind="stable")
    got_xyz, got_rgb = ctx.voxel_downsample(pts, None, voxel, tdv.TDV_VOXEL_ORDER_FIRST)
    assert got_rgb is None
    assert len(got_xyz) == len(ref_xyz)
    assert got_xyz.tobytes() == ref_xyz[order].tobytes()
    got_ref, _ = ctx.voxel_downsample(pts, None, voxel, tdv.TDV_VOXEL_ORDER_REFERENCE)
    assert got_ref.tobytes() == ref_xyz.tobytes()


def test_all_points_in_one_voxel_and_idempotence(ctx, orc, synth, tdv):
    pts, _ = synth.sample_object(3000, 5)
    got, _ = ctx.voxel_downsample(pts, None, 10.0, tdv.TDV_VOXEL_ORDER_FIRST)
    ref, _, _ = orc.voxel_downsample(pts, None, 10.0)
    assert len(ref) <= 8 and got.tobytes() == ref[np.argsort(orc.voxel_downsample(pts, None, 10.0)[2])].tobytes()
    # downsampling a downsampled cloud with the same voxel cannot increase the count
    a, _ = ctx.voxel_downsample(pts, None, 0.01, tdv.TDV_VOXEL_ORDER_FIRST)
    b, _ = ctx.voxel_downsample(a, None, 0.01, tdv.TDV_VOXEL_ORDER_FIRST)
    assert len(b) <= len(a)


def test_voxel_operator_api_drops_normals(tdv, orc, synth):
    """Registration::voxelDownsample returns points (+ colours), never normals (registration.cpp:42-56)."""
    pts, nrm = synth.sample_object(2000, 9)
    cloud = tdv.PointCloud(points=pts, normals=nrm)
    out = tdv.Registration.voxelDownsample(cloud, 0.01)
    ref, _, _ = orc.voxel_downsample(pts, None, 0.01)
    assert not out.hasNormals() and not out.hasColors()
    assert out.points.tobytes() == ref.tobytes()


def test_voxel_full_size_properties(ctx, synth, tdv):
    """200k points (BASELINE size): count conservation and mean-of-means invariants."""
    pts, _ = synth.sample_object(200000, 11)
    out, _ = ctx.voxel_downsample(pts, None, 0.002, tdv.TDV_VOXEL_ORDER_FIRST)
    keys = np.floor(pts * np.float32(1.0 / np.float32(0.002))).astype(np.int64)
    uniq, first_idx, counts = np.unique(keys, axis=0, return_index=True, return_counts=True)
    assert len(out) == len(uniq)
    # every output point lies inside its voxel (up to rounding of the mean) and the voxels come in first-occurrence order
    okeys = np.floor(out.astype(np.float64) / 0.002 + 1e-3).astype(np.int64)
    exp = keys[np.sort(first_idx)]
    assert (np.abs(okeys - exp) <= 1).all()
    w = counts[np.argsort(first_idx)][:, None].astype(np.float64)
    assert np.allclose((out.astype(np.float64) * w).sum(0) / len(pts), pts.astype(np.float64).mean(0), atol=1e-6)


@pytest.mark.parametrize("n,voxel", [(200000, 0.0008), (120000, 0.003), (7, 0.001), (100000, 1e-5)])
def test_reference_order_emulation_at_size(ctx, orc, synth, tdv, n, voxel):
    """The container-order emulation (libstdc++ node-list manipulation replayed on the leaders only) against the
    oracle's real std::unordered_map over all points, across many rehashes (up to 200k distinct voxels), negative keys
    and heavy sharing (many points per voxel)."""
    pts, _ = synth.sample_object(n, 11)
    pts = pts - np.float32(0.07)
    ref_xyz, _, _ = orc.voxel_downsample(pts, None, voxel)
    got_xyz, _ = ctx.voxel_downsample(pts, None, voxel, tdv.TDV_VOXEL_ORDER_REFERENCE)
    assert len(got_xyz) == len(ref_xyz)
    assert got_xyz.tobytes() == ref_xyz.tobytes()


@pytest.mark.parametrize("kmax", [15, 16, 17, 40])
def test_member_counts_around_the_hash_path_row_size(ctx, orc, tdv, kmax):
    """The hash-table path keeps up to 16 member indices per voxel and hands a call with a fuller voxel to the counting-sort
    path: voxels holding exactly 1 .. kmax points (shuffled input order, colours, negative cells) straddle that limit; means
    (sums in ascending input index) and both orders are the oracle's either way."""
    rng = np.random.default_rng(kmax)
    voxel = np.float32(0.01)
    cells = rng.permutation(np.arange(-60, 60))[:kmax * 3].reshape(-1, 3)[:kmax]          # distinct cells
    pts = []
    for k, c in enumerate(cells):
        for rep in range(1 + (k % 3 == 0)):                                              # some counts appear twice (in other cells)
            cell = c + np.array([rep * 200, 0, 0])
            pts.append((cell + 0.05 + 0.9 * rng.random((k + 1, 3))) * float(voxel))
    pts = np.concatenate(pts).astype(np.float32)
    perm = rng.permutation(len(pts)); pts = pts[perm]
    rgb = rng.random((len(pts), 3)).astype(np.float32)
    ref_xyz, ref_rgb, first = orc.voxel_downsample(pts, rgb, float(voxel))
    got_xyz, got_rgb = ctx.voxel_downsample(pts, rgb, float(voxel), tdv.TDV_VOXEL_ORDER_REFERENCE)
    assert got_xyz.tobytes() == ref_xyz.tobytes() and got_rgb.tobytes() == ref_rgb.tobytes()
    order = np.argsort(first, kind="stable")
    f_xyz, f_rgb = ctx.voxel_downsample(pts, rgb, float(voxel), tdv.TDV_VOXEL_ORDER_FIRST)
    assert f_xyz.tobytes() == ref_xyz[order].tobytes() and f_rgb.tobytes() == ref_rgb[order].tobytes()


@pytest.mark.parametrize("n,voxel", [(200000, 0.0008), (120000, 0.003), (7, 0.001), (100000, 1e-5), (30000, 0.004), (60000, 0.0015)])
def test_reference_order_on_the_device_at_size(ctx, orc, synth, tdv, n, voxel):
    """The container order computed ON THE DEVICE (closed form per rehash period: what the batch uses for all its instances, and the
    single-cloud call from 10k voxels) against the oracle's real std::unordered_map over all points and against the host replay:
    up to 17 rehash periods, negative keys, heavy sharing, colours carried through the permutation."""
    import os
    pts, _ = synth.sample_object(n, 11)
    pts = pts - np.float32(0.07)
    rgb = np.random.default_rng(n).random((n, 3)).astype(np.float32)
    ref_xyz, ref_rgb, _ = orc.voxel_downsample(pts, rgb, voxel)
    try:
        os.environ["TDV_VOXEL_DEVICE_ORDER"] = "1"
        got_xyz, got_rgb = ctx.voxel_downsample(pts, rgb, voxel, tdv.TDV_VOXEL_ORDER_REFERENCE)
        os.environ["TDV_VOXEL_DEVICE_ORDER"] = "0"
        host_xyz, host_rgb = ctx.voxel_downsample(pts, rgb, voxel, tdv.TDV_VOXEL_ORDER_REFERENCE)
    finally:
        os.environ.pop("TDV_VOXEL_DEVICE_ORDER", None)
    assert len(got_xyz) == len(ref_xyz)
    assert got_xyz.tobytes() == ref_xyz.tobytes() and got_rgb.tobytes() == ref_rgb.tobytes()
    assert host_xyz.tobytes() == ref_xyz.tobytes() and host_rgb.tobytes() == ref_rgb.tobytes()
