"""R6 parity: ICP through the C ABI vs the CPU oracle (reference src/registration.cpp:297-414).
Bar: correspondence indices, squared distances, accepted flags and n_corr bit-exact for a given T;
final transform within 1e-4 rad rotation and 1e-3 mm translation (BASELINE.json)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4        # rad
TRANS_TOL = 1e-6      # m  (= 1e-3 mm)


def _pair(synth, ns, nt, seed=42):
    tgt, nrm = synth.sample_object(nt, seed)
    src, T_gt = synth.make_scene(ns, seed)
    return src, tgt, nrm, T_gt


@pytest.mark.parametrize("ns,nt", [(3000, 2000), (1, 1), (5, 3), (1025, 9), (4097, 1031)])
def test_correspondences_bit_exact(ctx, orc, synth, ns, nt):
    src, tgt, nrm, T_gt = _pair(synth, ns, nt)
    T = synth.perturb(T_gt)
    thr = 0.004
    ref = orc.icp_correspondences(src, tgt, nrm, T, thr)
    got = ctx.icp_correspondences(src, tgt, T, thr)
    assert np.array_equal(got["corr"], ref["corr"])
    assert got["d2"].tobytes() == ref["d2"].tobytes()
    assert np.array_equal(got["accepted"], ref["accepted"])
    assert got["n_corr"] == ref["n_corr"]


def test_correspondences_ties_lowest_index(ctx, orc, synth):
    """Duplicated target points: strict < keeps the lowest index (registration.cpp:331)."""
    src, tgt, nrm, T_gt = _pair(synth, 2000, 700)
    tgt = np.concatenate([tgt, tgt[::-1], tgt[:300]], 0)  # every point appears 2-3 times
    ref = orc.icp_correspondences(src, tgt, None, T_gt, 0.01, point_to_plane=False)
    got = ctx.icp_correspondences(src, tgt, T_gt, 0.01)
    assert np.array_equal(got["corr"], ref["corr"])
    assert got["corr"].max() < 700  # always the first copy
    assert got["n_corr"] == ref["n_corr"]


def test_threshold_is_inclusive_and_sqrt_exact(ctx, orc):
    """d > thr rejects (registration.cpp:337-338): points exactly at the threshold are accepted."""
    tgt = np.zeros((4, 3), np.float32)
    tgt[1:] = 100.0
    thr = np.float32(0.003)
    d = np.array([0.003, np.nextafter(np.float32(0.003), np.float32(1)), np.nextafter(np.float32(0.003), np.float32(0)), 0.0029999], np.float32)
    src = np.zeros((len(d), 3), np.float32)
    src[:, 0] = d
    ref = orc.icp_correspondences(src, tgt, None, np.eye(4, dtype=np.float32), thr, point_to_plane=False)
    got = ctx.icp_correspondences(src, tgt, np.eye(4, dtype=np.float32), thr)
    assert np.array_equal(got["accepted"], ref["accepted"])
    assert got["d2"].tobytes() == ref["d2"].tobytes()


@pytest.mark.parametrize("p2plane", [True, False])
def test_icp_converges_like_oracle(ctx, orc, synth, p2plane):
    ns, nt = 4000, 3000
    src, tgt, nrm, T_gt = _pair(synth, ns, nt)
    T0 = synth.perturb(T_gt)
    thr = 0.004
    ref = orc.icp(src, tgt, nrm, T0, thr, 60, p2plane, trace=True)
    got = ctx.icp(src, tgt, nrm, T0, thr, 60, p2plane)
    ang = synth.rotation_angle(ref["T"][:3, :3], got.transformation[:3, :3])
    dt = np.abs(ref["T"][:3, 3].astype(np.float64) - got.transformation[:3, 3]).max()
    print("p2plane", p2plane, "iters", ref["iterations"], got.iterations, "ang", ang, "dt", dt, "fitness", ref["fitness"], got.fitness)
    assert ang <= ROT_TOL and dt <= TRANS_TOL
    assert abs(got.iterations - ref["iterations"]) <= 1
    assert abs(float(got.fitness) - float(ref["fitness"])) <= 2.0 / ns
    assert abs(float(got.rmse) - float(ref["rmse"])) <= 1e-6
    # and the result is the ground truth to sensor-noise level
    assert synth.rotation_angle(T_gt[:3, :3], got.transformation[:3, :3]) < 5e-3


def test_icp_first_iteration_matches_trace(ctx, orc, synth):
    """One iteration: same accepted set => T after the update agrees to rounding of the sums."""
    src, tgt, nrm, T_gt = _pair(synth, 3000, 2500)
    T0 = synth.perturb(T_gt)
    ref = orc.icp(src, tgt, nrm, T0, 0.004, 1, True, trace=True)
    got = ctx.icp(src, tgt, nrm, T0, 0.004, 1, True)
    assert got.iterations == 1 and ref["iterations"] == 1
    assert got.n_corr == int(ref["trace"][0, 18])
    assert np.abs(got.transformation - ref["T"]).max() < 2e-6
    assert abs(float(got.rmse) - float(ref["rmse"])) < 1e-7


def test_icp_too_few_correspondences_keeps_initial(ctx, orc, synth):
    """n_corr < 3 -> break with the initial transform, fitness 0, rmse 0 (registration.cpp:361)."""
    src, tgt, nrm, T_gt = _pair(synth, 500, 400)
    T0 = np.eye(4, dtype=np.float32)
    T0[:3, 3] = 5.0  # far away: nothing within the threshold
    ref = orc.icp(src, tgt, nrm, T0, 0.001, 20, True)
    got = ctx.icp(src, tgt, nrm, T0, 0.001, 20, True)
    assert ref["iterations"] == 0 and got.iterations == 0
    assert np.array_equal(got.transformation, T0) and got.fitness == 0 and got.rmse == 0


def test_icp_no_normals_falls_back_to_point_to_point(ctx, orc, synth):
    """point_to_plane && target.hasNormals() (registration.cpp:343)."""
    src, tgt, nrm, T_gt = _pair(synth, 2000, 1500)
    T0 = synth.perturb(T_gt)
    a = ctx.icp(src, tgt, None, T0, 0.004, 30, True)
    b = ctx.icp(src, tgt, nrm, T0, 0.004, 30, False)
    assert np.array_equal(a.transformation, b.transformation)


def test_icp_is_reproducible(ctx, synth):
    """Fixed-order reductions, no float atomics: identical bits run to run."""
    src, tgt, nrm, T_gt = _pair(synth, 6000, 5000)
    T0 = synth.perturb(T_gt)
    a = ctx.icp(src, tgt, nrm, T0, 0.004, 25, True)
    b = ctx.icp(src, tgt, nrm, T0, 0.004, 25, True)
    assert a.transformation.tobytes() == b.transformation.tobytes() and a.rmse == b.rmse and a.iterations == b.iterations


def test_gpu_registration_operator_api(tdv, orc, synth):
    """The reference's operator names: GPURegistration::icpRefine(source, target, T0, thr, max_iter)."""
    src, tgt, nrm, T_gt = _pair(synth, 1500, 1200)
    source = tdv.PointCloud(points=src)
    target = tdv.PointCloud(points=tgt, normals=nrm)
    assert tdv.GPURegistration.isCudaAvailable()
    res = tdv.GPURegistration.icpRefine(source, target, synth.perturb(T_gt), 0.004, 40)
    ref = orc.icp(src, tgt, nrm, synth.perturb(T_gt), 0.004, 40, True)
    assert synth.rotation_angle(ref["T"][:3, :3], res.transformation[:3, :3]) <= ROT_TOL
    assert np.abs(ref["T"][:3, 3] - res.transformation[:3, 3]).max() <= TRANS_TOL


# ---- exact pruned correspondence search (same results as the scan, bit for bit) -------------------------------

@pytest.fixture(params=["pruned", "grid"])
def pruned(ctx, request):
    """The two exact searches that replace the scan: the box walk, and the hash grid (which hands over to the walk when
    the threshold is large against the spacing - those cases then test the hand-over)."""
    ctx.set_icp_search(request.param)
    yield ctx
    ctx.set_icp_search("auto")


@pytest.mark.parametrize("ns,nt,thr", [(3000, 2000, 0.004), (1, 1, 0.5), (5, 3, 0.5), (1025, 9, 0.01), (4097, 1031, 0.004),
                                       (6000, 5000, 10.0), (6000, 5000, 1e-4), (300, 70000, 0.02)])
def test_pruned_search_matches_oracle(pruned, orc, synth, ns, nt, thr):
    """Accepted correspondences (index, d2) identical to the oracle's full scan; the rest are rejected in both."""
    src, tgt, nrm, T_gt = _pair(synth, ns, nt)
    T = synth.perturb(T_gt)
    ref = orc.icp_correspondences(src, tgt, nrm, T, thr)
    got = pruned.icp_correspondences(src, tgt, T, thr)
    acc = ref["accepted"].astype(bool)
    assert np.array_equal(got["accepted"], ref["accepted"])
    assert np.array_equal(got["corr"][acc], ref["corr"][acc])
    assert got["d2"][acc].tobytes() == ref["d2"][acc].tobytes()
    assert got["n_corr"] == ref["n_corr"]
    if thr >= 10.0:
        assert acc.all()   # loose threshold: the seeded walk still returns the true nearest neighbour of every point
        assert pruned.last_icp_search() == "pruned"      # far too many points per cell for the grid
    if pruned.icp_search_name == "grid" and (ns, nt, thr) in ((3000, 2000, 0.004), (4097, 1031, 0.004), (6000, 5000, 1e-4)):
        assert pruned.last_icp_search() == "grid"


def test_pruned_search_ties_lowest_index(pruned, orc, synth):
    src, tgt, nrm, T_gt = _pair(synth, 2000, 700)
    tgt = np.concatenate([tgt, tgt[::-1], tgt[:300]], 0)
    ref = orc.icp_correspondences(src, tgt, None, T_gt, 0.01, point_to_plane=False)
    got = pruned.icp_correspondences(src, tgt, T_gt, 0.01)
    acc = ref["accepted"].astype(bool)
    assert acc.sum() > 1000
    assert np.array_equal(got["accepted"], ref["accepted"])
    assert np.array_equal(got["corr"][acc], ref["corr"][acc])
    assert got["corr"][acc].max() < 700


def test_pruned_threshold_inclusive(pruned, orc):
    tgt = np.zeros((4, 3), np.float32)
    tgt[1:] = 100.0
    thr = np.float32(0.003)
    d = np.array([0.003, np.nextafter(np.float32(0.003), np.float32(1)), np.nextafter(np.float32(0.003), np.float32(0)), 0.0029999], np.float32)
    src = np.zeros((len(d), 3), np.float32)
    src[:, 0] = d
    ref = orc.icp_correspondences(src, tgt, None, np.eye(4, dtype=np.float32), thr, point_to_plane=False)
    got = pruned.icp_correspondences(src, tgt, np.eye(4, dtype=np.float32), thr)
    assert np.array_equal(got["accepted"], ref["accepted"])


@pytest.mark.parametrize("p2plane", [True, False])
@pytest.mark.parametrize("thr", [0.004, 0.05])
def test_icp_pruned_and_brute_identical_bits(ctx, synth, p2plane, thr):
    src, tgt, nrm, T_gt = _pair(synth, 9000, 7000)
    T0 = synth.perturb(T_gt)
    try:
        ctx.set_icp_search("brute")
        a = ctx.icp(src, tgt, nrm, T0, thr, 30, p2plane)
        ctx.set_icp_search("pruned")
        b = ctx.icp(src, tgt, nrm, T0, thr, 30, p2plane)
        ctx.set_icp_search("grid")
        c = ctx.icp(src, tgt, nrm, T0, thr, 30, p2plane)
        used = ctx.last_icp_search()
    finally:
        ctx.set_icp_search("auto")
    for o in (b, c):
        assert a.transformation.tobytes() == o.transformation.tobytes()
        assert (a.iterations, a.n_corr) == (o.iterations, o.n_corr)
        assert a.rmse == o.rmse and a.fitness == o.fitness
    assert used == ("grid" if thr == 0.004 else "pruned")   # 7,000 points of the cuboid are 3.6 mm apart: 0.05 m puts hundreds into a cell


@pytest.mark.parametrize("thr", [0.0, -1.0, 1e-30, 1e18])
def test_pruned_search_threshold_edges(pruned, orc, synth, thr):
    """Zero, negative, denormal-square and huge thresholds: the accepted set and its correspondences match the scan.
    (A threshold whose inclusive bound is not finite falls back to the scan inside the library.)"""
    _, tgt, nrm, _ = _pair(synth, 10, 1500)
    src = np.concatenate([tgt[:300], tgt[300:600] + np.float32(1e-4)], 0)   # exact hits and near misses
    T = np.eye(4, dtype=np.float32)
    ref = orc.icp_correspondences(src, tgt, None, T, thr, point_to_plane=False)
    got = pruned.icp_correspondences(src, tgt, T, thr)
    acc = ref["accepted"].astype(bool)
    assert np.array_equal(got["accepted"], ref["accepted"]) and got["n_corr"] == ref["n_corr"]
    assert np.array_equal(got["corr"][acc], ref["corr"][acc])
    assert got["d2"][acc].tobytes() == ref["d2"][acc].tobytes()
    if thr == 0.0:
        assert acc[:300].all()          # d2 = 0 passes the inclusive test
    if thr < 0:
        assert not acc.any()


@pytest.mark.parametrize("offset,expect_grid", [(0.0, True), (-37.5, True), (50.0, True), (114.0, True), (116.5, False), (4000.0, False)])
def test_grid_search_far_from_the_origin(ctx, orc, synth, offset, expect_grid):
    """The hash grid's completeness argument is about rounding of cell coordinates: clouds moved away from the origin, on
    both sides of it, up to the 2^17-cell limit of the build (threshold 0.4 mm -> cells of 0.88 mm -> 115.3 m) and beyond
    it, where the search hands over to the box walk.  Accepted correspondences identical to the oracle's scan."""
    ns, nt = 6000, 5000
    src, tgt, nrm, T_gt = _pair(synth, ns, nt)
    shift = np.array([offset, -offset * 0.5, offset * 0.25], np.float32)
    tgt = (tgt + shift).astype(np.float32)
    T = synth.perturb(T_gt, angle_deg=0.05, trans=0.0002).astype(np.float64)
    T[:3, 3] += shift.astype(np.float64)
    T = T.astype(np.float32)
    thr = 0.0004
    ref = orc.icp_correspondences(src, tgt, None, T, thr, point_to_plane=False)
    try:
        ctx.set_icp_search("grid")
        got = ctx.icp_correspondences(src, tgt, T, thr)
        used = ctx.last_icp_search()
    finally:
        ctx.set_icp_search("auto")
    assert used == ("grid" if expect_grid else "pruned")
    acc = ref["accepted"].astype(bool)
    assert np.array_equal(got["accepted"], ref["accepted"]) and got["n_corr"] == ref["n_corr"]
    assert np.array_equal(got["corr"][acc], ref["corr"][acc]) and got["d2"][acc].tobytes() == ref["d2"][acc].tobytes()
    if abs(offset) < 60: assert acc.sum() > 50        # (far out the float grid is coarser than the threshold: few or no matches)
