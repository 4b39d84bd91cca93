"""R4a/R4b parity: kNN normals and FPFH through the C ABI vs the CPU oracle
(reference src/registration.cpp:63-130 and :83-102,133-201).
Bar: neighbour lists (sets AND (d2, idx) order) exact; normals / descriptors bit-exact where the
arithmetic is IEEE-determined; SPFH's theta comes from glibc's atan2f restated on the device (csrc/libm_f32.hpp, pinned against
the running libm by tests/test_libm_restatement.py), so descriptors are bit-exact too (round 4; rounds 1-3 allowed 0.5 % of rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(synth, n, seed=42):
    pts, _ = synth.sample_object(n, seed)
    T = synth.gt_transform(seed)
    Tinv = np.linalg.inv(T.astype(np.float64))
    return (pts.astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)  # camera-frame cloud, z ~ 0.8


@pytest.mark.parametrize("n,k", [(1500, 30), (700, 7), (257, 30), (3001, 32), (20, 30), (1, 30)])
def test_knn_lists_and_normals(ctx, orc, synth, n, k):
    pts = _cloud(synth, n)
    if n > 100:
        pts[50] = pts[10]; pts[51] = pts[10]  # duplicates: ties resolved by lower index
    ref_n, ref_knn = orc.estimate_normals(pts, k, want_knn=True)
    got_n, got_knn = ctx.estimate_normals(pts, k, want_knn=True)
    assert np.array_equal(got_knn, ref_knn)
    bit = got_n.tobytes() == ref_n.tobytes()
    print("normals bitwise equal:", bit, "max abs diff", np.abs(got_n - ref_n).max())
    assert bit


def test_knn_large_k_uses_global_lists(ctx, orc, synth):
    pts = _cloud(synth, 900)
    ref_n, ref_knn = orc.estimate_normals(pts, 64, want_knn=True)
    got_n, got_knn = ctx.estimate_normals(pts, 64, want_knn=True)
    assert np.array_equal(got_knn, ref_knn)
    assert got_n.tobytes() == ref_n.tobytes()


def test_demo_model_normals(ctx, orc):
    """Config C1's reference model (src/pipeline.cpp:275-282): 40x40 planar grid, 5 mm pitch —
    kNN distances tie massively, the covariance is rank 2."""
    model, _ = orc.demo_model()
    ref_n, ref_knn = orc.estimate_normals(model, 30, want_knn=True)
    got_n, got_knn = ctx.estimate_normals(model, 30, want_knn=True)
    assert np.array_equal(got_knn, ref_knn)
    assert got_n.tobytes() == ref_n.tobytes()


@pytest.mark.parametrize("n,radius", [(1500, 0.012), (1500, 0.05), (2500, 0.006)])
def test_fpfh(ctx, orc, synth, n, radius):
    """Neighbour lists exact (incl. the cap of 100 by (d2, idx)).  Descriptors bit for bit: every operation is IEEE-determined
    except atan2 (registration.cpp:154), for which the device runs glibc's own algorithm (the oracle calls this image's atan2f):
    every pair lands in the reference's theta bin."""
    pts = _cloud(synth, n)
    nrm = orc.estimate_normals(pts, 30)
    ref_d, ref_nb, ref_cnt = orc.compute_fpfh(pts, nrm, radius, want_neighbors=True)
    got_d, got_nb, got_cnt = ctx.compute_fpfh(pts, nrm, radius, want_neighbors=True)
    assert np.array_equal(got_cnt, ref_cnt)
    assert np.array_equal(got_nb, ref_nb)
    if radius >= 0.05:
        assert ref_cnt.max() == 100  # the cap is exercised
    same = (got_d.view(np.uint32) == ref_d.view(np.uint32)).all(1)
    print("fpfh rows bitwise equal: %d / %d ; max abs diff %.3g" % (same.sum(), n, np.abs(got_d - ref_d).max()))
    assert same.all(), "rows differing: %d of %d" % ((~same).sum(), n)


@pytest.mark.parametrize("n,radius", [(3000, 0.012), (3000, 0.004), (40, 0.05), (1, 0.01)])
def test_one_neighbour_walk_for_normals_and_fpfh(ctx, synth, n, radius):
    """tdv_normals_fpfh_dev (ONE radius search; the 30-NN list is the head of the sorted radius list, a kNN search of their own only for
    the points with fewer than 30 neighbours in radius - most of them at radius 0.004) == estimateNormals(30) then computeFPFH(radius),
    bit for bit.  (registration.cpp:68-74,95-99: both lists are sorted by (d2, idx).)"""
    import torch
    dev = torch.device("cuda", 0)
    pts = _cloud(synth, max(n, 40))[:n].copy()
    nrm = ctx.estimate_normals(pts, 30)
    fp = ctx.compute_fpfh(pts, nrm, radius)
    d_x = torch.from_numpy(pts).to(dev); d_n = torch.empty_like(d_x); d_f = torch.empty((n, 33), dtype=torch.float32, device=dev)
    ctx.normals_fpfh_dev(d_x.data_ptr(), n, 30, radius, d_n.data_ptr(), d_f.data_ptr())
    assert d_n.cpu().numpy().tobytes() == nrm.tobytes() and d_f.cpu().numpy().tobytes() == fp.tobytes()


@pytest.mark.study
@pytest.mark.parametrize("n", [1, 2, 3, 65, 1001])
def test_fpfh_two_points_per_wave_equals_one(ctx, orc, synth, n):
    """k_spfh_pairs / k_fpfh_pairs (a wave owns two points; the default) against the one-point-per-wave kernels
    (TDV_FPFH_PAIRS=0): identical descriptors and neighbour lists, bit for bit, for odd and tiny clouds too (a last wave with one
    point, a cloud of one point: no neighbour but itself), and the neighbour lists are the oracle's."""
    import os
    pts = _cloud(synth, max(n, 40))[:n].copy()
    nrm = orc.estimate_normals(_cloud(synth, max(n, 40)), 30)[:n].copy()
    out = {}
    try:
        for mode in ("1", "0"):
            os.environ["TDV_FPFH_PAIRS"] = mode
            out[mode] = ctx.compute_fpfh(pts, nrm, 0.02, want_neighbors=True)
    finally:
        os.environ.pop("TDV_FPFH_PAIRS", None)
    for a, b in zip(out["1"], out["0"]):
        assert a.tobytes() == b.tobytes()
    ref_d, ref_nb, ref_cnt = orc.compute_fpfh(pts, nrm, 0.02, want_neighbors=True)
    assert np.array_equal(out["1"][2], ref_cnt) and np.array_equal(out["1"][1], ref_nb)
    assert np.abs(out["1"][0] - ref_d).max() < 0.05


def test_fpfh_demo_model(ctx, orc):
    """Demo model: neighbours at exactly radius = 5*voxel = 5 mm sit on the d2 <= r2 edge."""
    model, _ = orc.demo_model()
    nrm = orc.estimate_normals(model, 30)
    ref_d, ref_nb, ref_cnt = orc.compute_fpfh(model, nrm, 0.001 * 5.0, want_neighbors=True)
    got_d, got_nb, got_cnt = ctx.compute_fpfh(model, nrm, 0.001 * 5.0, want_neighbors=True)
    assert np.array_equal(got_cnt, ref_cnt) and np.array_equal(got_nb, ref_nb)
    assert np.abs(got_d - ref_d).max() < 1e-6


def test_registration_operator_api_chain(tdv, orc, synth):
    """estimateNormals -> computeFPFH in the reference's own call shape (src/pipeline.cpp:93-95)."""
    pts = _cloud(synth, 800)
    cloud = tdv.PointCloud(points=pts)
    tdv.Registration.estimateNormals(cloud, 30)
    assert cloud.hasNormals()
    feats = tdv.Registration.computeFPFH(cloud, 0.015)
    ref_n = orc.estimate_normals(pts, 30)
    assert cloud.normals.tobytes() == ref_n.tobytes()
    ref_f = orc.compute_fpfh(pts, ref_n, 0.015)
    assert np.abs(feats - ref_f).max() < 0.05
    sums = feats.sum(1)  # isolated points keep an all-zero histogram (sum > 0 guard, registration.cpp:194)
    assert np.allclose(sums[sums > 0], 1.0, atol=1e-5) and (sums > 0).mean() > 0.9


def test_knn_massive_ties_exercise_the_fallbacks(ctx, orc, synth):
    """More candidates at exactly the k-th distance than a query's candidate row holds (128): the row overflows
    repeatedly, and cutting it to the best k cannot lower the bound below the tie.  (d2, idx) order still decides:
    the lowest indices win."""
    pts = _cloud(synth, 1200)
    dup = np.tile(pts[7], (300, 1))                       # 300 exact copies of point 7 (indices 1200..1499)
    pts = np.concatenate([pts, dup], 0).astype(np.float32)
    ref_n, ref_knn = orc.estimate_normals(pts, 30, want_knn=True)
    got_n, got_knn = ctx.estimate_normals(pts, 30, want_knn=True)
    assert np.array_equal(got_knn, ref_knn)
    assert list(got_knn[7][:5]) == [7, 1200, 1201, 1202, 1203]   # d2 = 0 ties broken by index
    assert np.array_equal(got_knn[1400][:3], [7, 1200, 1201])
    same = (got_n.view(np.uint32) == ref_n.view(np.uint32)).all(1) | (np.isnan(got_n).any(1) & np.isnan(ref_n).any(1))
    assert same.all()
    # the radius search sees the same pile-up: cap of 100 by (d2, idx)
    ref_d, ref_nb, ref_cnt = orc.compute_fpfh(pts, np.nan_to_num(ref_n), 0.01, want_neighbors=True)
    got_d, got_nb, got_cnt = ctx.compute_fpfh(pts, np.nan_to_num(ref_n), 0.01, want_neighbors=True)
    assert np.array_equal(got_cnt, ref_cnt) and np.array_equal(got_nb, ref_nb) and got_cnt.max() == 100


@pytest.mark.parametrize("n,k", [(700, 200), (900, 255), (300, 255), (1500, 193)])
def test_knn_very_large_k(ctx, orc, synth, n, k):
    """k above 192 takes the widest candidate row (512 keys per query); k above n is clamped like the CPU's
    std::min(k, dists.size())."""
    pts = _cloud(synth, n, seed=9)
    pts[5] = pts[2]
    ref_n, ref_knn = orc.estimate_normals(pts, k, want_knn=True)
    got_n, got_knn = ctx.estimate_normals(pts, k, want_knn=True)
    assert np.array_equal(got_knn, ref_knn)
    assert got_n.tobytes() == ref_n.tobytes()


def test_unbounded_searches_small_clouds(ctx, orc, synth):
    """Bounds that stay +inf: kNN with fewer points than k, and a radius so large that r*r overflows to +inf
    (every point is a neighbour; the cap of 100 by (d2, idx) decides)."""
    for n, k in ((100, 255), (64, 100), (65, 64), (3, 30)):
        pts = _cloud(synth, n, seed=4)
        ref_n, ref_knn = orc.estimate_normals(pts, k, want_knn=True)
        got_n, got_knn = ctx.estimate_normals(pts, k, want_knn=True)
        assert np.array_equal(got_knn, ref_knn), (n, k)
    pts = _cloud(synth, 700, seed=6)
    nrm = orc.estimate_normals(pts, 30)
    for radius in (1e20, 1e30):
        ref_d, ref_nb, ref_cnt = orc.compute_fpfh(pts, nrm, radius, want_neighbors=True)
        got_d, got_nb, got_cnt = ctx.compute_fpfh(pts, nrm, radius, want_neighbors=True)
        assert (got_cnt == 100).all() and np.array_equal(got_cnt, ref_cnt)
        assert np.array_equal(got_nb, ref_nb)
