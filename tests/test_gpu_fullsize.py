"""BASELINE.json sizes (100k-200k points) through the C ABI.  The O(N^2) oracle cannot run these sizes in
seconds, so parity is checked (a) exactly on a random SAMPLE of rows against the oracle / a float32 numpy
evaluation of the same expression tree, and (b) through size-independent properties (self-match, permutation
equivariance, reproducibility, count conservation)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 200000


def _d2_f32(pts, q):
    """(pts - q).squaredNorm() evaluated as the reference does: dx*dx + (dy*dy + dz*dz), all float32."""
    d = (pts - q[None, :]).astype(np.float32)
    return (d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])).astype(np.float32)


@pytest.fixture(scope="module")
def clouds(synth):
    tgt, nrm = synth.sample_object(N, 42)
    src, T_gt = synth.make_scene(N, 42)
    return src, tgt, nrm, T_gt


def test_icp_correspondences_200k_sampled(ctx, orc, synth, clouds):
    """Config 'headline' (200k x 200k): every 400th source row against the oracle, all rows self-consistent."""
    src, tgt, nrm, T_gt = clouds
    T = synth.perturb(T_gt)
    thr = 0.002
    got = ctx.icp_correspondences(src, tgt, T, thr)
    sel = np.arange(0, N, 400)
    ref = orc.icp_correspondences(src[sel], tgt, None, T, thr, point_to_plane=False)
    assert np.array_equal(got["corr"][sel], ref["corr"])
    assert got["d2"][sel].tobytes() == ref["d2"].tobytes()
    assert np.array_equal(got["accepted"][sel], ref["accepted"])
    # the reported d2 is the distance to the reported index, and n_corr is the accepted count
    p = (src.astype(np.float32) @ T[:3, :3].T.astype(np.float32))  # not bit-exact (BLAS order) -> compare loosely
    assert got["n_corr"] == int(got["accepted"].sum())
    chk = np.linalg.norm(p + T[:3, 3] - tgt[got["corr"]], axis=1) ** 2
    assert np.allclose(chk, got["d2"], rtol=1e-3, atol=1e-9)


def test_icp_self_match_and_permutation_200k(ctx, clouds):
    """source = a permutation of the target under the identity: every point finds itself (d2 = 0) and the
    index is the inverse permutation; permuting the sources permutes the answer."""
    _, tgt, _, _ = clouds
    rng = np.random.default_rng(1)
    perm = rng.permutation(N)
    got = ctx.icp_correspondences(tgt[perm], tgt, np.eye(4, dtype=np.float32), 1e-6)
    assert (got["d2"] == 0).all() and got["n_corr"] == N
    # duplicates in the synthetic cloud are possible: the reported index must hold an identical point and be the lowest such
    assert np.array_equal(tgt[got["corr"]], tgt[perm])
    assert (got["corr"] <= perm).all()


def test_icp_200k_reproducible_and_converges(ctx, synth, clouds):
    src, tgt, nrm, T_gt = clouds
    T0 = synth.perturb(T_gt)
    a = ctx.icp(src, tgt, nrm, T0, 0.003, 12, True)
    b = ctx.icp(src, tgt, nrm, T0, 0.003, 12, True)
    assert a.transformation.tobytes() == b.transformation.tobytes() and a.n_corr == b.n_corr
    assert synth.rotation_angle(T_gt[:3, :3], a.transformation[:3, :3]) < 1e-3
    assert np.abs(T_gt[:3, 3] - a.transformation[:3, 3]).max() < 5e-4


def test_icp_pruned_search_200k_identical_to_scan(ctx, synth, clouds):
    """Headline size: the pruned walk, the hash grid and the brute-force scan give the same accepted correspondences
    (all 200k rows) and the same ICP result bit for bit — at a threshold of four point spacings (the grid hands over to
    the walk: some 30 points per cell) and at the reference's 0.4 spacings (the grid's own regime)."""
    src, tgt, nrm, T_gt = clouds
    for thr, T0, grid_runs in ((0.003, synth.perturb(T_gt), False),
                               (0.4 * float(synth.mean_spacing(N)), synth.perturb(T_gt, angle_deg=0.05, trans=0.0001), True)):
        try:
            ctx.set_icp_search("brute")
            cb = ctx.icp_correspondences(src, tgt, T0, thr)
            a = ctx.icp(src, tgt, nrm, T0, thr, 12, True)
            others = {}
            for mode in ("pruned", "grid"):
                ctx.set_icp_search(mode)
                others[mode] = (ctx.icp_correspondences(src, tgt, T0, thr), ctx.icp(src, tgt, nrm, T0, thr, 12, True), ctx.last_icp_search())
        finally:
            ctx.set_icp_search("auto")
        acc = cb["accepted"].astype(bool)
        assert acc.sum() > (N // 4 if not grid_runs else N // 100)
        for mode, (cp, b, used) in others.items():
            assert used == (mode if grid_runs else "pruned")
            assert np.array_equal(cp["accepted"], cb["accepted"]) and cp["n_corr"] == cb["n_corr"]
            assert np.array_equal(cp["corr"][acc], cb["corr"][acc])
            assert cp["d2"][acc].tobytes() == cb["d2"][acc].tobytes()
            assert a.transformation.tobytes() == b.transformation.tobytes()
            assert (a.iterations, a.n_corr, a.rmse, a.fitness) == (b.iterations, b.n_corr, b.rmse, b.fitness)


def test_feature_match_100k_sampled(ctx, orc, synth):
    """Config C3 (100k x 100k descriptors): 300 sampled source rows against the oracle."""
    n = 100000
    fs = synth.random_features(n, 11); ft = synth.random_features(n, 12)
    ft[77777] = ft[123]  # a duplicate descriptor: lowest index wins
    fs[5] = ft[123]
    corr = ctx.feature_match(fs, ft)
    sel = np.concatenate([[5], np.random.default_rng(2).choice(n, 300, replace=False)])
    assert np.array_equal(corr[sel], orc.feature_match(fs[sel], ft))
    assert corr[5] == 123


def test_knn_normals_200k_sampled(ctx, orc, synth, clouds):
    """200k-point normals: neighbour lists of sampled queries against a float32 numpy brute force in the
    reference's (d2, idx) order; their normals against the oracle's PCA on exactly those neighbours."""
    _, tgt, _, _ = clouds
    T = synth.gt_transform(3)
    Tinv = np.linalg.inv(T.astype(np.float64))
    pts = (tgt.astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
    nrm, knn = ctx.estimate_normals(pts, 30, want_knn=True)
    assert np.isfinite(nrm).all() and np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-4)
    assert ((nrm * -pts).sum(1) >= -1e-6).all()  # flipped towards the origin (registration.cpp:125)
    for q in np.random.default_rng(3).choice(N, 40, replace=False):
        d2 = _d2_f32(pts, pts[q])
        order = np.lexsort((np.arange(N), d2))[:30]
        assert np.array_equal(knn[q], order), q
        sub = pts[order]                      # the oracle on the 30 neighbours alone gives the same PCA for their first point
        sub_n = orc.estimate_normals(sub, 30)
        assert order[0] == q and sub_n[0].tobytes() == nrm[q].tobytes()


def test_fpfh_100k_sampled(ctx, synth, clouds):
    """100k-point FPFH: radius lists of sampled queries exact (incl. order and the cap), histogram rows normalised."""
    _, tgt, nrm_true, _ = clouds
    n = 100000
    pts = tgt[:n].copy(); nrm = nrm_true[:n].copy()
    radius = 5.0 * synth.mean_spacing(n)
    desc, nb, cnt = ctx.compute_fpfh(pts, nrm, radius, want_neighbors=True)
    r2 = np.float32(radius) * np.float32(radius)
    for q in np.random.default_rng(4).choice(n, 40, replace=False):
        d2 = _d2_f32(pts, pts[q])
        inr = np.nonzero(d2 <= r2)[0]
        order = inr[np.lexsort((inr, d2[inr]))][:100]
        assert cnt[q] == len(order) and np.array_equal(nb[q, :cnt[q]], order), q
    s = desc.sum(1)
    assert np.isfinite(desc).all() and np.allclose(s[s > 0], 1.0, atol=1e-5) and (s > 0).mean() > 0.99


def test_empty_and_degenerate_inputs(ctx, tdv):
    """Edge cases the reference's callers can produce: empty clouds, single points, zero iterations."""
    e3 = np.zeros((0, 3), np.float32)
    one = np.array([[0.1, 0.2, 0.9]], np.float32)
    I = np.eye(4, dtype=np.float32)
    r = ctx.icp(e3, one, None, I, 0.01, 10, True)
    assert r.iterations == 0 and np.array_equal(r.transformation, I)
    r = ctx.icp(one, e3, None, I, 0.01, 10, True)
    assert r.iterations == 0 and np.array_equal(r.transformation, I)
    r = ctx.icp(one, one, None, I, 0.01, 0, True)
    assert r.iterations == 0 and r.fitness == 0
    r = ctx.icp(one, one, None, I, 0.01, 5, True)      # n_corr = 1 < 3 -> break
    assert r.iterations == 0 and np.array_equal(r.transformation, I)
    rs = ctx.ransac(e3, one, corr=np.zeros(0, np.int32), voxel=0.01, max_iterations=10)
    assert rs.best_iteration == -1 and np.array_equal(rs.transformation, I) and rs.iterations_run == 0
    rs = ctx.ransac(one, one, corr=np.zeros(1, np.int32), voxel=0.01, max_iterations=10, trace=True)
    assert (rs.trace_inliers == -1).all() and rs.best_iteration == -1  # every triple repeats the only index
    v, _ = ctx.voxel_downsample(e3, None, 0.01, tdv.TDV_VOXEL_ORDER_FIRST)
    assert len(v) == 0
    v, _ = ctx.voxel_downsample(one, None, 0.01, tdv.TDV_VOXEL_ORDER_REFERENCE)
    assert v.tobytes() == one.tobytes()
    assert len(ctx.estimate_normals(e3, 30)) == 0 and len(ctx.compute_fpfh(e3, e3, 0.01)) == 0
    n1 = ctx.estimate_normals(one, 30)   # zero covariance: eigenvectors of the zero matrix = identity basis, col(0)
    assert n1.shape == (1, 3) and np.isfinite(n1).all()
    assert len(ctx.feature_match(np.zeros((0, 33), np.float32), np.zeros((4, 33), np.float32))) == 0
    with pytest.raises(tdv.TdvError):
        ctx.voxel_downsample(one, None, 0.0, tdv.TDV_VOXEL_ORDER_FIRST)


# ---- BASELINE config C5's scene size: 500k points (more than 64 groups of 4096: the group loop runs twice) --------

def test_searches_500k_sampled(ctx, orc, synth):
    """kNN lists, radius lists (cap 100) and ICP correspondences on a 500k-point cloud: sampled rows against a float32
    numpy evaluation of the reference's expression with (d2, index) ordering; pruned ICP search against the scan."""
    n = 500000
    pts, nrm = synth.sample_object(n, 5)
    rng = np.random.default_rng(9)
    sel = rng.choice(n, 120, replace=False)
    # kNN (k = 30)
    _, knn = ctx.estimate_normals(pts, 30, want_knn=True)
    for i in sel[:60]:
        d2 = _d2_f32(pts, pts[i])
        order = np.lexsort((np.arange(n), d2))[:30]
        assert np.array_equal(knn[i], order), i
    # radius lists
    radius = float(synth.mean_spacing(n)) * 5.0
    r2 = np.float32(radius) * np.float32(radius)
    _, nb, cnt = ctx.compute_fpfh(pts, nrm, radius, want_neighbors=True)
    for i in sel[60:]:
        d2 = _d2_f32(pts, pts[i])
        inside = np.nonzero(d2 <= r2)[0]
        order = inside[np.lexsort((inside, d2[inside]))][:100]
        assert cnt[i] == len(order), i
        assert np.array_equal(nb[i][:cnt[i]], order), i
    # ICP correspondences: pruned search == scan on every row; sampled rows == oracle
    src, T_gt = synth.make_scene(n, 5)
    T0 = synth.perturb(T_gt)
    thr = 0.003
    try:
        ctx.set_icp_search("brute")
        cb = ctx.icp_correspondences(src, pts, T0, thr)
        cps = []
        for mode in ("pruned", "grid"):
            ctx.set_icp_search(mode)
            cps.append(ctx.icp_correspondences(src, pts, T0, thr))
    finally:
        ctx.set_icp_search("auto")
    acc = cb["accepted"].astype(bool)
    assert acc.sum() > n // 4
    for cp in cps:
        assert np.array_equal(cp["accepted"], cb["accepted"]) and cp["n_corr"] == cb["n_corr"]
        assert np.array_equal(cp["corr"][acc], cb["corr"][acc])
        assert cp["d2"][acc].tobytes() == cb["d2"][acc].tobytes()
    ref = orc.icp_correspondences(src[sel], pts, None, T0, thr, point_to_plane=False)
    assert np.array_equal(cb["corr"][sel], ref["corr"]) and cb["d2"][sel].tobytes() == ref["d2"].tobytes()
