"""The oracle against INDEPENDENT third-party implementations of the same published algorithms (LAPACK through numpy,
scipy's k-d tree, numpy's MT19937), stage by stage.  The reference holds no fixtures and cannot be built here (DESIGN.md
2: parity unpinned by the reference), so these are the strongest pins available in this image: they tie the restatement's
MATHEMATICS to code written by others — not to Eigen's last bit, which nothing here can reach.  Float comparisons carry
the tolerance of a float32 computation checked in float64."""
import numpy as np
import pytest
from scipy.spatial import cKDTree


def test_index_stream_against_numpy_mt19937(orc):
    """std::mt19937(42) == numpy's MT19937 with init_genrand seeding; libstdc++-11's uniform_int_distribution<size_t> over
    [0, n-1] == Lemire's nearly-divisionless method on 32-bit draws (bits/uniform_int_dist.h:244-268), written here in numpy."""
    bg = np.random.MT19937()
    bg._legacy_seeding(42)
    raw = iter(bg.random_raw(4000).astype(np.uint64))
    for n in (1600, 32129, 200000):
        bg._legacy_seeding(42); raw = iter(bg.random_raw(4000).astype(np.uint64))
        out = []
        for _ in range(3 * 200):
            m = int(next(raw)) * n
            lo = m & 0xFFFFFFFF
            if lo < n:
                t = ((1 << 32) - n) % n
                while lo < t:
                    m = int(next(raw)) * n
                    lo = m & 0xFFFFFFFF
            out.append(m >> 32)
        assert np.array_equal(orc.sample_triples(n, 200).reshape(-1).astype(np.int64), np.array(out, np.int64)), n


def test_small_solvers_against_lapack(orc):
    rng = np.random.default_rng(3)
    for _ in range(200):
        A = rng.standard_normal((3, 3)).astype(np.float32)
        _, S, _ = orc.jacobi_svd3(A)
        assert np.allclose(S, np.linalg.svd(A.astype(np.float64), compute_uv=False), rtol=2e-5, atol=2e-6)
        # Kabsch: R = V U^T with the reflection fix (registration.cpp:255-262) against the textbook form via LAPACK
        U, _, Vt = np.linalg.svd(A.astype(np.float64))
        D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
        assert np.allclose(orc.kabsch_rotation(A), Vt.T @ D @ U.T, atol=5e-4 / max(np.linalg.svd(A, compute_uv=False)[1], 1e-2))
        B = ((A + A.T) / 2).astype(np.float32)
        w, Q, rc = orc.self_adjoint_eig3(B)
        wr, Qr = np.linalg.eigh(B.astype(np.float64))
        assert rc == 0 and np.allclose(w, wr, atol=3e-6 * max(1, np.abs(B).max()))
        if wr[1] - wr[0] > 1e-2:                      # smallest eigenvector (the normal direction), up to sign
            assert abs(abs(Q[:, 0] @ Qr[:, 0]) - 1) < 1e-4
        M = rng.standard_normal((6, 9)); C = (M @ M.T).astype(np.float32); b = rng.standard_normal(6).astype(np.float32)
        assert np.allclose(orc.ldlt6_solve(C, b), np.linalg.solve(C.astype(np.float64), b.astype(np.float64)), rtol=2e-3, atol=2e-4)


def test_neighbour_lists_against_scipy_kdtree(orc, synth):
    pts, _ = synth.sample_object(3000, 5)
    tree = cKDTree(pts.astype(np.float64))
    _, knn = orc.estimate_normals(pts, 30, want_knn=True)
    d, idx = tree.query(pts.astype(np.float64), 30)
    # the same neighbour SETS wherever the 30th and 31st distances differ clearly (ties are the reference's (d2, idx) rule)
    d31, _ = tree.query(pts.astype(np.float64), 31)
    clear = (d31[:, 30] - d31[:, 29]) > 1e-6
    assert clear.mean() > 0.9
    assert all(set(knn[i]) == set(idx[i]) for i in np.nonzero(clear)[0])
    radius = 0.012
    _, nb, cnt = orc.compute_fpfh(pts, orc.estimate_normals(pts, 30), radius, want_neighbors=True)
    balls = tree.query_ball_point(pts.astype(np.float64), radius)
    for i in range(0, 3000, 7):
        inside = np.array(sorted(balls[i]))
        dd = np.linalg.norm(pts[inside].astype(np.float64) - pts[i], axis=1)
        if len(inside) <= 100 and (np.abs(dd - radius) > 1e-6).all():      # not capped, nobody on the rim
            assert cnt[i] == len(inside) and set(nb[i, :cnt[i]]) == set(inside)


def test_normals_against_lapack_pca(orc, synth):
    pts, _ = synth.sample_object(2000, 9)
    nrm, knn = orc.estimate_normals(pts, 30, want_knn=True)
    for i in range(0, 2000, 11):
        q = pts[knn[i]].astype(np.float64)
        c = np.cov((q - q.mean(0)).T, bias=True)
        w, V = np.linalg.eigh(c)
        if w[1] - w[0] > 1e-3 * w[2]:                 # a well-defined normal
            n = V[:, 0] if V[:, 0] @ (-pts[i]) >= 0 else -V[:, 0]      # flipped towards the origin (registration.cpp:125-127)
            assert np.abs(nrm[i] - n).max() < 2e-3, i


def test_voxel_means_against_numpy_groupby(orc, synth):
    pts, _ = synth.sample_object(20000, 2)
    voxel = 0.004
    out, _, _ = orc.voxel_downsample(pts, None, voxel)
    inv = np.float32(1.0) / np.float32(voxel)                                 # float inv = 1.0f / voxel_size (registration.cpp:32): 249.99998, not 250
    key = np.floor(pts * inv).astype(np.int64)                               # :35-37
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    assert len(out) == len(uniq)
    mean = np.zeros((len(uniq), 3)); np.add.at(mean, inv.ravel(), pts.astype(np.float64)); mean /= np.bincount(inv.ravel())[:, None]
    # the oracle's order is the unordered_map's: match the two sets of means point by point
    d, j = cKDTree(mean).query(out.astype(np.float64))
    assert d.max() < 1e-6 and len(set(j.tolist())) == len(out)


def test_icp_step_against_numpy_least_squares(orc, synth):
    """One point-to-plane iteration (registration.cpp:325-372): correspondences by k-d tree, normal equations and their
    solution by numpy in float64, small-angle update composed as Rx Ry Rz."""
    tgt, nrm = synth.sample_object(4000, 1)
    src, T_gt = synth.make_scene(3000, 1, outlier_frac=0.0)
    T0 = synth.perturb(T_gt, angle_deg=0.5, trans=0.001)
    thr = 0.01
    r = orc.icp_correspondences(src, tgt, nrm, T0, thr)
    p = src.astype(np.float64) @ T0[:3, :3].astype(np.float64).T + T0[:3, 3]
    d, j = cKDTree(tgt.astype(np.float64)).query(p)
    acc = d <= thr
    clear = np.abs(d - thr) > 1e-6
    assert np.array_equal(r["accepted"][clear], acc[clear])
    same = r["corr"] == j
    assert same[acc].mean() > 0.999                      # exact ties between two targets aside
    q = tgt[r["corr"]].astype(np.float64); n = nrm[r["corr"]].astype(np.float64)
    a = r["accepted"]
    J = np.concatenate([np.cross(p[a], n[a]), n[a]], 1); res = ((p[a] - q[a]) * n[a]).sum(1)
    assert np.allclose(r["ATA"], J.T @ J, rtol=2e-3, atol=1e-4) and np.allclose(r["ATb"], J.T @ res, rtol=2e-3, atol=1e-5)
    one = orc.icp(src, tgt, nrm, T0, thr, 1, True)
    x = np.linalg.solve(J.T @ J, -(J.T @ res))
    cx, sx, cy, sy, cz, sz = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    D = np.eye(4); D[:3, :3] = Rx @ Ry @ Rz; D[:3, 3] = x[3:]
    assert np.allclose(one["T"], D @ T0.astype(np.float64), atol=2e-5)


def test_icp_sums_are_sequential_float32_sums_in_source_order(orc, synth):
    """What the GPU's reference-order mode is held to (tests/test_gpu_icp_reference_order.py) is itself pinned here, without the
    oracle's code: registration.cpp:340-341,353-354 add one correspondence after the other in float, in ascending source index.  numpy's
    cumsum IS that sequential float32 sum (add.reduce would be pairwise): its last element must equal the oracle's total_error, ATA and ATb
    bit for bit, the terms being single float32 operations in the oracle's expression order."""
    f = np.float32
    tgt, nrm = synth.sample_object(3000, 2)
    src, T_gt = synth.make_scene(2500, 2)
    T0 = synth.perturb(T_gt, angle_deg=0.4, trans=0.0008)
    thr = 0.004
    r = orc.icp_correspondences(src, tgt, nrm, T0, thr)
    a = r["accepted"]; assert a.sum() > 500
    R = T0[:3, :3].astype(f); t = T0[:3, 3].astype(f)
    s = src.astype(f)
    p = np.stack([(R[k, 0] * s[:, 0] + (R[k, 1] * s[:, 1] + R[k, 2] * s[:, 2])) + t[k] for k in range(3)], 1).astype(f)     # c0 + (c1 + c2), then + t
    q = tgt[r["corr"]].astype(f); n = nrm[r["corr"]].astype(f)
    d = p - q
    d2 = (d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])).astype(f)
    assert d2.tobytes() == r["d2"].tobytes()
    J = np.stack([p[:, 1] * n[:, 2] - p[:, 2] * n[:, 1], p[:, 2] * n[:, 0] - p[:, 0] * n[:, 2], p[:, 0] * n[:, 1] - p[:, 1] * n[:, 0], n[:, 0], n[:, 1], n[:, 2]], 1).astype(f)
    res = (d[:, 0] * n[:, 0] + (d[:, 1] * n[:, 1] + d[:, 2] * n[:, 2])).astype(f)
    seq = lambda v: np.cumsum(np.concatenate([[f(0)], v[a].astype(f)]), dtype=f)[-1]          # 0 + v0 + v1 + ... one rounding per step
    assert f(seq(d2)).tobytes() == f(r["total_error"]).tobytes()
    for i in range(6):
        assert f(seq(J[:, i] * res)).tobytes() == f(r["ATb"][i]).tobytes(), i
        for k in range(6):
            assert f(seq(J[:, i] * J[:, k])).tobytes() == f(r["ATA"][i, k]).tobytes(), (i, k)
    # and a pairwise (tree) sum of the same terms is NOT that number: the order is what the test pins
    assert f(np.add.reduce((J[:, 0] * J[:, 0])[a].astype(f))).tobytes() != f(r["ATA"][0, 0]).tobytes() or a.sum() < 64


def test_ransac_hypothesis_and_score_against_numpy(orc, synth):
    tgt, _ = synth.sample_object(2500, 4)
    src, T_gt = synth.make_scene(2000, 4)
    rng = np.random.default_rng(0)
    corr = rng.integers(0, 2500, 2000).astype(np.int32)
    good = rng.random(2000) < 0.6
    p = src.astype(np.float64) @ T_gt[:3, :3].astype(np.float64).T + T_gt[:3, 3]
    corr[good] = cKDTree(tgt.astype(np.float64)).query(p[good])[1]
    voxel = 0.004
    r = orc.ransac(src, tgt, corr=corr, voxel=voxel, max_iterations=300, confidence=2.0, trace=True)
    tri = orc.sample_triples(2000, 300).astype(np.int64)
    checked = 0
    for it in range(300):
        i = tri[it]
        if len(set(i.tolist())) < 3:
            assert r["inliers"][it] == -1
            continue
        S = src[i].astype(np.float64); Q = tgt[corr[i]].astype(np.float64)
        Sc = S - S.mean(0); Qc = Q - Q.mean(0)
        H = Sc.T @ Qc
        if np.linalg.svd(H, compute_uv=False)[1] < 1e-6 * np.linalg.svd(H, compute_uv=False)[0]:
            continue                                     # nearly collinear triple: the rotation about the line is arbitrary
        U, _, Vt = np.linalg.svd(H)
        R = Vt.T @ np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))]) @ U.T
        t = Q.mean(0) - R @ S.mean(0)
        err = np.linalg.norm(src.astype(np.float64) @ R.T + t - tgt[corr].astype(np.float64), axis=1)
        near = int((np.abs(err - voxel * 1.5) < 2e-5).sum())           # float32 vs float64 at the threshold
        assert abs(int((err < voxel * 1.5).sum()) - int(r["inliers"][it])) <= near, it
        checked += 1
    assert checked > 250


def test_fpfh_against_float64_numpy(orc, synth):
    """computeFPFH (registration.cpp:133-201) written a second time — vectorised numpy in float64 over scipy's radius
    neighbourhoods — against the oracle's float32 loops.  A pair whose angle feature falls within rounding distance of a bin
    edge may land in the neighbouring bin (float32 vs float64), so the comparison is on the descriptors' L1 distance."""
    pts, _ = synth.sample_object(1500, 9)
    nrm = orc.estimate_normals(pts, 30)
    radius = 5.0 * float(synth.mean_spacing(1500))
    got = orc.compute_fpfh(pts, nrm, radius).astype(np.float64)
    P = pts.astype(np.float64); N = nrm.astype(np.float64)
    tree = cKDTree(P)
    nb = []
    for i, lst in enumerate(tree.query_ball_point(P, radius * (1 + 1e-9))):   # findRadiusNN: the 100 nearest in (d2, index) order
        lst = np.array(sorted(lst), int)
        d2 = ((P[lst] - P[i]) ** 2).sum(1)
        nb.append(list(lst[np.lexsort((lst, d2))][:100]))
    spfh = np.zeros((len(P), 33))
    for i, lst in enumerate(nb):
        j = np.array([q for q in lst if q != i], int)
        if len(j) == 0: continue
        d = P[j] - P[i]; dist = np.linalg.norm(d, axis=1); keep = dist >= 1e-8
        j, d, dist = j[keep], d[keep] / dist[keep, None], dist[keep]
        u = N[i]; v = np.cross(u, d); w = np.cross(u, v)
        alpha = (v * N[j]).sum(1); phi = d @ u; theta = np.arctan2((w * N[j]).sum(1), N[j] @ u)
        for off, val in ((0, alpha + 1.0), (11, phi + 1.0), (22, theta / np.pi + 1.0)):
            np.add.at(spfh[i], off + np.clip((val * 5.5).astype(int), 0, 10), 1.0)
        spfh[i] /= spfh[i].sum()
    ref = np.zeros_like(spfh)
    for i, lst in enumerate(nb):
        j = np.array([q for q in lst if q != i], int)
        f = spfh[i].copy()
        if len(j):
            dist = np.linalg.norm(P[j] - P[i], axis=1); keep = dist >= 1e-8
            f += (spfh[j[keep]] / dist[keep, None]).sum(0)
        ref[i] = f / f.sum() if f.sum() > 0 else f
    l1 = np.abs(got - ref).sum(1)
    assert np.median(l1) < 1e-5 and (l1 < 2e-2).all() and (l1 < 1e-3).mean() > 0.9, (np.median(l1), l1.max(), (l1 < 1e-3).mean())


def test_descriptor_match_against_scipy_kdtree(orc, synth):
    """ransacRegistration's correspondence step (registration.cpp:216-232: nearest target descriptor by the 33-term float sum,
    lowest index on ties) against scipy's k-d tree in float64: the same match wherever the two nearest distances differ clearly."""
    fs = synth.random_features(1200, 5); ft = synth.random_features(900, 6)
    got = orc.feature_match(fs, ft)
    d, idx = cKDTree(ft.astype(np.float64)).query(fs.astype(np.float64), 2)
    clear = (d[:, 1] - d[:, 0]) > 1e-6 * np.maximum(d[:, 1], 1e-12)
    assert clear.mean() > 0.95 and np.array_equal(got[clear], idx[clear, 0])
    dd = np.linalg.norm(fs.astype(np.float64) - ft[got].astype(np.float64), axis=1)      # and never worse than the best by more than rounding
    assert (dd <= d[:, 0] * (1 + 1e-5) + 1e-9).all()
