"""CPU suite: the C-ABI library loads without a GPU, exports every symbol include/tdv_hip.h declares,
and its host-side pieces (index stream, pose composition, error behaviour) are correct.
No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_is_built_and_exports_the_abi(tdv):
    assert os.path.exists(tdv.LIB_PATH), "run __graft_entry__.build()"
    lib = tdv.lib()
    header = open(os.path.join(ROOT, "include", "tdv_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|double|void|void\*|const char\*|unsigned long long)\s+(tdv_[a-z0-9_]+)\s*\(", header, re.M)))
    assert len(declared) >= 30
    for sym in declared:
        assert hasattr(lib, sym), "header declares %s but the library does not export it" % sym
    assert sorted(tdv.ABI_SYMBOLS) == declared, "ABI_SYMBOLS out of sync with the header"
    assert b"gfx950" in lib.tdv_version()


def test_index_stream_matches_libstdcpp(tdv, orc):
    """Own mt19937 + Lemire (ctx.hip) == std::mt19937 + std::uniform_int_distribution<size_t>
    (what src/registration.cpp:235-236 instantiates on libstdc++ 11)."""
    for n in (1, 2, 3, 7, 1600, 32129, 200000, 500000, 2 ** 31 - 1, 2 ** 31 + 12345, 2 ** 32 - 1, 2 ** 32):
        assert np.array_equal(tdv.sample_triples(n, 700), orc.sample_triples(n, 700)), n
    assert np.array_equal(tdv.sample_triples(1000, 50, seed=7), orc.sample_triples(1000, 50, seed=7))
    assert tdv.sample_triples(1600, 1).tolist() == [[599, 1274, 1521]]
    with pytest.raises(tdv.TdvError):
        tdv.sample_triples(0, 1)


def test_pose_compose(tdv, orc, synth):
    T = synth.gt_transform(3)
    E = np.array([[0.00705456, 0.99996948, -0.00335601, 0.43244419], [0.99984465, -0.00710781, -0.01612942, -0.03129219],
                  [-0.01615278, -0.0032417, -0.99986428, 0.39502932], [0, 0, 0, 1]], np.float32)  # config/pipeline_config.yaml:41-57
    got = tdv.pose_compose(E, T)
    assert np.array_equal(got, orc.pose_compose(E, T))
    assert np.allclose(got.astype(np.float64) @ T, E, atol=1e-6)
    with pytest.raises(tdv.TdvError):
        tdv.pose_compose(E, np.zeros((4, 4), np.float32))


def test_status_strings_and_no_device_behaviour(tdv):
    lib = tdv.lib()
    assert lib.tdv_status_string(0) == b"ok" and b"device" in lib.tdv_status_string(-1)
    n = tdv.device_count()
    assert n >= 0
    if n == 0:
        # reference behaviour without a GPU: isCudaAvailable() false; preprocess / icpRefine throw
        # "CUDA not available" (src/gpu_impl.cpp:64,258); generate returns an empty cloud (:126)
        assert not tdv.GPUDepth.isCudaAvailable() and not tdv.GPURegistration.isCudaAvailable()
        with pytest.raises(RuntimeError, match="CUDA not available"):
            tdv.GPUDepth.preprocess(np.zeros((2, 2), np.uint16), None, 1000.0)
        with pytest.raises(RuntimeError, match="CUDA not available"):
            tdv.GPURegistration.icpRefine(tdv.PointCloud(), tdv.PointCloud(), np.eye(4), 0.1)
        assert tdv.GPUPointCloud.generate(np.zeros((2, 2), np.float32), None, 1, 1, 0, 0).empty()
        with pytest.raises(tdv.TdvError):
            tdv.Context(0)  # no CPU fallback: the product fails loudly


def test_product_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "3dvision_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "oracle/" not in text.replace("# oracle/", ""), f


def test_filter_duplicates_matches_reference_semantics(tdv, orc):
    """src/pipeline.cpp:153-180: greedy, order-dependent; a duplicate replaces the kept pose only if it is
    closer to the origin; comparison is against the CURRENT kept pose."""
    rng = np.random.default_rng(3)
    poses = np.tile(np.eye(4, dtype=np.float32), (60, 1, 1))
    centers = rng.random((8, 3)).astype(np.float32) * 2
    poses[:, :3, 3] = centers[rng.integers(0, 8, 60)] + (rng.random((60, 3)).astype(np.float32) - 0.5) * 0.08
    got = tdv.filter_duplicates(poses, 0.1)
    ref = orc.filter_duplicates(poses, 0.1)
    assert got.shape == ref.shape and np.array_equal(got, ref) and 8 <= len(got) <= 20
    assert len(tdv.filter_duplicates(poses[:0], 0.1)) == 0
    two = poses[:2].copy(); two[0, :3, 3] = [1, 0, 0]; two[1, :3, 3] = [0.95, 0, 0]
    out = tdv.filter_duplicates(two, 0.1)
    assert len(out) == 1 and out[0, 0, 3] == np.float32(0.95)  # the closer one replaces the first


def test_ply_loader_keeps_the_reference_quirk(tdv, orc, tmp_path):
    """src/registration.cpp:430-440: the header loop swallows the first vertex line; the last read fails."""
    p = tmp_path / "model.ply"
    verts = [(0.1 * i, 0.2 * i, 0.3 * i, 10 * i, 20 * i, 30 * i) for i in range(1, 6)]
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
                 "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n"
                 + "".join("%g %g %g %d %d %d\n" % v for v in verts))
    cloud = tdv.load_reference_model(str(p))
    ref_xyz, ref_rgb = orc.load_ply(str(p))
    assert cloud.size() == 5 and cloud.hasColors()
    assert np.array_equal(cloud.points, ref_xyz) and np.array_equal(cloud.colors, ref_rgb)
    assert np.allclose(cloud.points[0], [0.2, 0.4, 0.6])          # vertex 0 was skipped
    assert np.array_equal(cloud.points[4], [0, 0, 0])             # the failed last read
    assert np.allclose(cloud.colors[0], np.array([20, 40, 60]) / 255.0)
    q = tmp_path / "nocolor.ply"
    q.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nend_header\n1 2 3\n4 5 6\n7 8 9\n")
    c2 = tdv.load_reference_model(str(q))
    assert c2.size() == 3 and not c2.hasColors() and np.array_equal(c2.points[:2], [[4, 5, 6], [7, 8, 9]])
    assert tdv.load_reference_model(str(tmp_path / "missing.ply")).empty()


def test_ply_loader_hostile_vertex_count_returns_at_once(tdv, tmp_path):
    """A header that promises 2^30 vertices over a 3-line body: the reference would push 2^30 garbage points; the loader
    reports that count without looping over it and fills what fits with zeros after the last real vertex."""
    import ctypes as C
    import time
    p = tmp_path / "liar.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 1073741824\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
                 "1 2 3\n4 5 6\n7 8 9\n")
    n = C.c_int(); hc = C.c_int()
    t0 = time.time()
    assert tdv.lib().tdv_load_ply_ascii(str(p).encode(), None, None, 0, C.byref(n), C.byref(hc)) == 0
    assert n.value == 1 << 30 and hc.value == 0
    xyz = np.full((8, 3), -1, np.float32)
    st = tdv.lib().tdv_load_ply_ascii(str(p).encode(), xyz.ctypes.data_as(C.c_void_p), None, 8, C.byref(n), C.byref(hc))
    assert st != 0 and n.value == 1 << 30                      # does not fit: TDV_ERR_BAD_ARG, needed count reported
    assert np.array_equal(xyz[:2], [[4, 5, 6], [7, 8, 9]]) and np.array_equal(xyz[2:], np.zeros((6, 3)))
    assert time.time() - t0 < 2.0


# ---- mask-directory loader (Segmentation::loadMasksFromDir, src/segmentation.cpp:12-42) -----------------------------

def _png_bytes(pix, depth=8, alpha=False, filters=None):
    """Encode a grey image (uint values < 2**depth, [h, w]) as a PNG with the given bit depth and per-row filter types,
    using only zlib: an encoder independent of the decoder under test."""
    import struct, zlib
    h, w = pix.shape
    ch = 2 if alpha else 1
    rows = []
    for y in range(h):
        if depth == 8:
            row = np.zeros((w, ch), np.uint8); row[:, 0] = pix[y]
            if alpha: row[:, 1] = 200
            raw = row.tobytes()
        elif depth == 16:
            row = np.zeros((w, ch), ">u2"); row[:, 0] = pix[y]
            if alpha: row[:, 1] = 40000
            raw = row.tobytes()
        else:
            bits = np.zeros(((w * depth + 7) // 8) * 8, np.uint8)
            for x in range(w):
                for b in range(depth):
                    bits[x * depth + b] = (int(pix[y, x]) >> (depth - 1 - b)) & 1
            raw = np.packbits(bits).tobytes()
        rows.append(bytearray(raw))
    bpp = max(1, depth * ch // 8)
    out = bytearray(); prev = bytearray(len(rows[0]))
    for y, cur in enumerate(rows):
        ft = filters[y % len(filters)] if filters else 0
        enc = bytearray(len(cur))
        for x in range(len(cur)):
            a = cur[x - bpp] if x >= bpp else 0; b = prev[x]; c = prev[x - bpp] if x >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                pp = a + b - c; pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            enc[x] = (cur[x] - p) & 255
        out.append(ft); out += enc; prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    ihdr = struct.pack(">IIBBBBB", w, h, depth, 4 if alpha else 0, 0, 0, 0)
    comp = zlib.compress(bytes(out), 6)
    idat = chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:])   # split IDAT on purpose
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"tEXt", b"k\x00v") + idat + chunk(b"IEND", b"")


@pytest.mark.parametrize("depth,alpha", [(8, False), (8, True), (1, False), (2, False), (4, False), (16, False), (16, True)])
def test_mask_png_decoding_and_threshold(tdv, tmp_path, depth, alpha):
    rng = np.random.default_rng(depth * 2 + alpha)
    h, w = 37, 53                                   # odd sizes: partial bytes at sub-byte depths
    pix = rng.integers(0, 2 ** depth, (h, w)).astype(np.uint32)
    pix[:5] = np.arange(w)[None, :] % (2 ** depth)  # smooth rows make every filter type non-trivial
    p = tmp_path / "m.png"
    p.write_bytes(_png_bytes(pix, depth, alpha, filters=[0, 1, 2, 3, 4]))
    got = tdv.load_mask_png(str(p))
    grey = {1: pix * 255, 2: pix * 85, 4: pix * 17, 8: pix, 16: pix >> 8}[depth]
    assert got is not None and got.shape == (h, w)
    assert np.array_equal(got, np.where(grey > 10, 255, 0).astype(np.uint8))


def test_masks_from_dir_order_and_skips(tdv, tmp_path):
    h, w = 24, 32
    d = tmp_path / "masks"; d.mkdir()
    imgs = {}
    for name in ["b_mask.png", "a_mask.PNG", "c_mask.png"]:
        pix = np.random.default_rng(len(imgs)).integers(0, 256, (h, w)).astype(np.uint32)
        (d / name).write_bytes(_png_bytes(pix, 8, False, filters=[4, 1]))
        imgs[name] = np.where(pix > 10, 255, 0).astype(np.uint8)
    (d / "wrong_size.png").write_bytes(_png_bytes(np.zeros((5, 5), np.uint32)))
    (d / "photo.jpg").write_bytes(b"\xff\xd8\xff\xe0 not decodable here")
    (d / "notes.txt").write_text("ignored by extension")
    masks, skipped = tdv.load_masks_from_dir(str(d), w, h)
    assert masks.shape == (3, h, w) and skipped == 2
    for got, name in zip(masks, sorted(imgs)):      # std::sort on the paths: byte order, "a_mask.PNG" first
        assert np.array_equal(got, imgs[name])
    none, sk = tdv.load_masks_from_dir(str(tmp_path / "missing"), w, h)
    assert none.shape == (0, h, w) and sk == 0
    assert tdv.load_mask_png(str(d / "photo.jpg")) is None


# ---- the exactness argument of the pruned searches, checked on the CPU in IEEE float32 ------------------------------

def _f32(x):
    return np.asarray(x, np.float32)


@pytest.mark.parametrize("scale", [1e-6, 1.0, 1e3, 1e15])
def test_box_lower_bound_never_exceeds_a_distance(scale):
    """csrc/knn.hip box_lower_bound / csrc/icp.hip point_box_lb: per-axis gap by ONE float subtraction, then the
    reference's expression gx*gx + (gy*gy + gz*gz).  Float -, *, + are monotone under round-to-nearest, so the bound
    can never exceed fl(d2) of a point inside the box — with no safety margin.  Millions of random cases, including
    points on the box faces, queries inside the box, tiny and huge magnitudes."""
    rng = np.random.default_rng(int(np.log10(scale) * 7) % 1000 + 5)
    n = 400000
    lo = _f32((rng.random((n, 3)) - 0.5) * 2 * scale)
    ext = _f32(rng.random((n, 3)) * scale * rng.choice([0.0, 1e-7, 1e-3, 0.3], (n, 1)))
    hi = _f32(lo + ext)
    u = rng.random((n, 3)); u[rng.random((n, 3)) < 0.2] = 0.0; u[rng.random((n, 3)) < 0.2] = 1.0   # faces and corners too
    t = np.minimum(np.maximum(_f32(lo + _f32(u) * (hi - lo)), lo), hi)                                 # a target inside the box
    q = _f32((rng.random((n, 3)) - 0.5) * 3 * scale)
    q[: n // 10] = t[: n // 10]                                                                         # queries inside / on the target
    g = np.maximum(_f32(0), np.maximum(_f32(lo - q), _f32(q - hi)))
    lb = _f32(g[:, 0] * g[:, 0]) + _f32(_f32(g[:, 1] * g[:, 1]) + _f32(g[:, 2] * g[:, 2]))
    for sign in (1, -1):   # (points[i] - query) in the neighbour searches, (query - target) in ICP: same squares
        d = _f32(sign * (t - q))
        d2 = _f32(d[:, 0] * d[:, 0]) + _f32(_f32(d[:, 1] * d[:, 1]) + _f32(d[:, 2] * d[:, 2]))
        assert lb.dtype == np.float32 and d2.dtype == np.float32
        assert (lb <= d2).all()


def test_descriptor_box_bound_never_exceeds_a_distance():
    """k_feature_match_pruned: the same argument in 33-D with the sequential summation order of the distance loop."""
    rng = np.random.default_rng(3)
    n = 60000
    lo = _f32(rng.random((n, 33)) * 0.1); hi = _f32(lo + _f32(rng.random((n, 33)) * rng.choice([0.0, 1e-6, 0.05], (n, 1))))
    t = np.minimum(np.maximum(_f32(lo + _f32(rng.random((n, 33))) * (hi - lo)), lo), hi)
    f = _f32(rng.random((n, 33)) * 0.15)
    lb = np.zeros(n, np.float32); d2 = np.zeros(n, np.float32)
    for d in range(33):
        g = np.maximum(_f32(0), np.maximum(_f32(lo[:, d] - f[:, d]), _f32(f[:, d] - hi[:, d])))
        lb = _f32(lb + _f32(g * g))
        diff = _f32(f[:, d] - t[:, d])
        d2 = _f32(d2 + _f32(diff * diff))
    assert (lb <= d2).all()


def test_mask_png_malformed_inputs_are_rejected(tdv, tmp_path):
    """Truncated files, corrupt zlib streams, wrong signatures and unsupported colour types return None (no crash)."""
    import struct, zlib
    good = _png_bytes(np.random.default_rng(0).integers(0, 256, (20, 30)).astype(np.uint32), 8, False, filters=[1, 4])
    cases = {
        "empty.png": b"",
        "sig_only.png": good[:8],
        "truncated.png": good[: len(good) // 2],
        "bad_sig.png": b"\x89PNX" + good[4:],
        "corrupt_idat.png": good[:60] + bytes(50) + good[110:],
    }
    # colour type 2 (RGB) is well-formed but unsupported here
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    rgb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(2 * (1 + 6)))) + chunk(b"IEND", b"")
    cases["rgb.png"] = rgb
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        assert tdv.load_mask_png(str(p)) is None, name
    assert tdv.load_mask_png(str(tmp_path / "does_not_exist.png")) is None


@pytest.mark.parametrize("offset", [0.0, 1.0, 40.0])
def test_ransac_fast_scoring_band_covers_the_two_arithmetics(offset):
    """csrc/ransac.hip RansacBand: the distance of a transformed point to its match evaluated the reference's way
    (mul, mul, mul, add, add, add per row; squared norm in the reference's order) and the fast pass's way (fused multiply-
    adds) differ by at most 12.3 u A + 3 u (D_ref + D_fma), A = max over the rows of |r0||px| + |r1||py| + |r2||pz| + |t|;
    the kernel's band E = 16 u (A_max + sqrt(tau)) is larger for every distance up to twice the threshold.  Millions of
    random cases in float32 (the fused ops emulated through float64: products of two floats are exact there), rotations
    with and without noise, points on top of their matches and far from them, coordinates offset from the origin."""
    rng = np.random.default_rng(17 + int(offset))
    n = 1500000
    u = 2.0 ** -24
    q_, _ = np.linalg.qr(rng.normal(size=(n // 1000, 3, 3)))
    R = np.repeat(q_, 1000, 0)[:n] + rng.normal(size=(n, 3, 3)) * rng.choice([0.0, 1e-7, 1e-3], (n, 1, 1))
    R = _f32(R)
    p = _f32((rng.random((n, 3)) - 0.5) * 0.6 + offset)
    t = _f32(rng.normal(size=(n, 3)) * 0.3 - (R.astype(np.float64) @ np.full(3, offset)) + offset)
    thr = 0.003
    pr = np.einsum("nij,nj->ni", R.astype(np.float64), p.astype(np.float64)) + t          # real arithmetic (f64 is exact enough here)
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    dist = thr * rng.choice([0.0, 0.5, 0.999, 1.0, 1.001, 2.0], n) * (1 + rng.normal(size=n) * 1e-5)
    q = _f32(pr + dirs * dist[:, None])

    def row_ref(c):
        a = _f32(R[:, c, 0] * p[:, 0]); b = _f32(R[:, c, 1] * p[:, 1]); cc = _f32(R[:, c, 2] * p[:, 2])
        return _f32(_f32(a + _f32(b + cc)) + t[:, c])

    def fma(a, b, c):
        return _f32(a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64))

    def row_fma(c):
        return fma(R[:, c, 0], p[:, 0], fma(R[:, c, 1], p[:, 1], fma(R[:, c, 2], p[:, 2], t[:, c])))

    dr = [_f32(row_ref(c) - q[:, c]) for c in range(3)]
    df = [_f32(row_fma(c) - q[:, c]) for c in range(3)]
    d2_ref = _f32(_f32(dr[0] * dr[0]) + _f32(_f32(dr[1] * dr[1]) + _f32(dr[2] * dr[2])))
    d2_fma = fma(df[0], df[0], fma(df[1], df[1], _f32(df[2] * df[2])))
    D_ref = np.sqrt(d2_ref.astype(np.float64)); D_fma = np.sqrt(d2_fma.astype(np.float64))
    A = (np.abs(R.astype(np.float64)) * np.abs(p.astype(np.float64))[:, None, :]).sum(2) + np.abs(t.astype(np.float64))
    A = A.max(1)
    bound = 12.3 * u * A + 3 * u * (D_ref + D_fma)
    ratio = np.abs(D_ref - D_fma) / np.maximum(bound, 1e-300)
    print("offset %g: largest |D_ref - D_fma| / bound = %.3f" % (offset, ratio.max()))
    assert ratio.max() <= 1.0
    # the kernel's band: A bounded through the largest |coordinate| of the cloud, distances up to twice the threshold
    P = np.abs(p).max()
    A_kernel = (np.abs(R.astype(np.float64)).sum(2) * P + np.abs(t.astype(np.float64))).max(1)
    E = 16 * u * (A_kernel + thr)
    near = np.maximum(D_ref, D_fma) <= 2 * thr
    assert (np.abs(D_ref - D_fma)[near] <= E[near]).all()


def test_ransac_bailout_scheme_is_exact_on_simulated_counts():
    """RansacPlan (csrc/ransac.hip) in plain Python on random inlier tables: the host loop fed with PARTIAL counts for the
    hypotheses the scheme drops returns what the reference's loop (registration.cpp:281-290) returns on the true counts - best
    iteration, best count, iterations run - for random batch sizes, prefixes, confidences, skipped iterations and heavy ties."""
    rng = np.random.default_rng(11)

    def reference_loop(true, valid, ns, confidence):
        best_f, best_it, best_c, run = np.float32(0), -1, 0, 0
        for it in range(len(true)):
            run = it + 1
            if not valid[it]: continue
            f = np.float32(true[it]) / np.float32(ns)
            if f > best_f: best_f, best_it, best_c = f, it, int(true[it])
            if f > np.float32(confidence): break
        return best_it, best_c, run

    def bailout_loop(inl, valid, ns, confidence, first, batch, drop_permille):
        # inl[h, i] = 1 if point i is an inlier of hypothesis h; the device keeps `best` = max count seen in EARLIER batches
        n_h = len(inl); best_dev = 0
        best_f, best_it, best_c, run = np.float32(0), -1, 0, 0
        it0 = 0; stop = False
        while it0 < n_h and not stop:
            cnt = min(first if it0 == 0 else batch, n_h - it0)
            rest = best_dev - max(best_dev * drop_permille // 1000, 1)
            split = ns if rest < ns // 8 else max(0, ns - rest)            # points in phase 1 (k_ransac_plan; chunking left out)
            counts = inl[it0:it0 + cnt, :split].sum(1)
            left = ns - split
            survive = valid[it0:it0 + cnt] & (left > 0) & (counts + left > best_dev)      # k_ransac_select
            counts = counts + np.where(survive, inl[it0:it0 + cnt, split:].sum(1), 0)     # phase 2
            for k in range(cnt):                                                         # the host loop of ransac_run_dev
                run = it0 + k + 1
                if not valid[it0 + k]: continue
                f = np.float32(counts[k]) / np.float32(ns)
                if f > best_f: best_f, best_it, best_c = f, it0 + k, int(counts[k])
                if f > np.float32(confidence): stop = True; break
            v = counts[valid[it0:it0 + cnt]]
            if len(v): best_dev = max(best_dev, int(v.max()))                             # k_ransac_best
            it0 += cnt
        return best_it, best_c, run

    for trial in range(300):
        ns = int(rng.integers(8, 200)); n_h = int(rng.integers(1, 400))
        quality = rng.choice([0.02, 0.3, 0.6, 0.95], size=n_h, p=[0.6, 0.2, 0.15, 0.05]) * rng.uniform(0.5, 1.0)
        inl = (rng.random((n_h, ns)) < quality[:, None]).astype(np.int64)
        if trial % 3 == 0: inl[rng.integers(0, n_h, n_h // 2)] = inl[rng.integers(0, n_h)]    # many exact ties
        valid = rng.random(n_h) > 0.1
        confidence = float(rng.choice([2.0, 0.9, 0.6, 0.3, 0.05, -1.0]))
        true = inl.sum(1)
        want = reference_loop(true, valid, ns, confidence)
        got = bailout_loop(inl, valid, ns, confidence, int(rng.integers(1, 40)), int(rng.integers(1, 120)), int(rng.choice([5, 100, 500])))
        assert got == want, (trial, got, want)


def test_every_environment_switch_is_documented():
    """INTEGRATION.md lists every TDV_* variable the product library reads (getenv) and, in its own table, every one only the study
    library reads (study_env): a switch nobody can find is a trap.  The product library's list stays short (round 3 had 34)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    product_doc, study_doc = doc.split("### The study library")
    knobs, study = set(), set()
    for f in glob.glob(os.path.join(root, "3dvision_amd", "csrc", "*.h*")):
        src = open(f).read()
        knobs.update(re.findall(r'(?<!_)getenv\("(TDV_[A-Z0-9_]+)"\)', src))
        study.update(re.findall(r'study_env\("(TDV_[A-Z0-9_]+)"\)', src))
    assert 8 <= len(knobs) <= 16, sorted(knobs)
    assert len(study) >= 15
    missing = sorted(k for k in knobs if k not in product_doc) + sorted(k for k in study - knobs if k not in study_doc)
    assert not missing, missing
