"""R3 in the batch: voxel grouping through pixel windows (csrc/voxel.hip: k_vs_group) - the path tdv_register_batch_dev takes for clouds
it unprojected itself - against the hash-table path (k_vh_insert), which earlier rounds hold against the oracle, and against the oracle
directly (reference src/registration.cpp:29-60: means summed in ascending input index; first-occurrence order = leaders in input order).

The pixel-window argument (two points of one voxel are less than fx * (s / z) * (1 + |x / z|) pixels apart) only covers clouds in row-major
pixel order with fine enough voxels and rows that fit the halo; everything else must be HANDED OVER to the table inside the call with
the same results: coarse voxels, full-width rows, points not in pixel order, huge key ranges, voxels with more members than a row holds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
W, H, F, CX, CY, SCALE, ZMAX = 1280, 720, 900.0, 640.0, 360.0, 10000.0, 1.5
CAM = (F, F, CX, CY)


def _frame(seed, n_inst, box, tilt=1.0, base=0.45, steps=False):
    """A depth frame (uint16, 0.1 mm units) with n_inst rectangular instances of box = (w, h) pixels on bumpy, tilted surfaces; a u16 label image."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    depth = np.zeros((H, W), np.uint16); label = np.zeros((H, W), np.uint16)
    bw, bh = box
    cols = max(1, (W - 8) // (bw + 6))
    for b in range(n_inst):
        x0 = 4 + (b % cols) * (bw + 6); y0 = 4 + (b // cols) * (bh + 6)
        if y0 + bh >= H - 4:
            break
        tl = tilt * min(1.0, 150.0 / max(bw, bh))                                                 # (the surface stays within ~5 cm of `base` whatever the box)
        z = base + 0.0003 * tl * (xx - x0) * rng.normal() + 0.0003 * tl * (yy - y0) * rng.normal()
        for _ in range(5):
            cx, cy, s, a = rng.uniform(x0, x0 + bw), rng.uniform(y0, y0 + bh), rng.uniform(4, 30), rng.uniform(-0.004, 0.004)
            z = z + a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        if steps:
            z = z + 0.02 * ((xx.astype(np.int64) // 7 + yy.astype(np.int64) // 5) % 3)            # depth terraces: discontinuities inside the instance
        m = np.zeros((H, W), bool); m[y0:y0 + bh, x0:x0 + bw] = True
        if b % 3 == 1:                                                                           # an ellipse: rows start and end at other columns
            m &= ((xx - (x0 + bw / 2)) / (bw / 2)) ** 2 + ((yy - (y0 + bh / 2)) / (bh / 2)) ** 2 <= 1.0
        label[m] = b + 1
        depth[m] = np.clip(np.round(z[m] * SCALE), 1, 65535).astype(np.uint16)
        if b % 4 == 2:                                                                           # holes and pixels beyond the clip inside the mask
            ys, xs = np.nonzero(m); sel = rng.choice(len(ys), max(4, len(ys) // 50), replace=False)
            depth[ys[sel[::2]], xs[sel[::2]]] = 0; depth[ys[sel[1::2]], xs[sel[1::2]]] = 60000
    return depth, label, int(label.max())


def _clouds(ctx, depth, label, n_inst):
    dev = torch.device("cuda", 0)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev); d_label = torch.from_numpy(label.view(np.int16)).to(dev)
    cap = int((label > 0).sum())
    d_xyz = torch.empty((max(cap, 1), 3), dtype=torch.float32, device=dev)
    off = ctx.depth_to_cloud_batch_dev(d_depth.data_ptr(), d_label.data_ptr(), None, n_inst, W, H, SCALE, F, F, CX, CY, ZMAX, d_xyz.data_ptr(), None, cap, mask_format=2)
    return d_xyz, off


def _both(ctx, d_xyz, off, voxel, cam=CAM):
    a = torch.empty_like(d_xyz); b = torch.empty_like(d_xyz)
    va = ctx.voxel_downsample_batch_dev(d_xyz.data_ptr(), off, voxel, a.data_ptr())
    assert ctx.last_voxel_grouping() == "table"
    vb = ctx.voxel_downsample_batch_dev(d_xyz.data_ptr(), off, voxel, b.data_ptr(), pinhole=cam)
    how = ctx.last_voxel_grouping()
    assert np.array_equal(va, vb), (va[:8], vb[:8])
    n = int(va[-1])
    assert a[:n].cpu().numpy().tobytes() == b[:n].cpu().numpy().tobytes()
    return va, a[:n].cpu().numpy(), how


@pytest.mark.parametrize("seed,n_inst,box,voxel_px,steps", [(1, 12, (200, 150), 1.2, False), (2, 40, (60, 45), 1.2, False), (3, 300, (25, 25), 1.2, False),
                                                            (4, 6, (448, 300), 1.2, False), (5, 12, (200, 150), 1.5, False), (6, 12, (200, 150), 0.8, True),
                                                            (7, 3, (600, 500), 1.6, True)])
def test_pixel_windows_equal_the_table_and_the_oracle(ctx, orc, seed, n_inst, box, voxel_px, steps):
    depth, label, n_inst = _frame(seed, n_inst, box, steps=steps)
    d_xyz, off = _clouds(ctx, depth, label, n_inst)
    voxel = float(np.float32(voxel_px * 0.45 / F))
    voff, vox, how = _both(ctx, d_xyz, off, voxel)
    assert how == "pixels", how
    xyz = d_xyz.cpu().numpy()
    for b in (0, n_inst // 2, n_inst - 1):                                      # sampled instances against the oracle: means, count, first-occurrence order
        pts = xyz[off[b]:off[b + 1]]
        ref_xyz, _, first = orc.voxel_downsample(pts, None, voxel)
        order = np.argsort(first, kind="stable")
        assert vox[voff[b]:voff[b + 1]].tobytes() == ref_xyz[order].tobytes(), b
    print("seed %d: %d instances, %d points -> %d voxels (%.2f points per voxel), grouped by %s" % (seed, n_inst, off[-1], voff[-1], off[-1] / max(voff[-1], 1), how))


@pytest.mark.parametrize("what", ["coarse_voxels", "full_width_rows", "not_pixel_order", "wrong_intrinsics", "many_members", "huge_key_range", "empty_clouds"])
def test_hand_over_to_the_table(ctx, orc, what):
    """Cases the window argument does not cover: the call must notice and give the table path's result."""
    rng = np.random.default_rng(11)
    voxel = float(np.float32(1.2 * 0.45 / F)); cam = CAM; expect = "table"
    if what == "full_width_rows":
        depth, label, n_inst = _frame(21, 1, (1270, 60))                        # rows of 1,270 points: two of them exceed the 2,048-point halo
    elif what == "empty_clouds":
        depth, label, n_inst = _frame(22, 9, (120, 90)); label[label == 4] = 0; label[label == 9] = 0; expect = "pixels"      # clouds 3 and 8 are empty (and the last one)
    else:
        depth, label, n_inst = _frame(20, 8, (200, 150))
    d_xyz, off = _clouds(ctx, depth, label, n_inst)
    if what == "coarse_voxels":
        voxel = 0.005                                                           # 10 pixels per voxel: windows of 21 x 21
    elif what == "not_pixel_order":
        x = d_xyz.cpu().numpy(); seg = x[off[2]:off[3]].copy(); rng.shuffle(seg); x[off[2]:off[3]] = seg
        d_xyz = torch.from_numpy(x).to(d_xyz.device)
    elif what == "wrong_intrinsics":
        cam = (F * 0.5, F * 0.5, CX + 300.0, CY)                                # several points land on one pixel / out of order
    elif what == "many_members":
        voxel = 0.0016                                                          # ~3 pixels per voxel, tilted surfaces: some voxels hold more than 16 points -> per-cloud legacy path
        expect = None
    elif what == "huge_key_range":
        voxel = 5e-7                                                            # a 0.1 m wide instance over a 0.5-um grid: the relative cells do not fit their 10 + 8 + 14 bits
    if what == "many_members":
        a = torch.empty_like(d_xyz); b = torch.empty_like(d_xyz)
        va = ctx.voxel_downsample_batch_dev(d_xyz.data_ptr(), off, voxel, a.data_ptr()); vb = ctx.voxel_downsample_batch_dev(d_xyz.data_ptr(), off, voxel, b.data_ptr(), pinhole=cam)
        assert np.array_equal(va, vb) and a[:va[-1]].cpu().numpy().tobytes() == b[:vb[-1]].cpu().numpy().tobytes()
        return
    voff, vox, how = _both(ctx, d_xyz, off, voxel, cam)
    assert how == expect, (what, how)
    if what == "empty_clouds":
        assert voff[4] == voff[3] and voff[9] == voff[8]
    xyz = d_xyz.cpu().numpy()
    b = min(2, n_inst - 1)
    ref_xyz, _, first = orc.voxel_downsample(xyz[off[b]:off[b + 1]], None, voxel)
    assert vox[voff[b]:voff[b + 1]].tobytes() == ref_xyz[np.argsort(first, kind="stable")].tobytes()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_random_frames(ctx, seed):
    """Random surfaces, boxes, voxel sizes and intrinsics: whichever way a call goes, it returns the table path's bytes."""
    rng = np.random.default_rng(100 + seed)
    box = (int(rng.integers(10, 500)), int(rng.integers(10, 300)))
    depth, label, n_inst = _frame(200 + seed, int(rng.integers(1, 60)), box, tilt=float(rng.uniform(0, 4)), base=float(rng.uniform(0.25, 1.2)), steps=bool(seed % 2))
    d_xyz, off = _clouds(ctx, depth, label, n_inst)
    for voxel_px in (0.7, 1.2, 1.9, 3.3):
        voxel = float(np.float32(voxel_px * 0.45 / F))
        voff, _, how = _both(ctx, d_xyz, off, voxel)
        print("seed %d box %s voxel %.1f px: %d points -> %d voxels by %s" % (seed, box, voxel_px, off[-1], voff[-1], how))
