"""The batch's shapes against each other.  tdv_register_batch_dev picks, by instance size, between walking every instance through the
chain on lanes (round 2's shape, which tests/test_gpu_chain.py / test_gpu_configs.py hold against the oracle and the operator chain)
and running stages over the whole batch (voxels of all clouds + the container order on the device, confined neighbour searches, one
descriptor match, batched RANSAC with the device index sampler, the one-launch ICP).  Every combination of those stages must return the
SAME bits for the same inputs: random frames with instances of every kind - empty labels, a handful of pixels (fewer voxels than the
normals' k: the batched feature stage must step aside), hundreds to thousands, one large enough to leave the small-problem kernels - in
all three mask formats and both voxel orders."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
W, H, F = 640, 480, 600.0
KNOBS = ("TDV_BATCH_STAGED", "TDV_BATCH_VOXEL", "TDV_VOXEL_PIXELS", "TDV_VOXEL_DEVICE_ORDER", "TDV_BATCH_FEATURES", "TDV_RANSAC_BATCH", "TDV_ICP_SMALL", "TDV_BATCH_LANES")


def _surface(rng, yy, xx):
    """A bumpy depth surface (metres) over pixel coordinates."""
    z = 0.5 + 0.0004 * (xx - W / 2) * rng.normal() + 0.0004 * (yy - H / 2) * rng.normal()
    for _ in range(6):
        cx, cy, s, a = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(8, 40), rng.uniform(-0.01, 0.01)
        z = z + a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return z


def _scene(seed, with_big):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    depth = np.zeros((H, W), np.uint16)
    label = np.zeros((H, W), np.uint16)
    kinds = ["empty", "tiny", "tiny2", "mid", "mid", "mid", "mid", "wide"] + (["big"] if with_big else [])
    rng.shuffle(kinds)
    boxes = []
    x0 = 4
    for b, kind in enumerate(kinds):
        w, h = {"empty": (0, 0), "tiny": (5, 4), "tiny2": (9, 7), "mid": (int(rng.integers(18, 45)), int(rng.integers(18, 45))),
                "wide": (70, 26), "big": (190, 170)}[kind]
        if x0 + w + 4 >= W:
            break
        y0 = int(rng.integers(4, H - max(h, 1) - 4))
        if w:
            m = np.zeros((H, W), bool); m[y0:y0 + h, x0:x0 + w] = True
            if kind in ("mid", "wide", "big"):                                   # an ellipse, so that rows start and end at other columns
                m &= ((xx - (x0 + w / 2)) / (w / 2)) ** 2 + ((yy - (y0 + h / 2)) / (h / 2)) ** 2 <= 1.0
            label[m] = b + 1
            z = _surface(rng, yy, xx)
            depth[m] = np.round(z[m] * 10000.0).astype(np.uint16)
            if kind == "mid" and rng.random() < 0.5:                             # a few pixels beyond the clip and a few zeros inside the mask
                ys, xs = np.nonzero(m)
                sel = rng.choice(len(ys), 5, replace=False)
                depth[ys[sel[:3]], xs[sel[:3]]] = 60000; depth[ys[sel[3:]], xs[sel[3:]]] = 0
        boxes.append(kind)
        x0 += w + 6
    return depth, label, boxes


def _run(ctx, tdv, depth_t, masks_t, fmt, n_inst, model, voxel, order, env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        prm = tdv.batch_params(width=W, height=H, scale_to_meters=10000.0, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, zmax=1.5, voxel_size=voxel,
                               ransac_max_iterations=1500, icp_max_iterations=12, voxel_order=order, mask_format=fmt)
        d_mx, d_mn, d_mf, nm = model
        return ctx.register_batch_dev(depth_t.data_ptr(), None, masks_t.data_ptr(), n_inst, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
    finally:
        for k in KNOBS:
            os.environ.pop(k, None)


def _same(a, b, what):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        for key in ("status", "n_points", "n_voxels", "coarse_inliers", "icp_iterations"):
            assert x[key] == y[key], (what, i, key, x[key], y[key])
        assert x["T"].tobytes() == y["T"].tobytes(), (what, i, "T")
        for key in ("fitness", "rmse", "coarse_fitness"):
            assert np.float32(x[key]).tobytes() == np.float32(y[key]).tobytes(), (what, i, key, x[key], y[key])


@pytest.mark.parametrize("seed,with_big,model_n", [(1, False, 900), (2, True, 900), (3, False, 2600), (4, True, 700), (5, False, 1500)])
def test_batch_shapes_agree(ctx, tdv, synth, seed, with_big, model_n):
    dev = torch.device("cuda", 0)
    depth, label, kinds = _scene(seed, with_big)
    n_inst = len(kinds)
    voxel = 0.5 / F * 1.3
    # the model: a bumpy patch, prepared on the device (voxel -> normals -> FPFH)
    rng = np.random.default_rng(100 + seed)
    side = int(np.sqrt(model_n * 1.6))
    gy, gx = np.mgrid[0:side, 0:side].astype(np.float64)
    mz = 0.5 + 0.004 * np.sin(gx * 0.31) * np.cos(gy * 0.27) + 0.002 * rng.normal(size=gx.shape) * 0
    mpts = np.stack([(gx - side / 2) * 0.5 / F, (gy - side / 2) * 0.5 / F, mz], -1).reshape(-1, 3).astype(np.float32)
    stacked = np.stack([np.where(label == b + 1, 255, 0).astype(np.uint8) for b in range(n_inst)])
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    formats = {0: torch.from_numpy(stacked).to(dev), 1: torch.from_numpy(label.astype(np.uint8)).to(dev), 2: torch.from_numpy(label.view(np.int16)).to(dev)}
    for order in (tdv.TDV_VOXEL_ORDER_REFERENCE, tdv.TDV_VOXEL_ORDER_FIRST):
        d_raw = torch.from_numpy(mpts).to(dev)
        d_mx = torch.empty_like(d_raw); d_mn = torch.empty_like(d_raw); d_mf = torch.empty((len(mpts), 33), dtype=torch.float32, device=dev)
        nm = ctx.prepare_model_dev(d_raw.data_ptr(), len(mpts), voxel, 30, 5.0, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), order=order)
        model = (d_mx, d_mn, d_mf, nm)
        base = _run(ctx, tdv, d_depth, formats[0], 0, n_inst, model, voxel, order, dict(TDV_BATCH_STAGED="0", TDV_BATCH_VOXEL="0"))   # round 2's shape
        assert [r["status"] for r in base] == [1 if k == "empty" else 0 for k in kinds]
        assert any(0 < r["n_voxels"] < 30 for r in base) and any(r["n_voxels"] > 300 for r in base)
        variants = [
            ("default", 0, {}), ("default, u8 labels", 1, {}), ("default, u16 labels", 2, {}),
            ("staged, everything batched", 2, dict(TDV_BATCH_STAGED="1")),
            ("staged, 1 lane", 0, dict(TDV_BATCH_STAGED="1", TDV_BATCH_LANES="1")),
            ("staged, RANSAC per instance", 2, dict(TDV_BATCH_STAGED="1", TDV_RANSAC_BATCH="0")),
            ("staged, features per instance", 0, dict(TDV_BATCH_STAGED="1", TDV_BATCH_FEATURES="0")),
            ("staged, ICP per iteration", 2, dict(TDV_BATCH_STAGED="1", TDV_ICP_SMALL="0")),
            ("staged, host replay of the order", 1, dict(TDV_BATCH_STAGED="1", TDV_VOXEL_DEVICE_ORDER="0")),
            ("staged, voxels through the table", 2, dict(TDV_BATCH_STAGED="1", TDV_VOXEL_PIXELS="0")),
            ("instance by instance, voxels through the table", 1, dict(TDV_BATCH_STAGED="0", TDV_VOXEL_PIXELS="0")),
            ("instance by instance, voxels batched", 2, dict(TDV_BATCH_STAGED="0")),
            ("instance by instance, ICP per iteration", 0, dict(TDV_BATCH_STAGED="0", TDV_ICP_SMALL="0", TDV_BATCH_VOXEL="0")),
        ]
        for what, fmt, env in variants:
            _same(_run(ctx, tdv, d_depth, formats[fmt], fmt, n_inst, model, voxel, order, env), base, "%s (order %d, seed %d)" % (what, order, seed))
            if env.get("TDV_BATCH_VOXEL") != "0":
                assert ctx.last_voxel_grouping() == ("table" if env.get("TDV_VOXEL_PIXELS") == "0" else "pixels"), (what, ctx.last_voxel_grouping())
            if "TDV_BATCH_LANES" in env:
                assert ctx.last_batch_lanes() == int(env["TDV_BATCH_LANES"]), (what, ctx.last_batch_lanes())     # the knob took effect
            elif n_inst > 1:
                assert ctx.last_batch_lanes() > 1, (what, ctx.last_batch_lanes())
        print("seed %d order %d: %d instances %s, model %d pts: %d variants equal; voxels %s" % (seed, order, n_inst, kinds, nm, len(variants), [r["n_voxels"] for r in base]))
