"""Per-operator measurements of the hot path on one MI355X, inputs resident in HBM (the *_dev ABI).  Shared by
bench.py (the `operators` block after the timed region: driver-run numbers for every stage, not only the headline)
and tools/bench_ops.py.  Every figure is the MEDIAN of >= 3 synchronized repetitions after a warm-up.

Each entry: {"op", "workload", "ms", "bound", ...}.  HBM-bound scans carry algorithmic bytes and the
fraction of the 8 TB/s roofline; the brute-force-shaped kernels carry f32 VALU lane-ops and the fraction of 78.6 Tops/s;
the exact pruned searches carry their brute-force-equivalent rate (no roofline fraction: they skip most of the work)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0
VALU_PEAK_TOPS = 78.6


def median_ms(fn, torch, reps=3, warm=2):   # two warm-up calls: the ctx arena grows on the first call of a size and is coalesced on the second
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


def kernel_ms(ctx, slot, fn, torch, reps=3, warm=1):
    """Median wall ms and mean HIP-event ms of the slot's kernels per call."""
    for _ in range(warm):
        fn()
    ctx.timing_enable(True); ctx.timing_read(slot)
    wall = median_ms(fn, torch, reps=reps, warm=0)
    ms, launches = ctx.timing_read(slot)
    ctx.timing_enable(False)
    return wall, ms / reps, launches // max(reps, 1)


def hbm(entry, nbytes, ms):
    entry.update(bound="hbm", algorithmic_bytes=int(nbytes), achieved_GBps=nbytes / (ms * 1e-3) / 1e9, peak_GBps=HBM_PEAK_GBPS,
                 frac=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)
    return entry


def valu(entry, ops, ms):
    entry.update(bound="valu_f32", lane_ops=float(ops), achieved_Tops=ops / (ms * 1e-3) / 1e12, peak_Tops=VALU_PEAK_TOPS,
                 frac=ops / (ms * 1e-3) / 1e12 / VALU_PEAK_TOPS)
    return entry


def pruned(entry, brute_ops, ms, what):
    entry.update(bound="exact pruned search (%s)" % what, bruteforce_equivalent_Tops=brute_ops / (ms * 1e-3) / 1e12)
    return entry


def cuboid_scene(synth, n, seed=42):
    pts, nrm = synth.sample_object(n, seed)
    T = synth.gt_transform(seed)
    Tinv = np.linalg.inv(T.astype(np.float64))
    cam = (pts.astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
    return cam, pts, nrm


def depth_ops(ctx, tdv, torch, dev, reps=5):
    out = []
    h, w = 720, 1280
    rng = np.random.default_rng(0)
    raw = (800 + rng.integers(0, 200, (h, w))).astype(np.uint16)
    mask = np.zeros((h, w), np.uint8); mask[136:584, 416:864] = 255
    d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_mask = torch.from_numpy(mask).to(dev)
    d_xyz = torch.empty((h * w, 3), dtype=torch.float32, device=dev)
    cnt = [0]

    def f():
        cnt[0] = ctx.depth_to_cloud_dev(d_raw.data_ptr(), d_mask.data_ptr(), None, w, h, 1000.0, 900, 900, 640, 360, 1.5, d_xyz.data_ptr(), None, h * w)
    wall, kms, _ = kernel_ms(ctx, tdv.TIMER_DEPTH, f, torch, reps=reps, warm=2)
    out.append(hbm(dict(op="depth_to_cloud", workload="one 1280x720 frame, 448x448 mask -> %d points" % cnt[0], ms=wall, kernels_ms=kms,
                        note="launch-bound: 3 small launches + one 4-byte read-back"), 2 * 3 * h * w + 12 * cnt[0], kms))
    B = 256
    masks = np.zeros((B, h, w), np.uint8)
    for b in range(B):
        y0 = (b * 37) % (h - 448); x0 = (b * 101) % (w - 448)
        masks[b, y0:y0 + 448, x0:x0 + 448] = 255
    d_masks = torch.from_numpy(masks).to(dev)
    cap = B * 448 * 448
    d_all = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    off = [None]

    def g():
        off[0] = ctx.depth_to_cloud_batch_dev(d_raw.data_ptr(), d_masks.data_ptr(), None, B, w, h, 1000.0, 900, 900, 640, 360, 1.5, d_all.data_ptr(), None, cap)
    wall, kms, _ = kernel_ms(ctx, tdv.TIMER_DEPTH, g, torch, reps=reps, warm=2)
    npts = int(off[0][-1])
    out.append(hbm(dict(op="depth_to_cloud_batch", workload="256 stacked 448x448 masks of one 1280x720 frame -> %d points" % npts, ms=wall, kernels_ms=kms),
                   tdv.DEPTH_BATCH_MASK_PASSES * B * h * w + 2 * h * w + 12 * npts, kms))
    return out


def cloud_ops(ctx, tdv, synth, torch, dev, n, reps=3, want_match=True):
    out = []
    cam, mdl, _ = cuboid_scene(synth, n)
    d_xyz = torch.from_numpy(cam).to(dev)
    spacing = float(synth.mean_spacing(n))
    d_out = torch.empty_like(d_xyz)
    m = [0]

    def f():
        m[0] = ctx.voxel_downsample_dev(d_xyz.data_ptr(), None, n, spacing * 1.5, d_out.data_ptr(), None, n)
    wall = median_ms(f, torch, reps=reps)
    out.append(hbm(dict(op="voxel_downsample", workload="%d points in RANDOM order -> %d voxels (first-occurrence order)" % (n, m[0]), ms=wall,
                        note="hash-table grouping: memset + 2 kernels; 12 N in + 12 V out; frac from the wall time of the call"), 12 * n + 12 * m[0], wall))
    d_nrm = torch.empty_like(d_xyz)
    wall = median_ms(lambda: ctx.estimate_normals_dev(d_xyz.data_ptr(), n, 30, d_nrm.data_ptr()), torch, reps=reps)
    out.append(pruned(dict(op="estimate_normals_k30", workload="%d points" % n, ms=wall), 9.0 * n * n, wall, "kNN, one wave per query"))
    d_desc = torch.empty((n, 33), dtype=torch.float32, device=dev)
    wall = median_ms(lambda: ctx.compute_fpfh_dev(d_xyz.data_ptr(), d_nrm.data_ptr(), n, spacing * 5.0, d_desc.data_ptr(), None, None), torch, reps=reps)
    out.append(pruned(dict(op="compute_fpfh", workload="%d points, radius 5 x spacing" % n, ms=wall), 2 * 9.0 * n * n, wall, "radius search + SPFH/FPFH gathers"))
    d_nrm2 = torch.empty_like(d_xyz); d_desc2 = torch.empty_like(d_desc)
    wall = median_ms(lambda: ctx.normals_fpfh_dev(d_xyz.data_ptr(), n, 30, spacing * 5.0, d_nrm2.data_ptr(), d_desc2.data_ptr()), torch, reps=reps)
    out.append(pruned(dict(op="normals_fpfh_one_walk", workload="%d points: estimateNormals(30) + computeFPFH(5 x spacing) sharing ONE radius search (the 30-NN list is the head of the "
                                                                "sorted radius list; a kNN search only for the points with fewer than 30 in radius); same bits as the two calls: %s"
                                                                % (n, bool(torch.equal(d_nrm2, d_nrm) and torch.equal(d_desc2, d_desc))), ms=wall), 3 * 9.0 * n * n, wall, "one radius search + SPFH/FPFH gathers"))
    if want_match:
        d_mx = torch.from_numpy(mdl).to(dev); d_mn = torch.empty_like(d_mx); d_mdesc = torch.empty((n, 33), dtype=torch.float32, device=dev)
        ctx.estimate_normals_dev(d_mx.data_ptr(), n, 30, d_mn.data_ptr())
        ctx.compute_fpfh_dev(d_mx.data_ptr(), d_mn.data_ptr(), n, spacing * 5.0, d_mdesc.data_ptr(), None, None)
        d_corr = torch.empty(n, dtype=torch.int32, device=dev)
        wall = median_ms(lambda: ctx.feature_match_dev(d_desc.data_ptr(), n, d_mdesc.data_ptr(), n, d_corr.data_ptr()), torch, reps=reps)
        out.append(pruned(dict(op="feature_match", workload="%d x %d FPFH descriptors of the cuboid (GPU chain), index build included" % (n, n), ms=wall),
                          98.0 * n * n, wall, "packed index; flat faces give bit-identical rows, which the index holds once"))
        if n == 100000:   # config C3's RANSAC half: 50,000 hypotheses over these correspondences, no early exit
            hyps = 50000
            f = lambda: ctx.ransac_dev(d_xyz.data_ptr(), n, d_mx.data_ptr(), n, None, None, d_corr.data_ptr(), spacing, hyps, 2.0, 42)
            wall = median_ms(f, torch, reps=reps)
            r = f()
            out.append(dict(op="ransac_c3", workload="C3: %d hypotheses x %d correspondences from the descriptor match above (cuboid), confidence 2.0" % (hyps, n),
                            ms=wall, hyps_per_s=hyps / (wall * 1e-3), scored_share=ctx.last_ransac_scored(), rescore_share=ctx.last_ransac_rescore(),
                            best_fitness=float(r.fitness), bound="valu_f32 (k_ransac_score_fast; two batches, the second with the exact bail-out)"))
    return out


def relief_match(ctx, tdv, synth, torch, dev, voxel_px, reps=3):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    bf = importlib.import_module("bench_fm")
    d_fs, ns, d_ft, nt = bf.descriptors(tdv, synth, ctx, voxel_px, dev)
    d_corr = torch.empty(ns, dtype=torch.int32, device=dev)
    wall = median_ms(lambda: ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_corr.data_ptr()), torch, reps=reps)
    d_ix = torch.empty(ns, dtype=torch.int32, device=dev)
    ctx.timing_enable(True); ctx.timing_read(tdv.TIMER_FEATURE_MATCH); ctx.timing_read(tdv.TIMER_FM_INDEX)
    for _ in range(reps):
        ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_ix.data_ptr())
    q_ms = ctx.timing_read(tdv.TIMER_FEATURE_MATCH)[0] / reps; b_ms = ctx.timing_read(tdv.TIMER_FM_INDEX)[0] / reps
    os.environ["TDV_FM_LEAFMAJOR"] = "0"          # round 2's search over the same index (a wave walks its two sources' leaves), for comparison
    try:
        ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_ix.data_ptr()); ctx.timing_read(tdv.TIMER_FEATURE_MATCH)
        for _ in range(reps):
            ctx.feature_match_dev(d_fs.data_ptr(), ns, d_ft.data_ptr(), nt, d_ix.data_ptr())
        walk_ms = ctx.timing_read(tdv.TIMER_FEATURE_MATCH)[0] / reps
    finally:
        del os.environ["TDV_FM_LEAFMAJOR"]
    ctx.timing_enable(False)
    return [pruned(dict(op="feature_match", workload="%d x %d FPFH descriptors of the relief part (instance vs model), index build included" % (ns, nt), ms=wall,
                        query_ms=q_ms, query_ms_round2_walk=walk_ms, index_build_ms=b_ms,
                        note="leaf-major search (k_lm_*); the batched chain builds the index once per model and pays query_ms per instance"),
                   98.0 * ns * nt, wall, "packed index")]


def icp_c2(ctx, tdv, synth, torch, dev, reps=3):
    ns, nt, iters = 50000, 10000, 50
    tgt, nrm = synth.sample_object(nt, 42)
    src, T_gt = synth.make_scene(ns, 42)
    T0 = synth.perturb(T_gt, angle_deg=0.3, trans=0.0005)
    d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_n = torch.from_numpy(nrm).to(dev)
    thr = 0.4 * float(synth.mean_spacing(nt))
    out = []
    for search in ("grid", "pruned", "brute"):
        ctx.set_icp_search(search)
        wall = median_ms(lambda: ctx.icp_dev(d_s.data_ptr(), ns, d_t.data_ptr(), d_n.data_ptr(), nt, T0, thr, iters, True, fixed_iterations=True), torch, reps=reps)
        used = ctx.last_icp_search()
        e = dict(op="icp_c2", workload="C2: 50,000 x 10,000, point-to-plane, %d fixed iterations, threshold 0.4 x spacing, %s search%s"
                 % (iters, used, " (what AUTO picks)" if search == "grid" else ""), ms=wall, iters_per_s=iters / (wall * 1e-3))
        out.append(valu(e, 8.0 * ns * nt * iters, wall) if search == "brute"
                   else pruned(e, 8.0 * ns * nt * iters, wall, "box walk" if used == "pruned" else "hash grid, cells of 2.2 x the threshold"))
    ctx.set_icp_search("auto")
    return out


def voxel_image_order(ctx, tdv, synth, torch, dev, reps=5):
    """voxelDownsample as processInstance meets it (src/pipeline.cpp:92): the cloud of a masked depth frame in row-major pixel order
    (neighbouring points share voxels), one instance and 256 instances in one set of launches."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    bb = importlib.import_module("bench_batch")
    B = 256
    px = bb.DIST / bb.F
    part = synth.ReliefPart(3, L=448 * px, W=448 * px, feature=6.0 * 1.2 * px, density=0.09)
    dense = torch.from_numpy(part.surface_points(px / 2.5)).to(dev)
    d_depth, d_mask = synth.render_depth_torch(dense, synth.instance_pose(0, bb.DIST, 30.0), bb.F, bb.F, bb.CX, bb.CY, bb.W, bb.H, bb.SCALE)
    n_px = int((d_mask > 0).sum())
    d_one = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
    n = ctx.depth_to_cloud_dev(d_depth.data_ptr(), d_mask.data_ptr(), None, bb.W, bb.H, bb.SCALE, bb.F, bb.F, bb.CX, bb.CY, bb.ZMAX, d_one.data_ptr(), None, n_px)
    voxel = 1.2 * px
    d_out = torch.empty_like(d_one)
    m = [0]

    def f():
        m[0] = ctx.voxel_downsample_dev(d_one.data_ptr(), None, n, voxel, d_out.data_ptr(), None, n)
    wall, kms, _ = kernel_ms(ctx, tdv.TIMER_VOXEL, f, torch, reps=reps, warm=2)
    out = [hbm(dict(op="voxel_downsample", workload="one instance cloud in pixel order: %d points -> %d voxels (first-occurrence order)" % (n, m[0]), ms=wall, kernels_ms=kms,
                    note="hash-table grouping: memset + 2 kernels; algorithmic bytes 12 N in + 12 V out; frac from the kernels' time"), 12 * n + 12 * m[0], kms)]
    d_all = d_one[:n].repeat(B, 1).contiguous()                # 256 instance clouds back to back (the same cloud: the table sees them as 256 clouds)
    d_allout = torch.empty_like(d_all)
    off = np.arange(B + 1, dtype=np.int32) * n
    voff = [None]

    def g():
        voff[0] = ctx.voxel_downsample_batch_dev(d_all.data_ptr(), off, voxel, d_allout.data_ptr())
    wall, kms, _ = kernel_ms(ctx, tdv.TIMER_VOXEL, g, torch, reps=3, warm=2)
    v = int(voff[0][-1])
    assert v == B * m[0], (v, B, m[0])
    out.append(hbm(dict(op="voxel_downsample_batch", workload="%d instance clouds of %d points in ONE set of launches -> %d voxels" % (B, n, v), ms=wall, kernels_ms=kms,
                        us_per_instance=wall * 1e3 / B, note="what tdv_register_batch_dev runs for its instances; frac from the kernels' time"),
                   12.0 * B * n + 12.0 * v, kms))
    d_allout2 = torch.empty_like(d_all)

    def gp():
        voff[0] = ctx.voxel_downsample_batch_dev(d_all.data_ptr(), off, voxel, d_allout2.data_ptr(), pinhole=(bb.F, bb.F, bb.CX, bb.CY))
    wall, kms, _ = kernel_ms(ctx, tdv.TIMER_VOXEL, gp, torch, reps=3, warm=2)
    same = bool(torch.equal(d_allout2[:v], d_allout[:v])) and int(voff[0][-1]) == v
    out.append(hbm(dict(op="voxel_downsample_batch_pixel_windows", workload="%d instance clouds of %d points in ONE set of launches -> %d voxels, grouped through pixel windows (%s); "
                                                                             "same voxels as the table path: %s" % (B, n, v, ctx.last_voxel_grouping(), same), ms=wall, kernels_ms=kms,
                        us_per_instance=wall * 1e3 / B, note="what tdv_register_batch_dev runs for its instances (their clouds come from its own unprojection); frac from the kernels' time"),
                   12.0 * B * n + 12.0 * v, kms))
    d_o1 = torch.empty_like(d_one); off1 = np.array([0, n], np.int32)

    def g1():
        voff[0] = ctx.voxel_downsample_batch_dev(d_one.data_ptr(), off1, voxel, d_o1.data_ptr(), pinhole=(bb.F, bb.F, bb.CX, bb.CY))
    wall, kms, _ = kernel_ms(ctx, tdv.TIMER_VOXEL, g1, torch, reps=reps, warm=2)
    same = bool(torch.equal(d_o1[:m[0]], d_out[:m[0]])) and int(voff[0][-1]) == m[0]
    out.append(hbm(dict(op="voxel_downsample_pixel_windows_one_cloud", workload="one instance cloud of %d points with its intrinsics -> %d voxels, grouped through pixel windows (%s); same voxels as "
                                                                                 "voxel_downsample: %s" % (n, m[0], ctx.last_voxel_grouping(), same), ms=wall, kernels_ms=kms,
                        note="tdv_voxel_downsample_batch_pinhole_dev with one cloud: 45 tiles of 4,096 points - a fifth of the chip; frac from the kernels' time"), 12 * n + 12 * m[0], kms))
    return out


def c4_batch(ctx, tdv, synth, torch, dev, instances=256, voxel_px=1.2, hyps=10000, reps=2):
    """Config C4 at its own size: `instances` DISTINCT instances (own pose, own frame, own ~190k-pixel mask) through ONE
    tdv_register_batch_dev call, in the reference's voxel order and in first-occurrence order."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    bb = importlib.import_module("bench_batch")
    out = []
    wl = None
    for order, name in ((tdv.TDV_VOXEL_ORDER_REFERENCE, "reference"), (tdv.TDV_VOXEL_ORDER_FIRST, "first-occurrence")):
        if wl is None:
            wl = bb.build_workload(tdv, synth, ctx, instances, voxel_px, 448, 3, order, dev)
        else:                                                   # same frames; the model in the other voxel order
            wl = dict(wl, model=bb.build_workload(tdv, synth, ctx, 1, voxel_px, 448, 3, order, dev)["model"])
        d_mx, d_mn, d_mf, nm = wl["model"]
        prm = tdv.batch_params(width=bb.W, height=bb.H, scale_to_meters=bb.SCALE, fx=bb.F, fy=bb.F, cx=bb.CX, cy=bb.CY, zmax=bb.ZMAX, voxel_size=wl["voxel"],
                               ransac_max_iterations=hyps, icp_max_iterations=50, voxel_order=order, n_frames=instances)
        res = [None]

        def f():
            res[0] = ctx.register_batch_dev(wl["depth"].data_ptr(), None, wl["masks"].data_ptr(), instances, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
        wall = median_ms(f, torch, reps=reps, warm=1)
        ang = max(synth.pose_error(r["T"], T)[0] for r, T in zip(res[0], wl["T_gt"]))
        out.append(dict(op="register_batch", workload="C4: %d distinct instances (~%d px masks, %d voxels) vs a %d-point model, %d hypotheses + ICP each, %s voxel order"
                        % (instances, int(np.mean(wl["mask_px"])), int(np.mean([r["n_voxels"] for r in res[0]])), nm, hyps, name),
                        ms=wall, instances_per_s=instances / (wall * 1e-3), ms_per_instance=wall / instances, max_angle_to_ground_truth_rad=ang,
                        icp_iterations_per_instance=float(np.mean([r["icp_iterations"] for r in res[0]])),
                        workspace_high_water_MiB=ctx.workspace_high_water() / 2 ** 20, bound="chain of the operators above"))
    return out


def c5_batch(ctx, tdv, synth, torch, dev, instances=1024):
    """Config C5, one rank's share: 1,024 instances cut by one uint16 label image from one frame (tools/c5_tray.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    c5 = importlib.import_module("c5_tray")
    out, _, _ = c5.measure(tdv, synth, ctx, torch, dev, instances, reps=2)
    return [out]


def measure_all(ctx, tdv, synth, torch, dev, quick=False):
    out = []
    out += depth_ops(ctx, tdv, torch, dev)
    out += voxel_image_order(ctx, tdv, synth, torch, dev)
    for n in ([100000] if quick else [100000, 200000]):
        out += cloud_ops(ctx, tdv, synth, torch, dev, n)
    out += relief_match(ctx, tdv, synth, torch, dev, 1.45)   # C3's size (~110k x 110k)
    out += relief_match(ctx, tdv, synth, torch, dev, 1.2)    # C4's size (~143k x 151k)
    out += icp_c2(ctx, tdv, synth, torch, dev)
    out += c4_batch(ctx, tdv, synth, torch, dev, instances=16 if quick else 256)
    out += c5_batch(ctx, tdv, synth, torch, dev, instances=128 if quick else 1024)
    return out
