"""RANSAC scoring on one C4 instance (relief part, real FPFH correspondences): time in both scoring modes and the share of
(wave, chunk) pairs the fast pass scores twice."""
import sys, os, importlib, json, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_batch as bb
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
dev = torch.device("cuda", 0); ctx = tdv.Context(0)
for voxel_px in (1.2, 2.0):
    wl = bb.build_workload(tdv, synth, ctx, 1, voxel_px, 448, 3, tdv.TDV_VOXEL_ORDER_FIRST, dev)
    d_mx, d_mn, d_mf, nm = wl["model"]
    n_px = wl["mask_px"][0]
    d_xyz = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
    n = ctx.depth_to_cloud_dev(wl["depth"][0].data_ptr(), wl["masks"][0].data_ptr(), None, bb.W, bb.H, bb.SCALE, bb.F, bb.F, bb.CX, bb.CY, bb.ZMAX, d_xyz.data_ptr(), None, n_px)
    d_v = torch.empty_like(d_xyz)
    v = ctx.voxel_downsample_dev(d_xyz.data_ptr(), None, n, wl["voxel"], d_v.data_ptr(), None, n)
    d_n = torch.empty((v, 3), dtype=torch.float32, device=dev); d_f = torch.empty((v, 33), dtype=torch.float32, device=dev)
    ctx.estimate_normals_dev(d_v.data_ptr(), v, 30, d_n.data_ptr())
    ctx.compute_fpfh_dev(d_v.data_ptr(), d_n.data_ptr(), v, wl["voxel"] * 5.0, d_f.data_ptr())
    d_corr = torch.empty(v, dtype=torch.int32, device=dev)
    ctx.feature_match_dev(d_f.data_ptr(), v, d_mf.data_ptr(), nm, d_corr.data_ptr())
    out = dict(voxel_px=voxel_px, ns=v, nt=nm)
    for mode in ("exact", "fast"):
        ctx.set_ransac_score(mode)
        for hyps in (10000, 65536):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t = time.perf_counter()
                r = ctx.ransac_dev(d_v.data_ptr(), v, d_mx.data_ptr(), nm, None, None, d_corr.data_ptr(), wl["voxel"], hyps, 2.0, 42)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            out["%s_%d" % (mode, hyps)] = dict(ms=float(np.median(ts)), inliers=int(r.inliers), best=int(r.best_iteration), rescore_share=ctx.last_ransac_rescore())
    print(json.dumps(out))
