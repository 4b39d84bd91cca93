"""One pass of the new code paths at 2,000,000 points: voxel (LDS bucket staging), descriptor index build (radix sorts) + match on a
subset, hash-grid ICP (no-wrap table), RANSAC with the bail-out."""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
tdv = importlib.import_module('3dvision_amd'); synth = importlib.import_module('3dvision_amd.synth')
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
n = 2000000
pts, nrm = synth.sample_object(n, 3)
src, T_gt = synth.make_scene(n, 3)
spacing = float(synth.mean_spacing(n))
v1, _ = ctx.voxel_downsample(src, None, spacing * 1.5, tdv.TDV_VOXEL_ORDER_FIRST)
v2, _ = ctx.voxel_downsample(src, None, spacing * 1.5, tdv.TDV_VOXEL_ORDER_REFERENCE)
assert len(v1) == len(v2) and np.array_equal(np.sort(v1.view([('x', 'f4'), ('y', 'f4'), ('z', 'f4')]).ravel()), np.sort(v2.view([('x', 'f4'), ('y', 'f4'), ('z', 'f4')]).ravel()))
print("voxel 2M -> %d voxels, both orders hold the same means" % len(v1))
T0 = synth.perturb(T_gt, 3, angle_deg=0.3, trans=0.0005)
ctx.set_icp_search("grid"); g = ctx.icp(src, pts, nrm, T0, spacing * 0.4, 30, True); print("icp grid:", ctx.last_icp_search(), g.iterations, float(g.fitness))
ctx.set_icp_search("pruned"); w = ctx.icp(src, pts, nrm, T0, spacing * 0.4, 30, True); print("icp walk:", ctx.last_icp_search(), w.iterations, float(w.fitness))
assert g.transformation.tobytes() == w.transformation.tobytes() and g.iterations == w.iterations
ctx.set_icp_search("auto")
fs = synth.random_features(600000, 5); ft = synth.random_features(500000, 6)
t = time.perf_counter(); c = ctx.feature_match(fs, ft); print("feature match 600k x 500k: %.1f ms (host API)" % ((time.perf_counter() - t) * 1e3))
sel = np.random.default_rng(1).choice(len(fs), 200, replace=False)
for i in sel:
    d = ((ft - fs[i]) ** 2).sum(1)
    assert d[c[i]] <= d.min() * (1 + 1e-5) + 1e-9, i
nn = ctx.icp_correspondences(src, pts, T_gt, 1.0)["corr"]
corr = np.where(np.random.default_rng(2).random(n) < 0.5, nn, np.random.default_rng(3).integers(0, n, n)).astype(np.int32)
a = ctx.ransac(src, pts, corr=corr, voxel=spacing, max_iterations=200000, confidence=2.0); sc = ctx.last_ransac_scored()
ctx.set_ransac_score("exact"); b = ctx.ransac(src, pts, corr=corr, voxel=spacing, max_iterations=200000, confidence=2.0); ctx.set_ransac_score("fast")
assert (a.best_iteration, a.inliers, a.fitness) == (b.best_iteration, b.inliers, b.fitness) and a.transformation.tobytes() == b.transformation.tobytes()
print("ransac 200k hyps x 2M points: best %d inliers at %d, %.3f of the tests scored; equals the exact kernel" % (a.inliers, a.best_iteration, sc))
