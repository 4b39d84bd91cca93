"""Study: bench.py's RANSAC workload at another share of true correspondences (bench.py: 0.5): hypotheses/s and the shares of the
(hypothesis, point) tests scored / scored again.  Usage: ransac_true_share_probe.py [points] [true share] [hypotheses].
(The workload profiles/r4/history/ransac_chunk_test.md was measured on.)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
true_share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
voxel = float(np.float32(synth.mean_spacing(n)))
tgt, nrm = synth.sample_object(n, 42)
src, T_gt = synth.make_scene(n, 42)
nn = ctx.icp_correspondences(src, tgt, T_gt, 1.0)["corr"]
rng = np.random.Generator(np.random.PCG64(1234))
corr = np.where(rng.random(n) < true_share, nn, rng.integers(0, n, n)).astype(np.int32)
d_src = torch.from_numpy(src).to(dev); d_tgt = torch.from_numpy(tgt).to(dev); d_corr = torch.from_numpy(corr).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = ctx.ransac_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, None, None, d_corr.data_ptr(), voxel, iters, 2.0, 42)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("n %d true %.2f: %.2f M hyps/s  scored %.3f rescored %.3f  inliers %d best %d" % (n, true_share, iters / dt / 1e6, ctx.last_ransac_scored(),
          ctx.last_ransac_rescore(), r.inliers, r.best_iteration), flush=True)
