"""Times the RANSAC scoring kernel alone (HIP events on the ctx's stream) at 200k points: used to A/B the scoring kernels
(TDV_RANSAC_SCORE=exact|mfma, default = the FMA pass) and their compile-time variants (tools/tune_variant.sh)."""
import importlib, os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
tdv = importlib.import_module("3dvision_amd")
synth = importlib.import_module("3dvision_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
hyps = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * 65536
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
tgt, nrm = synth.sample_object(n, 42)
src, T_gt = synth.make_scene(n, 42)
nn = ctx.icp_correspondences(src, tgt, T_gt, 1.0)["corr"]
rng = np.random.Generator(np.random.PCG64(1234))
corr = np.where(rng.random(n) < 0.5, nn, rng.integers(0, n, n)).astype(np.int32)
d_src = torch.from_numpy(src).to(dev); d_tgt = torch.from_numpy(tgt).to(dev); d_corr = torch.from_numpy(corr).to(dev)
voxel = float(np.float32(synth.mean_spacing(n)))
class _None: inliers = -1; best_iteration = -1
def run(h):
    try:
        return ctx.ransac_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), n, None, None, d_corr.data_ptr(), voxel, h, 2.0, 42)
    except tdv.TdvError as e:       # a probe build (RM_PROBE) counts wrongly on purpose: the library's own cross-check refuses the result
        print("#", str(e)[:120]); return _None()
run(65536)
ctx.timing_enable(True); ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
r = run(hyps)
ms, launches = ctx.timing_read(tdv.TIMER_RANSAC_SCORE)
print(json.dumps({"mode": os.environ.get("TDV_RANSAC_SCORE", "fast"), "n": n, "hyps": hyps, "kernel_ms_per_launch": ms / launches, "launches": launches,
                  "kernel_Mhyps_per_s": hyps / ms / 1e3, "T_pairs_per_s": hyps * n / ms / 1e9, "inliers": int(r.inliers), "best_iteration": int(r.best_iteration),
                  "rescore_share": ctx.last_ransac_rescore()}))
