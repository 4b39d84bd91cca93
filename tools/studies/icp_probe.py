import sys, importlib, json, torch, time, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import opbench
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth"); ctx = tdv.Context(0)
dev = torch.device("cuda", 0)
n = 200000
tgt, nrm = synth.sample_object(n, 42); src, T_gt = synth.make_scene(n, 42)
T0 = synth.perturb(T_gt, angle_deg=0.3, trans=0.0005)
d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_n = torch.from_numpy(nrm).to(dev)
thr = 0.4 * float(synth.mean_spacing(n))
f = lambda: ctx.icp_dev(d_s.data_ptr(), n, d_t.data_ptr(), d_n.data_ptr(), n, T0, thr, 50, True, fixed_iterations=True)
wall, kms, launches = opbench.kernel_ms(ctx, tdv.TIMER_ICP_NN, f, torch, reps=5, warm=2)
print(json.dumps(dict(op="icp 200k x 200k, 50 fixed iterations", ms=wall, iters_per_s=50 / (wall * 1e-3), nn_kernel_ms=kms / max(launches, 1))))
for e in opbench.icp_c2(ctx, tdv, synth, torch, dev): print(json.dumps({k: e[k] for k in ("workload", "ms", "iters_per_s")}))
