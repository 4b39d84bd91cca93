"""ICP with the reference's accumulation order at several sizes: whole-iteration rate, for rocprofv3 --kernel-trace --stats
(k_icp_rows / k_icp_fold_ref durations).  usage: python tools/studies/icp_ref_order_probe.py [n ...]"""
import sys, importlib, json, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import torch, numpy as np
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth"); ctx = tdv.Context(0)
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [50000, 200000]:
    tgt, nrm = synth.sample_object(n, 42); src, T_gt = synth.make_scene(n, 42)
    T0 = synth.perturb(T_gt, angle_deg=0.3, trans=0.0005)
    d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_n = torch.from_numpy(nrm).to(dev)
    thr = 0.4 * float(synth.mean_spacing(n))
    for mode in ("tree", "reference"):
        for p2pl in (True, False):
            ctx.set_icp_accumulation(mode)
            f = lambda: ctx.icp_dev(d_s.data_ptr(), n, d_t.data_ptr(), d_n.data_ptr() if p2pl else None, n, T0, thr, 20, p2pl, fixed_iterations=True)
            f(); torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(json.dumps(dict(n=n, accumulate=mode, mode="p2plane" if p2pl else "p2point", ms_per_iteration=dt / 20 * 1e3, iters_per_s=20 / dt, fitness=float(r.fitness))))
ctx.set_icp_accumulation("tree")
