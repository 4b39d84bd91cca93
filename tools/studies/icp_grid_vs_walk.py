"""ICP iteration time at 200k x 200k over acceptance thresholds (in point spacings): hash grid against box walk."""
import sys, os, importlib, json, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
n = 200000
tgt, nrm = synth.sample_object(n, 42); src, T_gt = synth.make_scene(n, 42)
T0 = synth.perturb(T_gt, angle_deg=0.3, trans=0.0005)
d_s = torch.from_numpy(src).to(dev); d_t = torch.from_numpy(tgt).to(dev); d_n = torch.from_numpy(nrm).to(dev)
sp = float(synth.mean_spacing(n))
for mult in (0.4, 1.0, 2.0, 4.0, 8.0):
    out = dict(threshold_spacings=mult)
    for mode in ("grid", "pruned"):
        ctx.set_icp_search(mode)
        f = lambda: ctx.icp_dev(d_s.data_ptr(), n, d_t.data_ptr(), d_n.data_ptr(), n, T0, mult * sp, 40, True, fixed_iterations=True)
        f(); ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        out[mode] = dict(ms_per_iter=round(sorted(ts)[1] / 40, 4), used=ctx.last_icp_search(), n_corr=int(r.n_corr))
    print(json.dumps(out))
ctx.set_icp_search("auto")
