"""Does the order of the SOURCE points matter to the hash-grid ICP search?  The same 200k x 200k pair with the sources in sampling
(random) order and in Morton order: kernel time of k_icp_nn_grid and whole iterations/s."""
import importlib, os, sys, json, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
tdv = importlib.import_module("3dvision_amd")
synth = importlib.import_module("3dvision_amd.synth")
n = 200000
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
tgt, nrm = synth.sample_object(n, 42)
src, T_gt = synth.make_scene(n, 42)
T0 = synth.perturb(T_gt, 42, angle_deg=0.3, trans=0.0005)
voxel = float(np.float32(synth.mean_spacing(n)))

def morton(p):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)

d_tgt = torch.from_numpy(tgt).to(dev); d_nrm = torch.from_numpy(nrm).to(dev)
for name, s in (("sampling order", src), ("morton order", src[np.argsort(morton(src), kind="stable")])):
    d_src = torch.from_numpy(np.ascontiguousarray(s)).to(dev)
    ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, voxel * 0.4, 20, True, fixed_iterations=True)
    ctx.timing_enable(True); ctx.timing_read(tdv.TIMER_ICP_NN)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.icp_dev(d_src.data_ptr(), n, d_tgt.data_ptr(), d_nrm.data_ptr(), n, T0, voxel * 0.4, 200, True, fixed_iterations=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, launches = ctx.timing_read(tdv.TIMER_ICP_NN); ctx.timing_enable(False)
    print(json.dumps({"sources": name, "search": ctx.last_icp_search(), "nn_kernel_us": ms / launches * 1e3, "iters_per_s": 200 / dt, "fitness": float(r.fitness)}))
