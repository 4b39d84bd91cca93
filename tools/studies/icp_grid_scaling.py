"""k_icp_nn_grid against the number of source points (same 200k-point target): the fixed and the per-point part of its time."""
import importlib, os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
tdv = importlib.import_module("3dvision_amd")
synth = importlib.import_module("3dvision_amd.synth")
nt = 200000
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
tgt, nrm = synth.sample_object(nt, 42)
voxel = float(np.float32(synth.mean_spacing(nt)))
d_tgt = torch.from_numpy(tgt).to(dev); d_nrm = torch.from_numpy(nrm).to(dev)
for ns in (12500, 25000, 50000, 100000, 200000, 400000, 800000):
    src, T_gt = synth.make_scene(ns, 42)
    T0 = synth.perturb(T_gt, 42, angle_deg=0.3, trans=0.0005)
    d_src = torch.from_numpy(src).to(dev)
    ctx.set_icp_search("grid")
    ctx.icp_dev(d_src.data_ptr(), ns, d_tgt.data_ptr(), d_nrm.data_ptr(), nt, T0, voxel * 0.4, 10, True, fixed_iterations=True)
    ctx.timing_enable(True); ctx.timing_read(tdv.TIMER_ICP_NN)
    ctx.icp_dev(d_src.data_ptr(), ns, d_tgt.data_ptr(), d_nrm.data_ptr(), nt, T0, voxel * 0.4, 100, True, fixed_iterations=True)
    ms, launches = ctx.timing_read(tdv.TIMER_ICP_NN); ctx.timing_enable(False)
    print(json.dumps({"ns": ns, "search": ctx.last_icp_search(), "nn_kernel_us": ms / launches * 1e3, "ns_per_point": ms / launches * 1e6 / ns}))
