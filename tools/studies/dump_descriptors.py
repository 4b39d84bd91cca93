#!/usr/bin/env python3
"""Offline-study helper (GPU box): FPFH descriptors of the C4 workload's model and of one instance, as .npy files
under gpurun_out/ for the descriptor-match studies in this directory."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_batch as bb  # noqa: E402

tdv = importlib.import_module("3dvision_amd")
synth = importlib.import_module("3dvision_amd.synth")
voxel_px = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2
dev = torch.device("cuda", 0)
ctx = tdv.Context(0)
wl = bb.build_workload(tdv, synth, ctx, 1, voxel_px, 448, 3, tdv.TDV_VOXEL_ORDER_FIRST, dev)
d_mx, d_mn, d_mf, nm = wl["model"]
n_px = wl["mask_px"][0]
d_xyz = torch.empty((n_px, 3), dtype=torch.float32, device=dev)
n = ctx.depth_to_cloud_dev(wl["depth"][0].data_ptr(), wl["masks"][0].data_ptr(), None, bb.W, bb.H, bb.SCALE, bb.F, bb.F, bb.CX, bb.CY, bb.ZMAX,
                           d_xyz.data_ptr(), None, n_px)
d_v = torch.empty_like(d_xyz)
v = ctx.voxel_downsample_dev(d_xyz.data_ptr(), None, n, wl["voxel"], d_v.data_ptr(), None, n)
d_n = torch.empty((v, 3), dtype=torch.float32, device=dev); d_f = torch.empty((v, 33), dtype=torch.float32, device=dev)
ctx.estimate_normals_dev(d_v.data_ptr(), v, 30, d_n.data_ptr())
ctx.compute_fpfh_dev(d_v.data_ptr(), d_n.data_ptr(), v, wl["voxel"] * 5.0, d_f.data_ptr())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = ("%.1f" % voxel_px).replace(".", "p")
np.save(os.path.join(ROOT, "gpurun_out", "desc_model_%s.npy" % tag), d_mf[:nm].cpu().numpy())
np.save(os.path.join(ROOT, "gpurun_out", "desc_inst_%s.npy" % tag), d_f.cpu().numpy())
np.save(os.path.join(ROOT, "gpurun_out", "xyz_model_%s.npy" % tag), d_mx[:nm].cpu().numpy())
np.save(os.path.join(ROOT, "gpurun_out", "xyz_inst_%s.npy" % tag), d_v[:v].cpu().numpy())
np.save(os.path.join(ROOT, "gpurun_out", "Tgt_inst_%s.npy" % tag), wl["T_gt"][0])
print("model", nm, "instance", v)
