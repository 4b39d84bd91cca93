"""One-off check (GPU box): tdv_register_batch_dev in the reference's voxel order on 64 distinct C4 instances, with its per-point
stages on the coherent ordering (default) and on the reference-ordered cloud itself (TDV_BATCH_COHERENT=0, read once per
process - hence two runs): every returned field must be identical bit for bit.

    TDV_BATCH_COHERENT=0 python tools/studies/batch_coherent_ab.py /tmp/a.npz && python tools/studies/batch_coherent_ab.py /tmp/b.npz /tmp/a.npz
"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
bb = importlib.import_module("bench_batch")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
B = 64
order = tdv.TDV_VOXEL_ORDER_REFERENCE
wl = bb.build_workload(tdv, synth, ctx, B, 1.2, 448, 3, order, dev)
d_mx, d_mn, d_mf, nm = wl["model"]
prm = tdv.batch_params(width=bb.W, height=bb.H, scale_to_meters=bb.SCALE, fx=bb.F, fy=bb.F, cx=bb.CX, cy=bb.CY, zmax=bb.ZMAX, voxel_size=wl["voxel"],
                       ransac_max_iterations=10000, ransac_confidence=0.999, icp_max_iterations=50, icp_distance_factor=0.4, voxel_order=order, n_frames=B)
res = ctx.register_batch_dev(wl["depth"].data_ptr(), None, wl["masks"].data_ptr(), B, prm, d_mx.data_ptr(), d_mn.data_ptr(), d_mf.data_ptr(), nm)
out = dict(T=np.stack([r["T"] for r in res]), inl=np.array([r["coarse_inliers"] for r in res]), cf=np.array([r["coarse_fitness"] for r in res], np.float32),
           fit=np.array([r["fitness"] for r in res], np.float32), rmse=np.array([r["rmse"] for r in res], np.float32), it=np.array([r["icp_iterations"] for r in res]),
           nv=np.array([r["n_voxels"] for r in res]))
np.savez(sys.argv[1], **out)
if len(sys.argv) > 2:
    ref = np.load(sys.argv[2])
    same = all(out[k].tobytes() == ref[k].tobytes() for k in out)
    print("coherent stages vs reference-ordered stages on %d instances: %s" % (B, "identical bit for bit" if same else "DIFFERENT"))
    sys.exit(0 if same else 1)
