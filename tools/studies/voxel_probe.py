import sys, os, importlib, json, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import opbench
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth"); ctx = tdv.Context(0)
dev = torch.device("cuda", 0)
for n in (100000, 200000):
    cam, _, _ = opbench.cuboid_scene(synth, n)
    d_xyz = torch.from_numpy(cam).to(dev); d_out = torch.empty_like(d_xyz)
    sp = float(synth.mean_spacing(n)) * 1.5
    f = lambda: ctx.voxel_downsample_dev(d_xyz.data_ptr(), None, n, sp, d_out.data_ptr(), None, n)
    wall, kms, launches = opbench.kernel_ms(ctx, tdv.TIMER_VOXEL, f, torch, reps=20, warm=3)
    print(json.dumps(dict(n=n, wall_ms=wall, events_ms=kms)))
