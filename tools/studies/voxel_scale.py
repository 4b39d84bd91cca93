"""Voxel downsampling over cloud sizes: where the operator stops being launch/latency-bound and what it reaches then.
A full 1280x720 frame is 0.9 M points, a 4-frame merge 3.7 M; 16 M is only there to see the asymptote."""
import sys, os, importlib, json, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import opbench
tdv = importlib.import_module("3dvision_amd"); ctx = tdv.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (200_000, 921_600, 3_686_400, 16_000_000):
    # a tilted plane patch with noise, ~2.3 points per voxel like the depth clouds of C4
    side = (n / 2.3) ** 0.5 * 0.001
    uv = torch.rand((n, 2), device=dev, generator=g) * side
    xyz = torch.stack([uv[:, 0], uv[:, 1], 0.8 + 0.3 * uv[:, 0] + 0.0002 * torch.rand(n, device=dev, generator=g)], 1).contiguous()
    out = torch.empty_like(xyz)
    f = lambda: ctx.voxel_downsample_dev(xyz.data_ptr(), None, n, 0.001, out.data_ptr(), None, n)
    v = f()
    wall, kms, launches = opbench.kernel_ms(ctx, tdv.TIMER_VOXEL, f, torch, reps=7, warm=2)
    alg = 12 * n + 12 * v
    print(json.dumps(dict(n=n, voxels=v, wall_ms=round(wall, 4), events_ms=round(kms, 4), algorithmic_MB=round(alg / 1e6, 1),
                          GBps_wall=round(alg / wall / 1e6, 1))))
