"""Study: tools/opbench.py's voxel rows (one cloud through the table, 256 clouds both ways, one cloud through pixel windows)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth"); ob = importlib.import_module("opbench")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
for r in ob.voxel_image_order(ctx, tdv, synth, torch, dev):
    print(json.dumps({k: r[k] for k in ("op", "ms", "kernels_ms", "achieved_GBps", "frac", "workload")}), flush=True)
