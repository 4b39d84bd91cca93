"""Study: one frame's depth -> cloud (csrc/depth.hip: k_depth_cloud_chain; TDV_DEPTH_THREE_PASS=1 with the study library = rounds 1-3).
Prints tools/opbench.py's two depth rows."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
tdv = importlib.import_module("3dvision_amd"); ob = importlib.import_module("opbench")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
for r in ob.depth_ops(ctx, tdv, torch, dev, reps=20):
    print(json.dumps({k: r[k] for k in ("op", "ms", "kernels_ms", "achieved_GBps", "frac", "workload")}), flush=True)
