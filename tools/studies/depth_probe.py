import sys, os, importlib, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import opbench
tdv = importlib.import_module("3dvision_amd"); ctx = tdv.Context(0)
e = opbench.depth_ops(ctx, tdv, torch, torch.device("cuda", 0))[1]
print(json.dumps({k: e[k] for k in ("op", "ms", "kernels_ms", "achieved_GBps")}))
