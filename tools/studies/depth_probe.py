import sys, importlib, json, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import opbench
tdv = importlib.import_module("3dvision_amd"); ctx = tdv.Context(0)
e = opbench.depth_ops(ctx, tdv, torch, torch.device("cuda", 0))[1]
print(json.dumps({k: e[k] for k in ("op", "ms", "kernels_ms", "achieved_GBps")}))
