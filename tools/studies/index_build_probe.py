"""Study: the descriptor index build (csrc/fmatch.hip, two radix sorts of csrc/sort.hip inside) and the query on tools/opbench.py's relief parts."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth"); ob = importlib.import_module("opbench")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
for scale in (1.45, 1.2):
    for r in ob.relief_match(ctx, tdv, synth, torch, dev, scale):
        print(json.dumps({k: r[k] for k in ("workload", "ms", "query_ms", "index_build_ms")}), flush=True)
