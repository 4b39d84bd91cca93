import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
ctx = tdv.Context(0); dev = torch.device("cuda", 0)
n = 100000
fs = torch.from_numpy(synth.random_features(n, 1)).to(dev); ft = torch.from_numpy(synth.random_features(n, 2)).to(dev)
corr = torch.empty(n, dtype=torch.int32, device=dev)
f = lambda: ctx.feature_match_dev(fs.data_ptr(), n, ft.data_ptr(), n, corr.data_ptr())
f(); ts = []
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("random rows 100k x 100k, limit", os.environ.get("TDV_FM_LIMIT"), "ms", round(sorted(ts)[1], 2))
