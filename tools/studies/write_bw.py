"""How fast can HBM be written?  (ceiling for the emit pass of the batched depth->cloud: 617 MB of points per call)
torch.fill_ / zero_ / copy_ on buffers well beyond the 256 MB Infinity Cache, HIP events, median of 9."""
import torch, json
dev = torch.device("cuda", 0)
def timed(fn, reps=9):
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
for mb in (617, 2048):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
    for name, fn, bytes_ in (("fill", lambda: x.fill_(1.5), 4 * n), ("zero", lambda: x.zero_(), 4 * n), ("copy", lambda: y.copy_(x), 8 * n)):
        fn(); ms = timed(fn)
        print(json.dumps({"op": name, "MB": mb, "ms": ms, "TBps": bytes_ / ms / 1e9}))
