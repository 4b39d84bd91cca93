#!/usr/bin/env python3
"""Per-operator timings of the hot path on one MI355X, inputs resident in HBM (the *_dev ABI): one JSON line per
measurement, from tools/opbench.py (the same functions bench.py runs after its timed region).

    python tools/bench_ops.py [--quick]
"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import torch
    import opbench
    tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
    ctx = tdv.Context(0)
    for e in opbench.measure_all(ctx, tdv, synth, torch, torch.device("cuda", 0), quick=args.quick):
        print(json.dumps(e), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
