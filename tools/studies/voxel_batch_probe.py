#!/usr/bin/env python3
"""Kernel times of the batched voxel stage (tools/opbench.py: voxel_image_order) - run under tools/prof_kernels.sh."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import opbench
tdv = importlib.import_module("3dvision_amd"); synth = importlib.import_module("3dvision_amd.synth")
ctx = tdv.Context(0)
for e in opbench.voxel_image_order(ctx, tdv, synth, torch, torch.device("cuda", 0)):
    print(json.dumps(e))
