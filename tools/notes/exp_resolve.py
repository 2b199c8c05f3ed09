#!/usr/bin/env python3
"""Timing experiment: resolve kernels only, with the jump-round cap WS_DEBUG_MAXIT (results wrong when it bites)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
eng = dev.DeviceEngine(0)
n = 8192
img = eng.random_field(n, n, 1); seeds = eng.find_local_minima(img)
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.segment(img, seeds, out=labels)
eng.ctx.set_profiling(True)
acc = 0.0
for _ in range(5):
    eng.segment(img, seeds, out=labels); acc += eng.stats()["ms_resolve"]
print("ms_resolve", round(acc / 5, 4))
