#!/usr/bin/env python3
"""find_local_minima on a device-resident 8192x8192 field: wall clock per call."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
eng = dev.DeviceEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
img = eng.random_field(n, n, 1)
for _ in range(2): s = eng.find_local_minima(img)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 10
for _ in range(K): s = eng.find_local_minima(img)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"find_local_minima {n}x{n}: {dt*1e3:.3f} ms, {s.shape[0]} seeds")
