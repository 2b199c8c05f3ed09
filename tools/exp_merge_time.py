#!/usr/bin/env python3
"""8192x8192 merging transform (final canonical labels, device resident): wall clock per transform."""
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
img = eng.random_field(n, n, 1); seeds = eng.find_local_minima(img)
out = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.merge(img, seeds, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 5
for _ in range(K): eng.merge(img, seeds, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"merge {n}x{n}: {dt*1e3:.3f} ms  {n*n/dt/1e9:.2f} Gpx/s")
