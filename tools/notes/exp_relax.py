#!/usr/bin/env python3
"""Timing experiment: one 8192^2 segmenting transform, engine stats (passes, tiles, in-tile sweeps)."""
import os, sys, time, json
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
seeds = eng.find_local_minima(img)
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.segment(img, seeds, out=labels)
eng.ctx.set_profiling(True)
acc = None
K = 5
for _ in range(K):
    eng.segment(img, seeds, out=labels)
    st = eng.stats()
    acc = st if acc is None else {k: acc[k] + st[k] for k in st}
print(json.dumps({k: round(v / K, 3) for k, v in acc.items()}))
