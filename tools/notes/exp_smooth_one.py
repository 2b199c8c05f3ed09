#!/usr/bin/env python3
"""One smooth 8192x8192 field (correlation length argv[1], default 64): a few transforms, for kernel traces."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
eng = dev.DeviceEngine(0)
n = 8192; corr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(3)
for c in (4, 16, 64, 256):          # same generator sequence as exp_smooth.py
    low = torch.rand((1, 1, n // c + 2, n // c + 2), device="cuda", generator=g)
    if c == corr: break
up = torch.nn.functional.interpolate(low, size=(n, n), mode="bicubic", align_corners=False)[0, 0]
up = (up - up.min()) / (up.max() - up.min())
img = (up * 253.0).to(torch.uint8).contiguous()
seeds = eng.find_local_minima(img)
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.segment(img, seeds, out=labels)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.segment(img, seeds, out=labels)
torch.cuda.synchronize(); print(f"corr {corr}: {(time.perf_counter() - t0) * 1e3:.2f} ms, seeds {seeds.shape[0]}")
