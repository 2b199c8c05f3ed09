#!/usr/bin/env python3
"""The README's call pair on one 8192 x 8192 random field, device resident, for `rocprofv3 --kernel-trace --stats`:
exp_pair.py [N=8192] [reps=10] [one|two]   (one: ws_segment_minima_device; two: ws_find_local_minima_device + ws_segment_device)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
form = sys.argv[3] if len(sys.argv) > 3 else "one"
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
img = eng.random_field(n, n, 1)
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
ts = []
for i in range(reps + 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if form == "one":
        eng.segment_minima(img, out=labels)
    else:
        eng.segment(img, eng.find_local_minima(img), out=labels)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts = sorted(ts[3:])
print(f"{form} call(s), {n}x{n}: median {ts[len(ts)//2]*1e3:.4f} ms, min {ts[0]*1e3:.4f} ms")
