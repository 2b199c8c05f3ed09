#!/usr/bin/env python3
"""32768^2 single-domain transforms, call by call: time and statistics, on torch's null stream or (WS_OWN_STREAM=1) a real one."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
if os.environ.get("WS_OWN_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
img = eng.random_field(n, n, 5)
seeds = eng.find_local_minima(img)
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eng.stats()
    print(f"call {i}: {dt*1e3:9.3f} ms  passes {st['relax_passes']} resolve {st['resolve_passes']} graph {st['graph_launches']} ms_total {st['ms_total']:.3f}", flush=True)
print("coloured", int((labels != 0).sum()), "of", (n - 2) * (n - 2), "+border seeds")
