#!/usr/bin/env python3
"""Wall-clock per transform with and without the per-launch HIP-event spans."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
if os.environ.get("WS_OWN_STREAM"):      # torch's default stream is the legacy null stream: no graph capture there
    torch.cuda.set_stream(torch.cuda.Stream())
eng = dev.DeviceEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
img = eng.random_field(n, n, 1)
seeds = eng.find_local_minima(img)
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for prof in (False, True, False, True):
    eng.ctx.set_profiling(prof)
    for _ in range(3): eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K): eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"profiling={prof}: {dt*1e3:.3f} ms/transform  {n*n/dt/1e9:.2f} Gpx/s  stats ms_total={eng.stats()['ms_total']:.3f}")
