#!/usr/bin/env python3
"""Device-resident segmenting transform at widths that are / are not multiples of 4, with and without edge correction."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
for (h, w, edge) in ((8192, 8192, False), (8192, 8192, True), (8192, 8190, False), (8192, 8190, True), (8192, 8191, False), (4096, 4096, True), (2048, 2048, True), (2048, 2048, False)):
    img = eng.random_field(h, w, 1)
    seeds = eng.find_local_minima(img)
    e = 2 if edge else 0
    labels = torch.empty((h + e, w + e), dtype=torch.int32, device=eng.device)
    for _ in range(5): eng.segment(img, seeds, edge=edge, out=labels)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 10
    for _ in range(K): eng.segment(img, seeds, edge=edge, out=labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    eng.ctx.set_profiling(True); eng.segment(img, seeds, edge=edge, out=labels); st = eng.stats(); eng.ctx.set_profiling(False)
    print(f"{h}x{w} edge={edge}: {dt*1e3:.3f} ms  ({h*w/dt/1e9:.1f} Gpx/s)  relax {st['ms_relax']:.3f} resolve {st['ms_resolve']:.3f} other {st['ms_other']:.3f} passes {st['relax_passes']}", flush=True)
