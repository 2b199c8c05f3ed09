#!/usr/bin/env python3
"""Segmenting and merging transforms on SMOOTH fields (the shape of real maps, not of the bench): low-pass
noise at several correlation lengths, 8192x8192.  Prints time, passes and tile runs."""
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
g = torch.Generator(device="cuda").manual_seed(3)
for corr in (0, 4, 16, 64, 256):
    if corr == 0:
        img = eng.random_field(n, n, 1)
    else:
        low = torch.rand((1, 1, n // corr + 2, n // corr + 2), device="cuda", generator=g)
        up = torch.nn.functional.interpolate(low, size=(n, n), mode="bicubic", align_corners=False)[0, 0]
        up = (up - up.min()) / (up.max() - up.min())
        img = (up * 253.0).to(torch.uint8).contiguous()
    seeds = eng.find_local_minima(img)
    labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
    if seeds.shape[0] == 0:
        print(f"corr {corr}: no seeds"); continue
    eng.ctx.set_profiling(True); eng.segment(img, seeds, out=labels); st = eng.stats(); eng.ctx.set_profiling(False)
    for _ in range(2): eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 5
    for _ in range(K): eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    for _ in range(1): eng.merge(img, seeds, out=labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): eng.merge(img, seeds, out=labels)
    torch.cuda.synchronize(); dm = (time.perf_counter() - t0) / 3
    print(f"corr {corr:4d}: seeds {seeds.shape[0]:8d}  segment {dt*1e3:8.3f} ms ({n*n/dt/1e9:6.1f} Gpx/s)  relax passes {st['relax_passes']:3d} tiles {st['tiles_run_relax']:7d} rounds/tile {st['relax_tile_iterations']/max(st['tiles_run_relax'],1):5.1f} ms_relax {st['ms_relax']:7.2f} ms_resolve {st['ms_resolve']:6.2f}  "
          f"coloured {int((labels != 0).sum())}  merge {dm*1e3:8.3f} ms", flush=True)
