#!/usr/bin/env python3
"""Config C4's per-GPU share: a batch of S independent N x N slices through ws_segment_batch_device; wall clock per batch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
eng = dev.DeviceEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cube = torch.stack([eng.random_field(n, n, k + 1) for k in range(s)]).contiguous()
seed_lists = [eng.find_local_minima(cube[k]) for k in range(s)]
offs = [0]
for t in seed_lists: offs.append(offs[-1] + t.shape[0])
seeds = torch.cat(seed_lists).contiguous()
out = torch.empty((s, n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.segment_batch(cube, seeds, offs, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 5
for _ in range(K): eng.segment_batch(cube, seeds, offs, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"batch {s} x {n}x{n}: {dt*1e3:.3f} ms  {s*n*n/dt/1e9:.2f} Gpx/s  ({dt*1e3/s:.3f} ms per slice)")
one = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.segment(cube[0], seed_lists[0], out=one)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): eng.segment(cube[0], seed_lists[0], out=one)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"single {n}x{n}: {dt*1e3:.3f} ms  {n*n/dt/1e9:.2f} Gpx/s")
assert torch.equal(out[0], one)
