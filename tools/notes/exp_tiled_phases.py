#!/usr/bin/env python3
"""Phases of the tiled protocol on ONE block (the upper half of an N x N field), timed in one process: what a rank pays
besides the collectives.  usage: exp_tiled_phases.py [N=8192]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
wd = importlib.import_module("rustronomy_watershed_amd.distributed")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
full = eng.random_field(n, n, 5)
seeds = eng.find_local_minima(full)
blocks = []
for r in range(2):
    r0, r1, lo, hi = wd.row_block(n, r, 2)
    loc, col = wd.local_seeds(seeds, lo, hi)
    b = wd.HipBlockEngine(dev.DeviceEngine(0), full[lo:hi].contiguous(), loc, col)
    b.set_halos(r > 0, r < 1)
    blocks.append(b)
def sync(): torch.cuda.synchronize()
def timed(f, reps=1):
    sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    sync(); return (time.perf_counter() - t0) / reps * 1e3
for rep in range(3):
    t_begin = timed(lambda: [b.try_begin() for b in blocks]) / 2
    rounds = []
    for rnd in range(8):
        recv = [(blocks[1], 0, blocks[0].keys[-2].clone()), (blocks[0], blocks[0].h - 1, blocks[1].keys[1].clone())]
        new = any(bool((t != b.keys[row]).any()) for b, row, t in recv)
        if not new: break
        for b, row, t in recv: b.keys[row].copy_(t)
        rounds.append(timed(lambda: [b.relax_halo() for b in blocks]) / 2)
    t_res = timed(lambda: [b.resolve_local() for b in blocks]) / 2
    rows = [b.export_boundary(r) for r, b in enumerate(blocks)]
    table = torch.stack(rows).reshape(-1).contiguous()
    t_imp = timed(lambda: [b.import_boundary(table, r, 2) for r, b in enumerate(blocks)]) / 2
    print(f"rep {rep}: per block: begin {t_begin:.3f} ms, relax_halo rounds {[round(x, 3) for x in rounds]}, resolve_local {t_res:.3f}, import+chase {t_imp:.3f} ms", flush=True)
lab = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(5): eng.segment(full, seeds, out=lab)
print(f"single domain: {timed(lambda: eng.segment(full, seeds, out=lab), 10):.3f} ms")
