#!/usr/bin/env python3
"""In-process rehearsal of the tiled protocol (distributed.segment_tiled) without torch.distributed: `world` block
engines on one device, halo rows and the boundary table moved by tensor copies.  Prints where the result differs from
the single-domain transform.  usage: debug_tiled.py [N=8192] [world=2] [seed=5]"""
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
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if os.environ.get("WS_OWN_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
full = eng.random_field(n, n, seed)
seeds = eng.find_local_minima(full)
want = eng.segment(full, seeds)
keys_want = eng.last_arrival()
torch.cuda.synchronize()
blocks, spans = [], []
for r in range(world):
    r0, r1, lo, hi = wd.row_block(n, r, world)
    loc, col = wd.local_seeds(seeds, lo, hi)
    b = wd.HipBlockEngine(dev.DeviceEngine(0), full[lo:hi].contiguous(), loc, col)
    b.set_halos(r > 0, r < world - 1)
    blocks.append(b); spans.append((r0, r1, lo, hi))
    assert b.try_begin(), f"fast form refused: {getattr(b, 'why_not_fast', 'engine said no')}"
def keys_diff(tag):
    for r, b in enumerate(blocks):
        r0, r1, lo, hi = spans[r]
        own = b.keys[r0 - lo:r1 - lo]
        bad = (own != keys_want[r0:r1])
        print(f"  {tag}: rank {r} wrong stamps {int(bad.sum())}", end="")
        if bad.any():
            rows = torch.nonzero(bad.any(dim=1)).flatten()
            print(f" rows {int(rows.min()) + r0}..{int(rows.max()) + r0}", end="")
        print()
keys_diff("after begin")
for rnd in range(20):
    new = False
    recv = []
    for r, b in enumerate(blocks):
        if r > 0: recv.append((b, 0, blocks[r - 1].keys[-2].clone()))
        if r < world - 1: recv.append((b, b.h - 1, blocks[r + 1].keys[1].clone()))
    for b, row, t in recv:
        new |= bool((t != b.keys[row]).any())
    print(f"round {rnd}: new halo rows {new}")
    if not new: break
    for b, row, t in recv: b.keys[row].copy_(t)
    for b in blocks: b.relax_halo()
    torch.cuda.synchronize()
    keys_diff(f"round {rnd}")
for b in blocks: b.resolve_local()
rows = [b.export_boundary(r) for r, b in enumerate(blocks)]
table = torch.stack(rows).reshape(-1).contiguous()
for r, b in enumerate(blocks): b.import_boundary(table, r, world)
torch.cuda.synchronize()
for r, b in enumerate(blocks):
    r0, r1, lo, hi = spans[r]
    own = b.labels[r0 - lo:r1 - lo]
    bad = own != want[r0:r1]
    print(f"rank {r}: wrong labels {int(bad.sum())} of {own.numel()}", end="")
    if bad.any():
        rows_ = torch.nonzero(bad.any(dim=1)).flatten()
        print(f" rows {int(rows_.min()) + r0}..{int(rows_.max()) + r0}; refs left {int((own < 0).sum())}", end="")
    print()
# timing of the single-domain call on this stream
for _ in range(3): eng.segment(full, seeds, out=want)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): eng.segment(full, seeds, out=want)
torch.cuda.synchronize(); print(f"single domain {n}x{n}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms, stats {eng.stats()}")
