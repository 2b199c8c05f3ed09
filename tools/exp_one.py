#!/usr/bin/env python3
"""One workload, one line: median time of the device-resident segmenting transform, passes, tile runs and a checksum of
the labels (to compare tuning-knob settings: run under WS_HIP_LIB=<tuning build> with WS_* knobs set).
usage: exp_one.py <noise|smoothC[/stride]> [N=8192] [runs=5]      (smooth64 = correlation length 64; /s = every s-th seed)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
stride = 1
if "/" in kind:
    kind, s = kind.split("/"); stride = int(s)
if kind == "noise":
    img = eng.random_field(n, n, 1)
else:
    corr = int(kind[len("smooth"):])
    g = torch.Generator(device="cuda").manual_seed(3)
    low = torch.rand((1, 1, n // corr + 2, n // corr + 2), device="cuda", generator=g)
    up = torch.nn.functional.interpolate(low, size=(n, n), mode="bicubic", align_corners=False)[0, 0]
    up = (up - up.min()) / (up.max() - up.min())
    img = (up * 253.0).to(torch.uint8).contiguous()
    del low, up
seeds = eng.find_local_minima(img)[::stride].contiguous()
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
eng.ctx.set_profiling(True); eng.segment(img, seeds, out=labels); st = eng.stats(); eng.ctx.set_profiling(False)
for _ in range(3 if kind != "noise" else 10): eng.segment(img, seeds, out=labels)
ts = []
for _ in range(runs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
w = torch.arange(n * n, device=eng.device, dtype=torch.int64).reshape(n, n) % 1000003
chk = int(((labels.to(torch.int64) & 0xFFFFFFFF) * w).sum().item()) & 0xFFFFFFFFFFFF
knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("WS_") and k != "WS_HIP_LIB")
print(f"{kind}/{stride} {n}: {ts[len(ts)//2]*1e3:8.3f} ms (min {ts[0]*1e3:.3f})  seeds {seeds.shape[0]} passes {st['relax_passes']} tiles {st['tiles_run_relax']} "
      f"rounds/tile {st['relax_tile_iterations']/max(st['tiles_run_relax'],1):.2f} ms_relax {st['ms_relax']:.2f} ms_resolve {st['ms_resolve']:.2f} chk {chk:012x}  [{knobs}]", flush=True)
