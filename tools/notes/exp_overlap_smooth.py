#!/usr/bin/env python3
"""Throughput of independent transforms driven by several host threads, one context (own stream) each, with the ordinary
blocking calls: do the host-driven pass loops of smooth maps overlap on the GPU?
usage: exp_overlap_smooth.py <noise|smoothC> [N]"""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
api = importlib.import_module("rustronomy_watershed_amd.api")
wl = sys.argv[1] if len(sys.argv) > 1 else "smooth64"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
eng = dev.DeviceEngine(0)
if wl == "noise":
    img = eng.random_field(n, n, 1)
else:
    corr = int(wl[6:])
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    low = torch.rand((1, 1, n // corr + 3, n // corr + 3), device=eng.device, generator=g)
    up = torch.nn.functional.interpolate(low, scale_factor=corr, mode="bicubic", align_corners=False)[0, 0, corr:corr + n, corr:corr + n]
    up = (up - up.min()) / (up.max() - up.min())
    img = (up * 253).to(torch.uint8).contiguous()
seeds = eng.find_local_minima(img)
K = 12
def worker(ctx, out, count):
    e = dev.DeviceEngine.__new__(dev.DeviceEngine)
    e.device = eng.device; e.ctx = ctx; e.engine = eng.engine
    for _ in range(count): e.segment(img, seeds, out=out)
for nctx in (1, 2, 3, 4):
    ctxs = [api.Context(0) for _ in range(nctx)]          # own stream each
    outs = [torch.empty((n, n), dtype=torch.int32, device=eng.device) for _ in range(nctx)]
    for c, o in zip(ctxs, outs): worker(c, o, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(c, o, K // nctx)) for c, o in zip(ctxs, outs)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done = (K // nctx) * nctx
    print(f"{wl} {n}: {nctx} thread(s) / context(s): {dt / done * 1e3:.3f} ms per transform  {done * n * n / dt / 1e9:.1f} Gpx/s", flush=True)
    assert all((o == outs[0]).all().item() for o in outs)
