#!/usr/bin/env python3
"""Do two independent transforms overlap usefully on one GPU?  K transforms on one context vs the same K
split over two contexts (own streams) driven by two host threads."""
import os, sys, time, threading, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
api = importlib.import_module("rustronomy_watershed_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eng = dev.DeviceEngine(0)
img = eng.random_field(n, n, 1); seeds = eng.find_local_minima(img)
K = 20
def worker(ctx, out, count):
    e = dev.DeviceEngine.__new__(dev.DeviceEngine)
    e.device = eng.device; e.ctx = ctx; e.engine = eng.engine
    for _ in range(count): e.segment(img, seeds, out=out)
for nctx in (1, 2, 3):
    ctxs = [api.Context(0) for _ in range(nctx)]          # own stream each
    outs = [torch.empty((n, n), dtype=torch.int32, device=eng.device) for _ in range(nctx)]
    for c, o in zip(ctxs, outs): worker(c, o, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(c, o, K // nctx)) for c, o in zip(ctxs, outs)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done = (K // nctx) * nctx
    print(f"{nctx} context(s): {dt / done * 1e3:.3f} ms per transform  {done * n * n / dt / 1e9:.1f} Gpx/s", flush=True)
