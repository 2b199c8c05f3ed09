#!/usr/bin/env python3
"""Per-level hook through the host ABI (segmenting and merging, 2048x2048, trivial hook): wall clock."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import oracle_lib as ol
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
img = ol.random_field(m, m, 1)
for name, build in (("segmenting", "build_segmenting"), ("merging", "build_merging")):
    seen = []
    b = pkg.TransformBuilder.new().set_wlvl_hook(lambda ctx: seen.append(int(ctx.colours[m // 2, m // 2])))
    ws = getattr(b, build)()
    seeds = ws.find_local_minima(img)
    ws.transform_with_hook(img, seeds)
    seen.clear(); t0 = time.perf_counter()
    ws.transform_with_hook(img, seeds)
    dt = time.perf_counter() - t0
    print(f"{name} transform_with_hook {m}x{m}: {dt*1e3:.1f} ms for {len(seen)} levels ({m*m*8*len(seen)/dt/1e9:.1f} GB/s of planes)")
