#!/usr/bin/env python3
"""The reference's own benchmark shape (tests/core_bench.rs): merging transform_to_list on a 1024x1024
random field, through the host ABI (sparse records).  Prints wall clock per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import oracle_lib as ol
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
himg = ol.random_field(m, m, 1)
ws = pkg.TransformBuilder.default().build_merging()
hseeds = ws.find_local_minima(himg)
for _ in range(2): ws.transform_to_list_sparse(himg, hseeds)
t0 = time.perf_counter(); K = 3
for _ in range(K): r = ws.transform_to_list_sparse(himg, hseeds)
dt = (time.perf_counter() - t0) / K
print(f"merging transform_to_list {m}x{m}: {dt*1e3:.2f} ms, {sum(len(x[2]) for x in r)} lake records")
