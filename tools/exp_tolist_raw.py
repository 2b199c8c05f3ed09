#!/usr/bin/env python3
"""ws_transform_to_list through ctypes with REUSED (already touched) host buffers: the library's own time for the
reference's core_bench shape, without the first-touch page faults of a fresh output array."""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import importlib, oracle_lib as ol
ffi = importlib.import_module("rustronomy_watershed_amd._ffi")
api = importlib.import_module("rustronomy_watershed_amd.api")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img = ol.random_field(m, m, 1)
ws = pkg.TransformBuilder.default().build_merging()
seeds = np.ascontiguousarray(ws.find_local_minima(img), dtype=np.uint64)
ctx = ws._ctx()
levels = 255
cap = len(seeds) * 128 + 1024
lakes = np.zeros((cap, 2), dtype=np.uint64)            # touched once
offsets = np.zeros(levels + 1, dtype=np.uint64); unc = np.zeros(levels, dtype=np.uint64)
n = ctypes.c_size_t(0)
opt = ws._opt
def call():
    rc = ffi.lib().ws_transform_to_list(ctx.handle, 1, img.ctypes.data, m, m, m, seeds.ctypes.data, len(seeds), ctypes.byref(opt),
                                        lakes.ctypes.data, cap, ctypes.byref(n), offsets.ctypes.data, unc.ctypes.data)
    assert rc == 0, rc
for _ in range(2): call()
t0 = time.perf_counter(); K = 5
for _ in range(K): call()
dt = (time.perf_counter() - t0) / K
print(f"ws_transform_to_list {m}x{m} (reused buffers): {dt*1e3:.2f} ms, {n.value} records")
# the same without the record copies (cap = 0: every level is computed, nothing is copied, the call reports "capacity")
def call_nocopy():
    rc = ffi.lib().ws_transform_to_list(ctx.handle, 1, img.ctypes.data, m, m, m, seeds.ctypes.data, len(seeds), ctypes.byref(opt),
                                        lakes.ctypes.data, 0, ctypes.byref(n), offsets.ctypes.data, unc.ctypes.data)
    assert rc != 0
for _ in range(2): call_nocopy()
t0 = time.perf_counter()
for _ in range(K): call_nocopy()
print(f"  without the record copies: {(time.perf_counter() - t0) / K * 1e3:.2f} ms")
ctx.set_profiling(True)
call(); call()
st = ctx.stats()
print("device ms: total %.2f relax %.2f resolve %.2f other %.2f; merge_levels %d" % (st["ms_total"], st["ms_relax"], st["ms_resolve"], st["ms_other"], st["merge_levels"]))
