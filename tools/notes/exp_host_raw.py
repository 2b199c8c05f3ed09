#!/usr/bin/env python3
"""The host ABI (ws_find_local_minima, ws_segment) through ctypes with REUSED, already touched host buffers: the
library's own PCIe-inclusive time, without the first-touch page faults of fresh output arrays."""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import importlib, oracle_lib as ol
ffi = importlib.import_module("rustronomy_watershed_amd._ffi")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
img = ol.random_field(m, m, 1)
ws = pkg.TransformBuilder.default().build_segmenting()
ctx = ws._ctx(); opt = ws._opt
cap = (m // 2 + 1) * (m // 2 + 1)
seeds = np.zeros((cap, 2), dtype=np.uint64)
labels = np.zeros((m, m), dtype=np.uint64)
n = ctypes.c_size_t(0)
def minima():
    rc = ffi.lib().ws_find_local_minima(ctx.handle, img.ctypes.data, m, m, m, seeds.ctypes.data, cap, ctypes.byref(n)); assert rc == 0, rc
def segment():
    rc = ffi.lib().ws_segment(ctx.handle, img.ctypes.data, m, m, m, seeds.ctypes.data, n.value, ctypes.byref(opt), labels.ctypes.data); assert rc == 0, rc
for f, name in ((minima, "ws_find_local_minima"), (segment, "ws_segment")):
    for _ in range(2): f()
    t0 = time.perf_counter(); K = 4
    for _ in range(K): f()
    dt = (time.perf_counter() - t0) / K
    nb = m * m + n.value * 16 + (m * m * 8 if f is segment else 0)
    print(f"{name} {m}x{m} (reused buffers): {dt*1e3:.2f} ms  {nb/1e6:.0f} MB over PCIe = {nb/dt/1e9:.1f} GB/s, {n.value} seeds")
