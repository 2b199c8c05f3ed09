"""A host cube of 4096^2 slices (BASELINE config 4's shape): the loop of ws_segment_minima calls against ws_segment_batch
(three lanes: upload / transform / label copy of different slices overlap).  SLICES, SIZE from the environment."""
import ctypes, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
N, S = int(os.environ.get("SLICES", "16")), int(os.environ.get("SIZE", "4096"))
cube = np.stack([eng.random_field(S, S, 100 + k).cpu().numpy() for k in range(N)])
L = pkg._ffi.lib()
ws = pkg.api.TransformBuilder().build_segmenting()
c, opt = ws._ctx(), ws._opt
out_loop = np.zeros((N, S, S), dtype=np.uint64)
out_batch = np.zeros((N, S, S), dtype=np.uint64)
n = ctypes.c_size_t(0)
counts = np.zeros(N, dtype=np.uintp)
failed = ctypes.c_size_t(0)
def loop():
    for k in range(N):
        assert L.ws_segment_minima(c.handle, cube[k].ctypes.data, S, S, S, ctypes.byref(opt), out_loop[k].ctypes.data, None, 0, ctypes.byref(n)) == 0
def batch():
    assert L.ws_segment_batch(c.handle, cube.ctypes.data, N, S, S, S, S * S, None, None, ctypes.byref(opt), out_batch.ctypes.data,
                              counts.ctypes.data_as(pkg._ffi.szp), ctypes.byref(failed)) == 0
def med(fn, runs=5, warm=2):
    ts = []
    for i in range(runs + warm):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        if i >= warm: ts.append(dt)
    ts.sort(); return ts[len(ts) // 2] * 1e3
a, b = med(loop), med(batch)
assert (out_loop == out_batch).all()
print(f"{N} x {S}^2 host cube: loop of ws_segment_minima {a:.1f} ms ({a / N:.2f} per slice), ws_segment_batch {b:.1f} ms ({b / N:.2f} per slice), "
      f"{N * S * S / b / 1e3:.0f} Mpixel/s, labels equal", flush=True)
