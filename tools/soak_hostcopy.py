"""Soak of the host-side pipelines (csrc/ws_hostcopy.hip, ws_segment_batch): random planes of 2.1 .. 7 M pixels with 1 .. 8 widening
threads, the usize plane against the device's own u32 plane; random cubes through ws_segment_batch against the loop of calls.
ITER from the environment (default 60)."""
import ctypes, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
L = pkg._ffi.lib()
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ws = pkg.api.TransformBuilder().build_segmenting()
c, opt = ws._ctx(), ws._opt
n = ctypes.c_size_t(0)
failed = ctypes.c_size_t(0)
t0 = time.time()
for it in range(int(os.environ.get("ITER", "60"))):
    h = int(rng.integers(600, 2600)); w = int(rng.integers((1 << 21) // h + 1, 7_000_000 // h))
    img = np.ascontiguousarray(eng.random_field(h, w, 1000 + it).cpu().numpy())
    assert L.ws_ctx_set_host_threads(c.handle, int(rng.integers(1, 9))) == 0
    a = np.zeros((h, w), dtype=np.uint32); b = np.full(h * w + 1, 3, dtype=np.uint64); off = it % 2
    assert L.ws_segment_minima_u32(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), a.ctypes.data, None, 0, ctypes.byref(n)) == 0
    assert L.ws_segment_minima(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), b[off:].ctypes.data, None, 0, ctypes.byref(n)) == 0
    assert (b[off:off + h * w].reshape(h, w) == a).all(), ("plane", it, h, w)
    assert b[h * w if off == 0 else 0] == 3
    if it % 3 == 0:
        ns = int(rng.integers(2, 10)); sh = int(rng.integers(300, 1600)); sw = int(rng.integers(300, 1700))
        cube = np.stack([eng.random_field(sh, sw, 5000 + 10 * it + k).cpu().numpy() for k in range(ns)])
        loop = np.zeros((ns, sh, sw), dtype=np.uint64); batch = np.zeros((ns, sh, sw), dtype=np.uint64)
        for k in range(ns):
            assert L.ws_segment_minima(c.handle, cube[k].ctypes.data, sh, sw, sw, ctypes.byref(opt), loop[k].ctypes.data, None, 0, ctypes.byref(n)) == 0
        assert L.ws_segment_batch(c.handle, cube.ctypes.data, ns, sh, sw, sw, sh * sw, None, None, ctypes.byref(opt), batch.ctypes.data, None, ctypes.byref(failed)) == 0
        assert (loop == batch).all(), ("cube", it, ns, sh, sw)
    if it % 10 == 9: print(f"{it + 1} iterations, {time.time() - t0:.0f} s", flush=True)
print("soak ok", flush=True)
