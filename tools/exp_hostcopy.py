"""Times ws_segment_minima (u64 host labels) at 8192^2: the chunked copy of ws_hostcopy.hip under WS_HOST_THREADS /
WS_HOST_CHUNK_LOG2 (tuning build) -- one setting per process (the knobs are read once)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
H = W = int(os.environ.get("SIZE", "8192"))
import importlib
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
img = np.ascontiguousarray(eng.random_field(H, W, 1).cpu().numpy())
L = pkg._ffi.lib()
ws = pkg.api.TransformBuilder().build_segmenting()
c, opt = ws._ctx(), ws._opt
out = np.zeros(H * W, dtype=np.uint64)
out32 = np.zeros(H * W, dtype=np.uint32)
n = ctypes.c_size_t(0)
def run(f, *a):
    ts = []
    for i in range(7):
        t0 = time.perf_counter(); rc = f(*a); ts.append((time.perf_counter() - t0) * 1e3); assert rc == 0
    return sorted(ts[2:])[len(ts[2:]) // 2]
m64 = run(L.ws_segment_minima, c.handle, img.ctypes.data, H, W, W, ctypes.byref(opt), out.ctypes.data, None, 0, ctypes.byref(n))
m32 = run(L.ws_segment_minima_u32, c.handle, img.ctypes.data, H, W, W, ctypes.byref(opt), out32.ctypes.data, None, 0, ctypes.byref(n))
assert (out == out32).all()
print(f"threads {os.environ.get('WS_HOST_THREADS', 'default')} chunk 2^{os.environ.get('WS_HOST_CHUNK_LOG2', '22')}: u64 {m64:.2f} ms, u32 {m32:.2f} ms", flush=True)
