"""ws_transform_to_list with host buffers (merging, all levels): records narrowed to u32 over the bus and widened by the host's
threads against one 16-byte copy per group of levels (ws_ctx_set_host_threads 4 / 0).  SIZES from the environment."""
import ctypes, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
L = pkg._ffi.lib()
for S in [int(x) for x in os.environ.get("SIZES", "1024,2048,4096").split(",")]:
    himg = np.ascontiguousarray(eng.random_field(S, S, 21).cpu().numpy())
    t = eng.find_local_minima(eng.random_field(S, S, 21))
    hseeds = np.ascontiguousarray(t.cpu().numpy().astype(np.uint64))
    ws = pkg.api.TransformBuilder().build_merging()
    c = ws._ctx()
    cap = 10 * S * S
    rec = np.zeros((cap, 2), dtype=np.uint64)
    n = ctypes.c_size_t(0)
    off, unc = np.zeros(256, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
    res = []
    for threads in (4, 0):
        assert L.ws_ctx_set_host_threads(c.handle, threads) == 0
        ts = []
        for i in range(5):
            t0 = time.perf_counter()
            rc = L.ws_transform_to_list(c.handle, 1, himg.ctypes.data, S, S, S, hseeds.ctypes.data, len(hseeds), ctypes.byref(ws._opt), rec.ctypes.data, cap,
                                        ctypes.byref(n), off.ctypes.data, unc.ctypes.data)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0, rc
        res.append(sorted(ts[2:])[1])
    print(f"{S}^2: {n.value} records; host lists {res[0]:.1f} ms with the records as u32 over the bus, {res[1]:.1f} ms as 16-byte records", flush=True)
