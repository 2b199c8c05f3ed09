"""Latency of one transform at the small BASELINE sizes (configs 0 and 1: 512^2, 2048^2) and around them: the device-resident
call, the host call (u8 image + u64 seeds in, u64 labels out) and the README's pair as one host call."""
import ctypes, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))      # not the legacy null stream: the engine's graph replay needs a capturable stream (as bench.py)
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
L = pkg._ffi.lib()
ws = pkg.api.TransformBuilder().build_segmenting()
c, opt = ws._ctx(), ws._opt

def med(fn, runs=30, warm=5):
    ts = []
    for i in range(runs + warm):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        if i >= warm: ts.append(dt)
    ts.sort()
    return ts[len(ts) // 2] * 1e3

for S in [int(x) for x in os.environ.get("SIZES", "256,512,1024,2048,4096").split(",")]:
    img_t = eng.random_field(S, S, 1)
    seeds_t = eng.find_local_minima(img_t)
    out_t = torch.empty((S, S), dtype=torch.int32, device=img_t.device)
    def dev():
        eng.segment(img_t, seeds_t, out=out_t); torch.cuda.synchronize()
    def dev_pair():
        eng.segment_minima(img_t, out=out_t); torch.cuda.synchronize()
    img = np.ascontiguousarray(img_t.cpu().numpy())
    seeds = np.ascontiguousarray(seeds_t.cpu().numpy().astype(np.uint64))
    out = np.zeros((S, S), dtype=np.uint64)
    n = ctypes.c_size_t(0)
    def host():
        assert L.ws_segment(c.handle, img.ctypes.data, S, S, S, seeds.ctypes.data, len(seeds), ctypes.byref(opt), out.ctypes.data) == 0
    def host_pair():
        assert L.ws_segment_minima(c.handle, img.ctypes.data, S, S, S, ctypes.byref(opt), out.ctypes.data, None, 0, ctypes.byref(n)) == 0
    r = (med(dev), med(dev_pair), med(host), med(host_pair))
    print(f"{S}x{S}: device {r[0]:.3f} ms, device pair-as-one {r[1]:.3f} ms, host ws_segment {r[2]:.3f} ms, host ws_segment_minima {r[3]:.3f} ms  ({len(seeds)} seeds)", flush=True)
