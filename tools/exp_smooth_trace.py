#!/usr/bin/env python3
"""ONE smooth-field segmenting transform after two warm-ups, for `rocprofv3 --kernel-trace` (tools/trace_passes.py lists
the last transform's launches).  usage: exp_smooth_trace.py [N=8192] [corr=64] [seed_stride=1] [persistent_pass=0]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
corr = int(sys.argv[2]) if len(sys.argv) > 2 else 64
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
if len(sys.argv) > 4:
    pkg = importlib.import_module("rustronomy_watershed_amd")
    assert pkg._ffi.lib().ws_ctx_set_persistent_pass(eng.ctx.handle, int(sys.argv[4])) == 0
g = torch.Generator(device="cuda").manual_seed(3)
low = torch.rand((1, 1, n // corr + 2, n // corr + 2), device="cuda", generator=g)
up = torch.nn.functional.interpolate(low, size=(n, n), mode="bicubic", align_corners=False)[0, 0]
up = (up - up.min()) / (up.max() - up.min())
img = (up * 253.0).to(torch.uint8).contiguous()
seeds = eng.find_local_minima(img)[::stride].contiguous()
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
eng.ctx.set_profiling(True); eng.segment(img, seeds, out=labels); st = eng.stats(); eng.ctx.set_profiling(False)
eng.segment(img, seeds, out=labels)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.segment(img, seeds, out=labels)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{n}x{n} corr {corr}: {seeds.shape[0]} seeds, {dt*1e3:.2f} ms, passes {st['relax_passes']}, tiles run {st['tiles_run_relax']}, "
      f"rounds/tile {st['relax_tile_iterations']/max(st['tiles_run_relax'],1):.1f}, ms_relax {st['ms_relax']:.2f} ms_resolve {st['ms_resolve']:.2f}", flush=True)
