#!/usr/bin/env python3
"""The persistent pass's auto rule on maps it was not tuned on: smooth 8192^2 fields of other correlation lengths and other
generator seeds, the passes (mode 0) against the default (mode 3) and the queue forced (mode 2).
usage: exp_auto_check.py [N=8192] [corr ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
L = pkg._ffi.lib()
labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for corr in ([int(a) for a in sys.argv[2:]] or [8, 32, 48, 96, 128, 192]):
    for gseed in (3, 11):
        g = torch.Generator(device="cuda").manual_seed(gseed)
        low = torch.rand((1, 1, n // corr + 2, n // corr + 2), device="cuda", generator=g)
        up = torch.nn.functional.interpolate(low, size=(n, n), mode="bicubic", align_corners=False)[0, 0]
        up = (up - up.min()) / (up.max() - up.min())
        img = (up * 253.0).to(torch.uint8).contiguous()
        del low, up
        seeds = eng.find_local_minima(img)
        res = {}
        for mode in (0, 3, 2):
            assert L.ws_ctx_set_persistent_pass(eng.ctx.handle, mode) == 0
            for _ in range(2):
                eng.segment(img, seeds, out=labels)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.segment(img, seeds, out=labels)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[mode] = (sorted(ts)[1] * 1e3, int(labels.to(torch.int64).sum().item()))
        assert res[0][1] == res[3][1] == res[2][1], "labels differ"
        print(f"corr {corr:4d} gen {gseed:3d}: {seeds.shape[0]:8d} seeds  passes {res[0][0]:7.3f} ms  default {res[3][0]:7.3f} ms  queue forced {res[2][0]:7.3f} ms", flush=True)
