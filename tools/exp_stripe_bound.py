#!/usr/bin/env python3
"""Upper bound on what cutting ONE transform into S concurrently scheduled row stripes could buy (round 3): S independent
fields of 8192 / S rows x 8192 columns, one per context and stream, begun together and ended together, against one
8192 x 8192 transform.  Independent stripes have no seam dependencies, so a striped transform cannot beat this."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
torch.cuda.set_stream(torch.cuda.Stream(0))
W = 8192
for S in (1, 2, 4, 8):
    H = 8192 // S
    engs, imgs, seeds, labs = [], [], [], []
    for i in range(S):
        with torch.cuda.stream(torch.cuda.Stream(0)):
            e = dev.DeviceEngine(0)
        e.ctx.check(0)
        pkg._ffi.lib().ws_ctx_set_seam_repair_min_pixels(e.ctx.handle, 1 << 20)      # the stripes take the big-plane flow too
        engs.append(e)
        im = e.random_field(H, W, 1 + i)
        imgs.append(im); seeds.append(e.find_local_minima(im).clone()); labs.append(torch.empty((H, W), dtype=torch.int32, device=e.device))
    for _ in range(4):
        for e, im, s, l in zip(engs, imgs, seeds, labs):
            e.segment(im, s, out=l)
    torch.cuda.synchronize()
    K = 20
    for delay_us in (0, 20, 40, 60, 90, 130):
        if S == 1 and delay_us:
            continue
        t0 = time.perf_counter()
        for _ in range(K):
            for j, (e, im, s, l) in enumerate(zip(engs, imgs, seeds, labs)):
                if j and delay_us:      # a staggered start: stripe j begins delay_us after stripe j - 1 (busy wait)
                    t1 = time.perf_counter()
                    while (time.perf_counter() - t1) * 1e6 < delay_us:
                        pass
                e.segment_begin(im, s, l)
            for e in engs:
                e.segment_end()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / K
        print(f"S={S}: {S} independent stripes of {H} x {W}, begun {delay_us} us apart: {ms:.4f} ms for the whole 8192 x 8192 worth of pixels", flush=True)
