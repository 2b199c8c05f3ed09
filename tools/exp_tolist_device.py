#!/usr/bin/env python3
"""ws_transform_to_list_device (records stay in HBM) against the host form.  usage: exp_tolist_device.py [N=1024]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import importlib
dev = importlib.import_module("rustronomy_watershed_amd.device")
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = dev.DeviceEngine(0)
for n in [int(a) for a in sys.argv[1:]] or [1024]:
    img = eng.random_field(n, n, 1)
    seeds = eng.find_local_minima(img)
    lakes, off, unc = eng.transform_to_list(img, seeds)
    buf = torch.empty((int(off[-1]) + 16, 2), dtype=torch.int64, device=eng.device)
    for _ in range(3): eng.transform_to_list(img, seeds, lakes=buf)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 5
    for _ in range(K): eng.transform_to_list(img, seeds, lakes=buf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"ws_transform_to_list_device {n}x{n} merging: {dt*1e3:.2f} ms, {int(off[-1])} records ({int(off[-1])*16/1e6:.0f} MB stay in HBM), graph launches {eng.stats()['graph_launches']}")
