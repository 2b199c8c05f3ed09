#!/usr/bin/env python3
"""BASELINE config C3 (8192x8192 merging, 1 GPU) and the reference's own core_bench shape
(tests/core_bench.rs: merging transform_to_list on 1024x1024), with the CPU oracle beside it."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import importlib, oracle_lib as ol
dev = importlib.import_module("rustronomy_watershed_amd.device")
eng = dev.DeviceEngine(0)
res = {}
n = 8192
img = eng.random_field(n, n, 1); seeds = eng.find_local_minima(img)
out = torch.empty((n, n), dtype=torch.int32, device=eng.device)
for _ in range(2): eng.merge(img, seeds, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 10
for _ in range(K): eng.merge(img, seeds, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
res["C3_merge_final_labels_8192"] = {"ms": round(dt * 1e3, 3), "Mpixels_per_s": round(n * n / dt / 1e6, 1), "lakes_left": int(torch.unique(out).numel() - 1)}
# core_bench shape through the host API (sparse lists), GPU vs CPU oracle (canonical union-find form)
m = 1024
himg = ol.random_field(m, m, 1)
ws = pkg.TransformBuilder.default().build_merging()
hseeds = ws.find_local_minima(himg)
ws.transform_to_list_sparse(himg, hseeds)
t0 = time.perf_counter(); r = ws.transform_to_list_sparse(himg, hseeds); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); ol.merge(himg, hseeds, mode=ol.MAP_CANONICAL, hook=lambda l, mx, i, c: None); t_cpu = time.perf_counter() - t0
res["core_bench_1024_merging_to_list"] = {"gpu_ms_host_api": round(t_gpu * 1e3, 2), "cpu_oracle_s_1_thread": round(t_cpu, 2),
                                          "lake_records": int(sum(len(x[2]) for x in r))}
print(json.dumps(res))
