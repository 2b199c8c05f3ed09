#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer C ABI (what the Rust shim would call): pageable u8 image and
(row, col) u64 seeds in, u64 label plane out.  Also times find_local_minima through the host ABI."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import oracle_lib as ol
out = []
for n in (2048, 8192):
    img = ol.random_field(n, n, 1)
    ws = pkg.TransformBuilder.default().build_segmenting()
    seeds = ws.find_local_minima(img)
    ws.transform(img, seeds)                       # warm-up (workspace allocation)
    t0 = time.perf_counter(); K = 3
    for _ in range(K): seeds = ws.find_local_minima(img)
    t_min = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K): lab = ws.transform(img, seeds)
    t_seg = (time.perf_counter() - t0) / K
    out.append({"size": n, "seeds": int(len(seeds)), "find_local_minima_ms": round(t_min * 1e3, 2),
                "transform_ms": round(t_seg * 1e3, 2), "transform_Mpx_per_s": round(n * n / t_seg / 1e6, 1),
                "bytes_over_pcie": int(img.nbytes + seeds.nbytes + lab.nbytes)})
print(json.dumps({"host_api_pcie_inclusive": out}))
