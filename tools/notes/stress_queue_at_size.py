#!/usr/bin/env python3
"""Ad-hoc: random smooth maps of 3000-5000 px a side with thinned seed lists under the default schedule, every label against
the oracle (tests/oracle_lib).  usage: stress_queue_at_size.py [cases=6] [seed=1]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import cases, oracle_lib as ol
pkg = ge.load_package()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    h, w = int(rng.integers(3000, 5000)), int(rng.integers(750, 1250)) * 4
    img = cases.smooth_field(h, w, int(rng.integers(1, 1000)), octaves=int(rng.integers(6, 10)))
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    keep = int(rng.choice([1, 5, 50, 500, 5000]))
    seeds = seeds[::keep] if len(seeds) > keep else seeds
    ml = int(rng.choice([254, 200, 120]))
    t0 = time.time(); want = ol.segment_arrival(img, seeds, max_level=ml); t1 = time.time()
    ws = pkg.TransformBuilder.new().set_engine(pkg.ENGINE_FUSED).set_max_water_lvl(ml).build_segmenting()
    got = ws.transform(img, seeds)
    st = ws._ctx().stats()
    bad = int((got != want).sum())
    print(f"case {case}: {h}x{w} seeds {len(seeds)} max_level {ml}: passes {st['relax_passes']} bad {bad} (oracle {t1-t0:.1f} s)", flush=True)
    assert bad == 0
print("ok")
