#!/usr/bin/env python3
"""Ad-hoc large parity check (too slow for the test suite: the CPU oracle takes seconds per field): 4096^2 random,
3000x5000 smooth and an odd-sized random field, segmenting and merging final labels against the arrival-form oracle."""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build_hip(); pkg = ge.load_package()
import oracle_lib as ol, cases
ws = pkg.TransformBuilder.new().set_engine(pkg.ENGINE_FUSED).build_segmenting()
mer = pkg.TransformBuilder.new().build_merging()
for name, img in [("rand4096", ol.random_field(4096, 4096, 9)), ("smooth3000x5000", cases.smooth_field(3000, 5000, 4, octaves=7)),
                  ("rand_odd 3071x4097", ol.random_field(3071, 4097, 5))]:
    seeds = ol.find_local_minima(img)
    t0 = time.time(); want = ol.segment_arrival(img, seeds); t1 = time.time()
    got = ws.transform(img, seeds)
    ok = bool((got == want).all())
    wm = ol.merge_arrival(img, seeds); gm = mer.transform_final(img, seeds)
    print(name, "seeds", len(seeds), "segment ok", ok, "merge ok", bool((gm == wm).all()), "oracle %.1fs" % (t1 - t0), flush=True)
