#!/usr/bin/env python3
"""transform_to_list after a LARGER transform_to_list on the same context: which levels' records are wrong."""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_lib as ol, cases
big = cases.field(256, 256, 1)
bs = ol.find_local_minima(big)
mer = pkg.TransformBuilder.default().build_merging()
r = mer.transform_to_list_sparse(big, bs)
print("big done", len(bs), sum(len(x[2]) for x in r))
for maxlvl in (1, 60, 254):
    img = cases.field(72, 88, 9)
    seeds = ol.find_local_minima(img)
    want = []
    ol.merge(img, seeds, max_level=maxlvl, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(ol.canonicalise(c, seeds)[0])))
    ws = pkg.TransformBuilder.new().set_max_water_lvl(maxlvl).build_merging()
    got = ws.transform_to_list_sparse(img, seeds)
    bad = 0
    for (lvl, unc, cols, areas), w in zip(got, want):
        nz = np.nonzero(w[1:])[0] + 1
        ok = unc == w[0] and len(cols) == len(nz) and (np.sort(cols) == nz).all() and (areas[np.argsort(cols)] == w[nz]).all()
        if not ok:
            if bad < 4:
                extra = sorted(set(cols.tolist()) - set(nz.tolist()))[:5]; missing = sorted(set(nz.tolist()) - set(cols.tolist()))[:5]
                print(f"maxlvl {maxlvl} level {lvl}: records {len(cols)} want {len(nz)} unc {unc}/{w[0]}; extra colours {extra} missing {missing}")
            bad += 1
    print(f"maxlvl {maxlvl}: {bad} bad levels of {len(want)}; stats {ws._ctx().stats()['graph_launches']}")
