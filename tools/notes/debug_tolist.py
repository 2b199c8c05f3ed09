#!/usr/bin/env python3
"""transform_to_list (merging) against the oracle, level by level: first level whose records differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import oracle_lib as ol, cases
maxlvl = int(sys.argv[1]) if len(sys.argv) > 1 else 60
img = cases.field(72, 88, 9)
seeds = ol.find_local_minima(img)
want = []
ol.merge(img, seeds, max_level=maxlvl, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(ol.canonicalise(c, seeds)[0])))
ws = pkg.TransformBuilder.new().set_max_water_lvl(maxlvl).build_merging()
for rep in range(3):
    got = ws.transform_to_list_sparse(img, seeds)
    bad = 0
    for (lvl, unc, cols, areas), w in zip(got, want):
        nz = np.nonzero(w[1:])[0] + 1
        ok = unc == w[0] and len(cols) == len(nz) and (np.sort(cols) == nz).all() and (areas[np.argsort(cols)] == w[nz]).all()
        if not ok:
            if bad < 3:
                print(f"rep {rep} level {lvl}: records {len(cols)} want {len(nz)} unc {unc} want {w[0]}; max colour {cols.max() if len(cols) else 0}; sum areas {int(areas.sum())} want {int(w[1:].sum())}")
            bad += 1
    print(f"rep {rep}: {bad} bad levels of {len(want)}")
