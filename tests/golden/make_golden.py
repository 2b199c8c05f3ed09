#!/usr/bin/env python3
"""Generates the oracle fixtures under tests/golden/ (SURVEY 8c "golden vectors to commit").

The reference cannot run here (Rust; no toolchain) and holds no end-to-end outputs, so these
fixtures pin the deterministic CPU restatement (oracle/ws_oracle.c, tie-break col0): they are a
regression pin for the oracle and a second, file-based checker for the GPU path.  Inputs come from
the repo's own generator (mix64((seed << 40) + index) % 254).  Run: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import oracle_lib as ol  # noqa: E402


def lake_lists(fn, img, seeds, levels, **kw):
    out = {}
    fn(img, seeds, hook=lambda l, m, i, c: out.__setitem__(l, np.sort(ol.find_lake_sizes(c)[1:][ol.find_lake_sizes(c)[1:] > 0]))
       if l in levels else None, **kw)
    return out


def main():
    fx = {}
    for n, seed in ((16, 1), (64, 2), (256, 3)):
        img = cases.field(n, n, seed)
        seeds = ol.find_local_minima(img)
        seg = ol.segment(img, seeds)
        fx[f"f{n}_seed"] = np.array([seed])
        fx[f"f{n}_seeds"] = seeds.astype(np.uint32)
        fx[f"f{n}_seg"] = seg.astype(np.uint32)
        fx[f"f{n}_seg_edge"] = ol.segment(img, seeds, edge=True).astype(np.uint32)
        fx[f"f{n}_seg_max127"] = ol.segment(img, seeds, max_level=127).astype(np.uint32)
        levels = (0, 50, 127, 200, 254)
        ml = lake_lists(ol.merge, img, seeds, levels)
        for lvl in levels:
            fx[f"f{n}_merge_sizes_l{lvl}"] = ml[lvl].astype(np.uint32)
        fx[f"f{n}_merge_final"] = ol.canonicalise(ol.merge(img, seeds), seeds)[0].astype(np.uint32)
    for name, img, seeds in cases.adversarial_cases():
        if img.size == 0:
            continue
        seeds = cases.seeds_or_maxima(img, seeds)
        fx[f"adv_{name}_img"] = img
        fx[f"adv_{name}_seeds"] = np.asarray(seeds, dtype=np.uint32).reshape(-1, 2)
        fx[f"adv_{name}_seg"] = ol.segment(img, seeds).astype(np.uint32)
        fx[f"adv_{name}_seg_edge"] = ol.segment(img, seeds, edge=True).astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "oracle_fixtures.npz"), **fx)
    print("wrote", os.path.join(HERE, "oracle_fixtures.npz"), len(fx), "arrays")


if __name__ == "__main__":
    main()
