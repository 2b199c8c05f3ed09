#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are collected in separate
runs: MI355X_MICROARCH.md, rocprofv3 PMC slots).  Runs, on one 8192x8192 field:
  1. two calibration kernels of known size, one per access width the engine uses: a few steps of the
     sweep engine's k_flood_step (4 B per lane: reads the u8 image and the u32 label plane once,
     5N bytes, writes 4N) and one device-to-device copy of the stamp plane (16 B per lane: 4N bytes
     in, 4N bytes out);
  2. `steps` segmenting transforms through the C ABI.
tools/pmc_summarise.py turns the two counter CSVs into per-kernel bytes per launch."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--merge", action="store_true")
    ap.add_argument("--tolist", action="store_true", help="the merging transform_to_list (ws_transform_to_list_device), records left in HBM")
    args = ap.parse_args()
    ge.build_hip()
    ge.load_package()
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    n = args.size
    img = eng.random_field(n, n, 1)
    seeds = eng.find_local_minima(img)
    labels = torch.empty((n, n), dtype=torch.int32, device=eng.device)
    pkg = sys.modules["rustronomy_watershed_amd"]
    eng.segment(img, seeds, max_level=1, engine=pkg.ENGINE_SWEEP, out=labels)   # calibration launches (4 B/lane)
    eng.segment(img, seeds, out=labels)
    keys_copy = eng.last_arrival()      # calibration: one 4N-byte device-to-device copy with 16 B/lane accesses
    torch.cuda.synchronize()
    buf = None
    if args.tolist:
        _, off, _ = eng.transform_to_list(img, seeds)
        buf = torch.empty((int(off[-1]) + 16, 2), dtype=torch.int64, device=eng.device)
        torch.cuda.synchronize()
    for _ in range(args.steps):
        if args.tolist:
            eng.transform_to_list(img, seeds, lakes=buf)
        elif args.merge:
            eng.merge(img, seeds, out=labels)
        else:
            eng.segment(img, seeds, out=labels)
    torch.cuda.synchronize()
    print("pmc_probe done", n, int(seeds.shape[0]), eng.stats())


if __name__ == "__main__":
    main()
