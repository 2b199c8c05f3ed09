#!/usr/bin/env python3
"""Writes the smooth field of tools/exp_smooth.py (CPU torch, same recipe) and its seeds for sim_tile_schedule.c:
   sim_make_field.py N corr out_prefix"""
import sys
import numpy as np
import torch
n, corr, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
g = torch.Generator().manual_seed(3)
low = torch.rand((1, 1, n // corr + 2, n // corr + 2), generator=g)
up = torch.nn.functional.interpolate(low, size=(n, n), mode="bicubic", align_corners=False)[0, 0]
up = (up - up.min()) / (up.max() - up.min())
img = (up * 253.0).to(torch.uint8).numpy()
c = img[1:-1, 1:-1]
ok = np.ones_like(c, dtype=bool)
for dr in (-1, 0, 1):
    for dc in (-1, 0, 1):
        if dr or dc:
            ok &= img[1 + dr:n - 1 + dr, 1 + dc:n - 1 + dc] < c      # lib.rs:1190: every neighbour below the centre
r, cc = np.nonzero(ok)
seeds = ((r + 1) * n + (cc + 1)).astype(np.uint32)
img.tofile(out + ".u8"); seeds.tofile(out + ".seeds")
print(n, corr, len(seeds))
