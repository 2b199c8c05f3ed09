"""Experiment (CPU, oracle as solver): how deep do cross-tile dependencies reach?

For a random field, solve each 256x32 tile in isolation on (tile + apron a), outer ring = seeds only
(what relaxation pass 0 sees), and count the tiles whose own pixels and first apron ring already equal
the global fixpoint.  Decides whether an apron in pass 0 would let later passes become verifications."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import oracle_lib as ol

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
TW, TH = 256, 32
img = ol.random_field(N, N, 7)
seeds = ol.find_local_minima(img)
_, G = ol.segment_arrival(img, seeds, want_keys=True)
G = np.asarray(G).reshape(N, N)
seedmask = np.zeros((N, N), bool)
seedmask[seeds[:, 0], seeds[:, 1]] = True
for a in (0, 1, 2, 4, 8, 16):
    ok_tile = ok_ring = n = 0
    wrong_px = 0
    for ty in range(0, N, TH):
        for tx in range(0, N, TW):
            y0, y1 = max(ty - a - 1, 0), min(ty + TH + a + 1, N)
            x0, x1 = max(tx - a - 1, 0), min(tx + TW + a + 1, N)
            sub = np.ascontiguousarray(img[y0:y1, x0:x1])
            sm = seedmask[y0:y1, x0:x1]
            s = np.argwhere(sm).astype(np.uint64)
            _, K = ol.segment_arrival(sub, s, want_keys=True)
            K = np.asarray(K).reshape(sub.shape)
            # tile + ring 1, clipped to the image; exclude the sub-image's own outer ring unless it is the image border
            ry0, ry1 = max(ty - 1, 0), min(ty + TH + 1, N)
            rx0, rx1 = max(tx - 1, 0), min(tx + TW + 1, N)
            kt = K[ty - y0:ty - y0 + TH, tx - x0:tx - x0 + TW]
            gt = G[ty:ty + TH, tx:tx + TW]
            t_ok = np.array_equal(kt, gt)
            wrong_px += int((kt != gt).sum())
            if a >= 1:
                kr = K[ry0 - y0:ry1 - y0, rx0 - x0:rx1 - x0]
                gr = G[ry0:ry1, rx0:rx1]
                r_ok = np.array_equal(kr, gr)
            else:
                r_ok = False
            n += 1
            ok_tile += t_ok
            ok_ring += t_ok and r_ok
    print(f"apron {a:2d}: tiles exact {ok_tile}/{n}, tile+ring1 exact {ok_ring}/{n}, wrong px {wrong_px} ({wrong_px / N / N:.2e})", flush=True)
