"""transform_to_list (merging, all 255 levels) of an 8192^2 random field: one device whole (ws_transform_to_list_device) against
the field in R row blocks on R virtual ranks of the one device (ws_transform_to_list_tiled_device: flood on all ranks, gather,
lists on rank 0).  SIZE, RANKS from the environment."""
import ctypes, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
wsg = importlib.import_module('rustronomy_watershed_amd.group')
S = int(os.environ.get("SIZE", "8192"))
img = eng.random_field(S, S, 1)
seeds = eng.find_local_minima(img)
n_seeds = int(seeds.shape[0])
cap = int(os.environ.get("CAP", str(700_000_000 if S >= 8192 else 200_000_000)))
lakes = torch.empty((cap, 2), dtype=torch.int64, device=eng.device)
def med(fn, runs=3, warm=2):
    ts = []
    for i in range(runs + warm):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i >= warm: ts.append(dt)
    ts.sort(); return ts[len(ts) // 2] * 1e3, r
ms1, r1 = med(lambda: eng.transform_to_list(img, seeds, merging=True, lakes=lakes))
n1 = int(r1[1][-1])
print(f"{S}^2 whole: {ms1:.1f} ms, {n1} records", flush=True)
for R in [int(x) for x in os.environ.get("RANKS", "2,4").split(",")]:
    grp = wsg.Group.local(R, [0] * R)
    blocks, spans, keep = grp.make_blocks(S, lambda lo, hi, rank: img[lo:hi], seeds)
    ms, r = med(lambda: grp.transform_to_list_tiled_device(S, S, n_seeds, blocks, lakes))
    assert r[0] == n1 and (r[1] == r1[1]).all() and (r[2] == r1[2]).all()
    ms_seg, _ = med(lambda: grp.segment_tiled_device(S, S, n_seeds, blocks))
    print(f"{S}^2 in {R} row blocks on one device: {ms:.1f} ms ({r[0]} records, {r[3]} exchange rounds); the tiled segmenting transform alone {ms_seg:.1f} ms", flush=True)
    grp.close()
