"""Soak of transform_to_list over groups: random fields and random cuts (row blocks and py x px tiles, local groups on one device),
records of every level against the one-device call.  ITER, SEED from the environment."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import torch
torch.cuda.set_stream(torch.cuda.Stream(0))
eng = importlib.import_module('rustronomy_watershed_amd.device').DeviceEngine(0)
wsg = importlib.import_module('rustronomy_watershed_amd.group')
ffi = pkg._ffi
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
def sorted_levels(rec, off):
    return [rec[int(off[l]):int(off[l + 1])][np.argsort(rec[int(off[l]):int(off[l + 1]), 0])] for l in range(len(off) - 1)]
for it in range(int(os.environ.get("ITER", "40"))):
    H, W = int(rng.integers(12, 260)), int(rng.integers(12, 300))
    img = eng.random_field(H, W, 7000 + it)
    seeds = eng.find_local_minima(img)
    if rng.random() < 0.3 and seeds.shape[0] > 4:
        seeds = seeds[:: int(rng.integers(2, 6))].contiguous()
    ns = int(seeds.shape[0])
    merging = bool(rng.integers(0, 2))
    maxl = int(rng.choice([254, 254, 120, 30]))
    lakes1, off1, unc1 = eng.transform_to_list(img, seeds, merging=merging, max_level=maxl)
    want = sorted_levels(lakes1.cpu().numpy(), off1)
    cap = max(int(off1[-1]), 1) + 5
    py, px = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    if py > H or px > W: continue
    grp = wsg.Group.local(py * px)
    lakes = torch.zeros((cap, 2), dtype=torch.int64, device=eng.device)
    n = ctypes.c_size_t(0); off = np.zeros(maxl + 2, dtype=np.uint64); unc = np.zeros(maxl + 1, dtype=np.uint64)
    opt = ffi.Options(maxl)
    if px == 1 and rng.random() < 0.7:
        blocks, spans, keep = grp.make_blocks(H, lambda lo, hi, rank: img[lo:hi].contiguous(), seeds)
        rc = ffi.lib().ws_transform_to_list_tiled_device(grp._h, H, W, ns, blocks, ctypes.byref(opt), int(merging), lakes.data_ptr(), cap, ctypes.byref(n),
                                                         off.ctypes.data, unc.ctypes.data, None)
        how = f"{py} row blocks"
    else:
        blocks, spans, keep = grp.make_blocks2d(img, seeds, py, px)
        rc = ffi.lib().ws_transform_to_list_tiled2d_device(grp._h, H, W, py, px, ns, blocks, ctypes.byref(opt), int(merging), lakes.data_ptr(), cap, ctypes.byref(n),
                                                           off.ctypes.data, unc.ctypes.data, None)
        how = f"{py} x {px} tiles"
    assert rc == 0, (rc, ffi.lib().ws_group_last_error(grp._h), H, W, how)
    assert (off == np.asarray(off1, dtype=np.uint64)).all() and (unc == np.asarray(unc1, dtype=np.uint64)).all(), (it, H, W, how)
    got = sorted_levels(lakes.cpu().numpy(), off)
    for l, (a, b) in enumerate(zip(got, want)):
        assert (a == b).all(), (it, H, W, how, l)
    grp.close()
print("soak ok", flush=True)
