"""Full-size correctness of BASELINE.json's multi-GPU shapes on ONE device, and the edge-correction options.

The oracle cannot run 4096^2 or 32768^2 in seconds, so these tests check the defining equations of the result on
the device with plain torch ops (`_verify_fixpoint_on_device`: the arrival stamps are the unique fixpoint of the
flood and every label is its parent's label -- a proof of correctness that does not involve the HIP kernels), plus
equality between the different routes to the same answer (stacked batch == slice-by-slice, tiled == single domain).
"""
import importlib
import os
import sys

import numpy as np
import pytest

import __graft_entry__ as ge
import cases
import oracle_lib as ol
from test_gpu_parity import _verify_fixpoint_on_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    ge.build_hip()
    return ge.load_package()


def _engine():
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    return dev.DeviceEngine(0)


# ---- C4: 64 slices of 4096^2 over 8 GPUs = 8 stacked slices per GPU ----------------------------------------------------

def test_c4_per_gpu_shape_eight_stacked_4096_slices(pkg):
    # BASELINE config 4 on one rank: 8 x 4096^2 through ws_segment_batch_device as ONE stacked transform (134 M pixels,
    # 32-row relaxation tiles and 64-row resolve tiles straddle no slice here, 4096 % 64 == 0 -- the small batch tests
    # cover the straddling shapes).  Every slice must equal its own single-image transform, whose stamps and labels are
    # verified against the flood equations; tests/integration.rs:267,356 treat cube slices independently.
    import torch
    eng = _engine()
    S, H, W = 8, 4096, 4096
    cube = torch.empty((S, H, W), dtype=torch.uint8, device=eng.device)
    seeds, offs = [], [0]
    for k in range(S):
        cube[k] = eng.random_field(H, W, 100 + k)
        sk = eng.find_local_minima(cube[k])
        seeds.append(sk)
        offs.append(offs[-1] + int(sk.shape[0]))
    allseeds = torch.cat(seeds).contiguous()
    out = eng.segment_batch(cube, allseeds, offs)
    torch.cuda.synchronize()
    st = eng.stats()
    assert st["relax_passes"] >= 2 and st["launches_relax"] < 3 * S      # one transform for the stack, not S of them
    single = torch.empty((H, W), dtype=torch.int32, device=eng.device)
    for k in range(S):
        eng.segment(cube[k], seeds[k], out=single)
        assert bool((out[k] == single).all()), k
        if k in (0, S - 1):          # the defining equations, on the first and the last slice of the stack
            keys = eng.last_arrival()
            flooded = _verify_fixpoint_on_device(cube[k], seeds[k], out[k], keys)
            assert flooded + seeds[k].shape[0] == (H - 2) * (W - 2)
    # colours restart in every slice: seed i of slice k carries i + 1
    for k in (1, S - 1):
        s = seeds[k].to(torch.int64)
        idx = torch.arange(1, s.shape[0] + 1, device=eng.device, dtype=torch.int64)
        assert bool(((out[k].to(torch.int64) & 0xFFFFFFFF)[s[:, 0], s[:, 1]] == idx).all())


# ---- C5: one 32768^2 field --------------------------------------------------------------------------------------------

def _verify_fixpoint_banded(img, seeds, labels, keys, band=4096):
    """_verify_fixpoint_on_device on overlapping row bands (a 32768^2 plane in int64 temporaries would not fit the
    check's own working set comfortably): band rows [r0, r1) are checked inside the window [r0 - 1, r1 + 1), whose first
    and last rows only serve as neighbours."""
    import torch
    H, W = img.shape
    INF = 0xFF000000
    seedmask_rows = seeds[:, 0].to(torch.int64)
    total = 0
    for r0 in range(0, H, band):
        r1 = min(r0 + band, H)
        a, b = max(r0 - 1, 0), min(r1 + 1, H)
        k = keys[a:b].to(torch.int64) & 0xFFFFFFFF
        lab = labels[a:b].to(torch.int64) & 0xFFFFFFFF
        v = img[a:b].to(torch.int64)
        h = b - a
        big = torch.full((h + 2, W + 2), INF, dtype=torch.int64, device=img.device)
        big[1:-1, 1:-1] = k
        d, r, l, u = big[2:, 1:-1], big[1:-1, 2:], big[1:-1, :-2], big[:-2, 1:-1]
        m = torch.minimum(torch.minimum(d, r), torch.minimum(l, u))
        rows = torch.arange(a, b, device=img.device).view(-1, 1)
        cols = torch.arange(W, device=img.device).view(1, -1)
        inter = (rows >= 1) & (rows < H - 1) & (cols >= 1) & (cols < W - 1)
        base = torch.where(inter & (v <= 254), (v << 24) | 1, torch.full_like(v, INF))
        want = torch.minimum(torch.maximum(base, m + 1), torch.full_like(base, INF))
        sel = (seedmask_rows >= a) & (seedmask_rows < b)
        s = seeds[sel].to(torch.int64)
        seedmask = torch.zeros((h, W), dtype=torch.bool, device=img.device)
        seedmask[s[:, 0] - a, s[:, 1]] = True
        mine = (rows >= r0) & (rows < r1)                     # the rows this band answers for
        assert bool((k[seedmask & mine] == 0).all())
        assert bool((k[~seedmask & mine] == want[~seedmask & mine]).all()), f"stamps are not the flood fixpoint in rows {r0}..{r1}"
        lbig = torch.zeros((h + 2, W + 2), dtype=torch.int64, device=img.device)
        lbig[1:-1, 1:-1] = lab
        ld, lr, ll, lu = lbig[2:, 1:-1], lbig[1:-1, 2:], lbig[1:-1, :-2], lbig[:-2, 1:-1]
        parent = torch.where(d < k, ld, torch.where(r < k, lr, torch.where(l < k, ll, lu)))
        flooded = (~seedmask) & (k != INF) & mine
        assert bool((lab[flooded] == parent[flooded]).all()), f"a label is not its parent's label in rows {r0}..{r1}"
        assert bool((lab[(~seedmask) & (k == INF) & mine] == 0).all())
        total += int(flooded.sum())
        del big, lbig, m, want, parent, base
    idx = torch.arange(1, seeds.shape[0] + 1, device=img.device, dtype=torch.int64)
    s = seeds.to(torch.int64)
    assert bool(((labels.to(torch.int64) & 0xFFFFFFFF)[s[:, 0], s[:, 1]] == idx).all())
    return total


def test_banded_check_equals_whole_plane_check(pkg):
    import torch
    eng = _engine()
    img = eng.random_field(1500, 1024, 9)
    seeds = eng.find_local_minima(img)
    labels = eng.segment(img, seeds)
    keys = eng.last_arrival()
    torch.cuda.synchronize()
    assert _verify_fixpoint_banded(img, seeds, labels, keys, band=300) == _verify_fixpoint_on_device(img, seeds, labels, keys)
    bad = keys.clone()
    bad[700, 500] += 1                                              # the check must see a wrong stamp
    with pytest.raises(AssertionError):
        _verify_fixpoint_banded(img, seeds, labels, bad, band=300)


def test_c5_single_32768_field_on_one_device(pkg):
    # BASELINE config 5's field, whole, on one device (13 GiB of planes): every stamp and every label against the
    # flood equations.  2^30 pixels: the widest index arithmetic the engine does (31-bit pixel references).
    import torch
    eng = _engine()
    size = 32768
    img = eng.random_field(size, size, 5)
    seeds = eng.find_local_minima(img)
    assert 0.105 < seeds.shape[0] / (size * size) < 0.113
    labels = eng.segment(img, seeds)
    keys = eng.last_arrival()
    torch.cuda.synchronize()
    flooded = _verify_fixpoint_banded(img, seeds, labels, keys)
    assert flooded + seeds.shape[0] == (size - 2) * (size - 2)


# ---- edge correction: virtual ring of zeros, and the seed_shift option ------------------------------

@pytest.mark.parametrize("shape", [(62, 62), (97, 130), (300, 254), (1, 7), (5, 1)])
@pytest.mark.parametrize("engine_name", ["ENGINE_FUSED", "ENGINE_SWEEP"])
def test_edge_correction_without_a_padded_copy(pkg, shape, engine_name):
    # lib.rs:1640-1677: the plane is (h + 2) x (w + 2), the image sits inside a ring of zeros, seeds are NOT shifted.
    # The engine never builds that padded image; results must be what the oracle gets from a real padded copy.
    h, w = shape
    img = cases.field(h, w, 31)
    seeds = ol.find_local_minima(img) if min(h, w) >= 3 else np.array([[0, 0]], np.uint64)
    seeds = np.concatenate([seeds, np.array([[h + 1, w + 1], [0, w + 1]], np.uint64)])      # corners of the padded plane are legal seeds
    b = pkg.TransformBuilder.new().enable_edge_correction().set_engine(getattr(pkg, engine_name))
    got = b.build_segmenting().transform(img, seeds)
    want = ol.segment_arrival(img, seeds, edge=True)
    assert got.shape == (h + 2, w + 2) and (got == want).all()
    # strided input view (ArrayView2 with a row stride)
    wide = np.zeros((h, w + 5), np.uint8)
    wide[:, :w] = img
    assert (b.build_segmenting().transform(wide[:, :w], seeds) == want).all()


@pytest.mark.parametrize("shape", [(62, 62), (120, 201), (254, 510)])
def test_seed_shift_equals_reference_behaviour_on_moved_seeds(pkg, shape):
    # ws_options.seed_shift = 1: every seed moves by (+1, +1) onto its own pixel in the padded plane.  Not reference
    # behaviour (default 0); defined as the reference's result for the seed list with 1 added to every coordinate.
    import torch
    h, w = shape
    img = cases.field(h, w, 77)
    seeds = ol.find_local_minima(img)
    want = ol.segment_arrival(img, seeds + 1, edge=True)
    ws = pkg.TransformBuilder.new().enable_edge_correction().shift_seeds_into_padded_plane().build_segmenting()
    assert (ws.transform(img, seeds) == want).all()
    unshifted = pkg.TransformBuilder.new().enable_edge_correction().build_segmenting().transform(img, seeds)
    assert (unshifted == ol.segment_arrival(img, seeds, edge=True)).all() and not (unshifted == want).all()
    # merging history and the device entry points take the same option
    mg = pkg.TransformBuilder.new().enable_edge_correction().shift_seeds_into_padded_plane().set_max_water_lvl(90).build_merging()
    levels = {}
    ol.merge_arrival(img, seeds + 1, max_level=90, edge=True, hook=lambda lvl, mx, im, lab: levels.__setitem__(lvl, lab.copy()))
    for lvl, lab in mg.transform_history(img, seeds):
        assert (lab == levels[lvl]).all(), lvl
    eng = _engine()
    dimg = torch.from_numpy(img).to(eng.device)
    dseeds = torch.from_numpy(seeds.astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
    got = eng.segment(dimg, dseeds, edge=True, seed_shift=True).cpu().numpy().view(np.uint32)
    assert (got == want).all()
    # a stacked batch of padded slices with shifted seeds
    cube = torch.stack([dimg, dimg.flip(0).contiguous()]).contiguous()
    s1 = ol.find_local_minima(img[::-1].copy())
    both = torch.from_numpy(np.concatenate([seeds, s1]).astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
    out = eng.segment_batch(cube, both, [0, len(seeds), len(seeds) + len(s1)], edge=True, seed_shift=True).cpu().numpy().view(np.uint32)
    assert (out[0] == want).all()
    assert (out[1] == ol.segment_arrival(img[::-1].copy(), s1 + 1, edge=True)).all()
    # a seed on the last row / column cannot move into the plane's interior but stays inside the plane: accepted
    edge_seed = np.array([[h - 1, w - 1]], np.uint64)
    assert (ws.transform(img, edge_seed) == ol.segment_arrival(img, edge_seed + 1, edge=True)).all()
    with pytest.raises(IndexError):
        ws.transform(img, np.array([[h + 1, 0]], np.uint64))         # (h + 2, 1) is outside the padded plane
