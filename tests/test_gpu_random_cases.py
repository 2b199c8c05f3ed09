"""Randomised sweep over small shapes, seed placements and options: the HIP engine (both engines,
host ABI) against the sweep oracle, bit-exact.  One process, ~300 transforms."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_random_small_cases_bit_exact():
    ge.build_hip()
    pkg = ge.load_package()
    rng = np.random.default_rng(2024 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))      # offset: ad-hoc wider sweeps
    for case in range(150):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 300))
        kind = case % 5
        if kind == 0:
            img = rng.integers(0, 254, (h, w), dtype=np.uint8)
        elif kind == 1:                                   # few levels: big plateaus, long rings
            img = rng.integers(0, 4, (h, w), dtype=np.uint8) * 60
        elif kind == 2:                                   # walls and floors (NEVER_FILL / ALWAYS_FILL)
            img = rng.choice(np.array([0, 255, 17, 200], dtype=np.uint8), (h, w), p=[0.3, 0.2, 0.3, 0.2])
        elif kind == 3:                                   # smooth ramp + noise
            img = ((np.add.outer(np.arange(h), np.arange(w)) * 3 + rng.integers(0, 9, (h, w))) % 254).astype(np.uint8)
        else:
            img = np.full((h, w), int(rng.integers(0, 256)), dtype=np.uint8)
        n_seeds = int(rng.integers(0, max(2, h * w // 12)))
        seeds = np.stack([rng.integers(0, h, n_seeds), rng.integers(0, w, n_seeds)], axis=1).astype(np.uint64)   # duplicates, borders allowed
        max_level = int(rng.choice([1, 3, 17, 128, 254]))
        edge = bool(rng.integers(0, 2))
        want = ol.segment(img, seeds, max_level=max_level, edge=edge)
        for engine in ((pkg.ENGINE_FUSED, pkg.ENGINE_SWEEP) if case % 3 == 0 else (pkg.ENGINE_FUSED,)):
            b = pkg.TransformBuilder.new().set_max_water_lvl(max_level).set_engine(engine)
            if edge:
                b.enable_edge_correction()
            got = b.build_segmenting().transform(img, seeds)
            assert got.shape == want.shape and (got == want).all(), (case, h, w, kind, n_seeds, max_level, edge, engine)
        if case % 10 == 0 and h * w > 0:                  # merging: final canonical partition
            mer = pkg.TransformBuilder.new().set_max_water_lvl(max_level)
            if edge:
                mer.enable_edge_correction()
            got = mer.build_merging().transform_final(img, seeds)
            assert (got == ol.merge_arrival(img, seeds, max_level=max_level, edge=edge)).all(), (case, "merge")


def test_random_medium_cases_seed_forms_and_merging():
    """Multi-tile shapes, against the arrival-form oracle: sorted-unique seed lists (side tables), the same lists
    shuffled or with duplicates (painted plane) on ONE context so that the prediction flips back and forth, and
    the merging transform's final labels at full and partial water levels (one-lake tiles and the general path)."""
    ge.build_hip()
    pkg = ge.load_package()
    rng = np.random.default_rng(77 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))
    for case in range(48):
        h = int(rng.integers(3, 420))
        w = int(rng.integers(1, 200)) * 4 if case % 4 else int(rng.integers(3, 800))      # mostly W % 4 == 0 (table form allowed)
        kind = case % 3
        if kind == 0:
            img = rng.integers(0, 254, (h, w), dtype=np.uint8)
        elif kind == 1:
            img = (rng.integers(0, 6, (h, w), dtype=np.uint8) * 40).astype(np.uint8)
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            img = ((np.sin(yy / 17.0) + np.cos(xx / 23.0) + 2.0) * 60 + rng.integers(0, 5, (h, w))).astype(np.uint8)
        n_seeds = int(rng.integers(1, max(2, h * w // 40)))
        flat = np.sort(rng.choice(h * w, size=min(n_seeds, h * w), replace=False))
        seeds = np.stack([flat // w, flat % w], axis=1).astype(np.uint64)                    # strictly increasing, row-major
        form = case % 3
        if form == 1:
            seeds = seeds[rng.permutation(len(seeds))]
        elif form == 2:
            seeds = np.concatenate([seeds, seeds[:: 3]])                                     # sorted part + duplicates at the end
        max_level = int(rng.choice([254, 254, 100, 31]))
        b = pkg.TransformBuilder.new().set_max_water_lvl(max_level).set_engine(pkg.ENGINE_FUSED)
        got = b.build_segmenting().transform(img, seeds)
        want = ol.segment_arrival(img, seeds, max_level=max_level)
        assert got.shape == want.shape and (got == want).all(), ("segment", case, h, w, kind, form, max_level)
        if case % 2 == 0:
            got = pkg.TransformBuilder.new().set_max_water_lvl(max_level).build_merging().transform_final(img, seeds)
            assert (got == ol.merge_arrival(img, seeds, max_level=max_level)).all(), ("merge", case, h, w, kind, form, max_level)


def test_random_mazes_long_range_segmenting():
    """Walls (NEVER_FILL) with random corridors, plateaus and ramps at the scale of the relaxation tiles (256 x 32):
    floods that wind through many tiles, hundreds of rings per level, dozens to hundreds of passes -- the alternating
    grids, the quadrant flags, the round cap of pass 0 and the row / column scans of the late passes."""
    ge.build_hip()
    pkg = ge.load_package()
    rng = np.random.default_rng(1234 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))
    most_passes = 0
    for case in range(16):
        h = int(rng.integers(40, 300))
        w = int(rng.integers(60, 330)) * 4 if case % 3 else int(rng.integers(200, 1300))
        img = np.full((h, w), 255, np.uint8)
        # corridors: random horizontal and vertical slits of random value through the wall
        for _ in range(int(rng.integers(6, 40))):
            v = int(rng.choice([3, 3, 3, 40, 120, 200]))
            if rng.integers(0, 2):
                y = int(rng.integers(1, h - 1)); x0, x1 = sorted(rng.integers(1, w - 1, 2))
                img[y, x0:x1 + 1] = np.minimum(img[y, x0:x1 + 1], v)
            else:
                x = int(rng.integers(1, w - 1)); y0, y1 = sorted(rng.integers(1, h - 1, 2))
                img[y0:y1 + 1, x] = np.minimum(img[y0:y1 + 1, x], v)
        # a few open rooms (plateaus / ramps)
        for _ in range(int(rng.integers(1, 5))):
            y0, x0 = int(rng.integers(1, h - 8)), int(rng.integers(1, w - 8))
            hh, ww = int(rng.integers(4, 60)), int(rng.integers(4, 300))
            room = img[y0:y0 + hh, x0:x0 + ww]
            ramp = ((np.arange(room.shape[1])[None, :] * int(rng.integers(0, 3))) // 4 + int(rng.integers(0, 200))) % 254
            room[:] = np.minimum(room, np.broadcast_to(ramp, room.shape).astype(np.uint8))
        open_px = np.argwhere(img < 255)
        n_seeds = int(rng.integers(1, 6))
        seeds = open_px[rng.choice(len(open_px), size=min(n_seeds, len(open_px)), replace=False)].astype(np.uint64)
        ws = pkg.TransformBuilder.new().set_engine(pkg.ENGINE_FUSED).build_segmenting()
        # every second maze through the passes themselves (mode 0); the others as the default takes them -- with so few seeds
        # the tile queue in flood order
        assert pkg._ffi.lib().ws_ctx_set_persistent_pass(ws._ctx().handle, 0 if case % 2 == 0 else 3) == 0
        got = ws.transform(img, seeds)
        want = ol.segment_arrival(img, seeds)
        assert got.shape == want.shape and (got == want).all(), ("maze", case, h, w, n_seeds)
        assert pkg._ffi.lib().ws_ctx_set_persistent_pass(ws._ctx().handle, 3) == 0
        if case % 2 == 0:
            most_passes = max(most_passes, ws._ctx().stats()["relax_passes"])
    assert most_passes >= 12, most_passes            # the late-pass kernel variants did run


def test_random_batches_of_slices_bit_exact():
    """ws_segment_batch_device on random stacks: shapes that take the stacked form (w' % 4 == 0, h' * w' % 128 == 0) and
    shapes that do not, sorted and unsorted lists, empty slices, border seeds, edge correction, low water levels."""
    import importlib
    import torch
    ge.build_hip()
    ge.load_package()
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    rng = np.random.default_rng(77 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))
    for case in range(24):
        edge = case % 4 == 3
        s = int(rng.integers(2, 7))
        if case % 3 == 2:                                  # anything goes: mostly the slice-by-slice loop
            h, w = int(rng.integers(3, 70)), int(rng.integers(3, 150))
        else:                                              # padded plane with w' % 4 == 0 and h' * w' % 128 == 0
            wp = int(rng.integers(2, 48)) * 4
            need = 128 // np.gcd(128, wp)
            hp = int(rng.integers(1, max(2, 80 // need))) * need
            h, w = (hp - 2, wp - 2) if edge else (hp, wp)
            if h < 1 or w < 1:
                continue
        himgs, hseeds = [], []
        for k in range(s):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                a = rng.integers(0, 254, (h, w), dtype=np.uint8)
            elif kind == 1:
                a = (rng.integers(0, 4, (h, w), dtype=np.uint8) * 60).astype(np.uint8)
            else:
                a = rng.choice(np.array([0, 255, 17, 200], dtype=np.uint8), (h, w), p=[0.3, 0.2, 0.3, 0.2])
            n = int(rng.integers(0, max(2, h * w // 10))) if rng.integers(0, 6) else 0
            flat = np.sort(rng.choice(h * w, size=min(n, h * w), replace=False))
            sd = np.stack([flat // w, flat % w], axis=1).astype(np.int64).reshape(-1, 2)
            if case % 5 == 4 and len(sd) > 2 and k == s // 2:
                sd = sd[::-1].copy()                        # one unsorted list: the whole batch goes slice by slice
            himgs.append(a)
            hseeds.append(sd)
        max_level = int(rng.choice([254, 254, 90, 3]))
        offs = np.concatenate([[0], np.cumsum([len(x) for x in hseeds])])
        cube = torch.from_numpy(np.stack(himgs)).to(eng.device)
        allseeds = torch.from_numpy(np.concatenate(hseeds).reshape(-1, 2)).to(torch.int32).to(eng.device).contiguous()
        if os.environ.get("WS_TEST_TRACE"): print("batch case", case, s, h, w, edge, max_level, [len(x) for x in hseeds], flush=True)
        got = eng.segment_batch(cube, allseeds, offs, max_level=max_level, edge=edge).cpu().numpy().view(np.uint32)
        for k in range(s):
            want = ol.segment_arrival(himgs[k], hseeds[k].astype(np.uint64), max_level=max_level, edge=edge)
            assert got[k].shape == want.shape and (got[k] == want).all(), (case, k, s, h, w, edge, max_level)


@pytest.mark.parametrize("seam", [False, True])
def test_random_call_sequences_on_one_context_bit_exact(seam):
    """(seam: planes wide enough for the seam repair, its size threshold lowered to 1 pixel and now and then raised again
    between calls, which retires the captured graphs.)
    One context on a real stream (graph capture allowed), a random sequence of segmenting / merging / batch calls
    over a few fixed sets of device buffers whose CONTENTS change: graphs get captured, replayed, retired by other
    shapes, fed unsorted lists (the replayed tables are then wrong and the transform repeats itself), corridors that
    need more passes than a graph holds."""
    import importlib
    import torch
    ge.build_hip()
    ge.load_package()
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    rng = np.random.default_rng(4242 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))
    with torch.cuda.stream(torch.cuda.Stream()):
        eng = dev.DeviceEngine(0)
        sets = []
        if seam:
            eng.ctx.set_seam_repair_min_pixels(1)
        for (h, w, n) in (((64, 520, 900), (96, 772, 60), (40, 300, 7)) if seam else ((64, 128, 300), (96, 256, 40), (40, 64, 7))):
            sets.append({"h": h, "w": w, "n": n,
                         "img": torch.empty((h, w), dtype=torch.uint8, device=eng.device),
                         "seeds": torch.empty((n, 2), dtype=torch.int32, device=eng.device),
                         "out": torch.empty((h, w), dtype=torch.int32, device=eng.device)})
        replays = 0
        for call in range(60):
            st = sets[int(rng.choice([0, 0, 0, 1, 1, 2]))]
            h, w, n = st["h"], st["w"], st["n"]
            if seam and rng.integers(0, 10) == 0:
                eng.ctx.set_seam_repair_min_pixels(int(rng.choice([0, 1])))      # 0: the default threshold (these planes: no repair)
            kind = int(rng.integers(0, 4))
            if kind == 0:
                img = rng.integers(0, 254, (h, w), dtype=np.uint8)
            elif kind == 1:
                img = (rng.integers(0, 4, (h, w), dtype=np.uint8) * 60).astype(np.uint8)
            elif kind == 2:                                  # a corridor: many passes
                img = np.full((h, w), 255, np.uint8)
                for k, y in enumerate(range(2, h - 2, 4)):
                    img[y, 2:w - 2] = 9
                    img[y:y + 5, (w - 3) if k % 2 == 0 else 2] = 9
            else:
                img = rng.choice(np.array([0, 255, 17, 200], dtype=np.uint8), (h, w), p=[0.3, 0.2, 0.3, 0.2])
            flat = np.sort(rng.choice(h * w, size=n, replace=False))
            if rng.integers(0, 8) == 0:
                flat = flat[rng.permutation(n)]              # the list is not sorted this time
            seeds = np.stack([flat // w, flat % w], axis=1).astype(np.int64)
            st["img"].copy_(torch.from_numpy(img))
            st["seeds"].copy_(torch.from_numpy(seeds).to(torch.int32))
            if rng.integers(0, 4) == 0:
                got = eng.merge(st["img"], st["seeds"], out=st["out"]).cpu().numpy().view(np.uint32)
                want = ol.merge_arrival(img, seeds.astype(np.uint64))
            else:
                got = eng.segment(st["img"], st["seeds"], out=st["out"]).cpu().numpy().view(np.uint32)
                want = ol.segment_arrival(img, seeds.astype(np.uint64))
            replays += eng.stats()["graph_launches"]
            assert (got == want).all(), (call, h, w, kind)
        assert replays >= 5, replays


def test_random_transform_to_list_bit_exact():
    """transform_to_list (merging and segmenting) on random small fields against the oracle's per-level lake sizes:
    plateaus, walls, sorted and shuffled seed lists, edge correction, low and high water levels."""
    ge.build_hip()
    pkg = ge.load_package()
    rng = np.random.default_rng(909 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))
    for case in range(14):
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 60))
        kind = case % 3
        if kind == 0:
            img = rng.integers(0, 254, (h, w), dtype=np.uint8)
        elif kind == 1:
            img = (rng.integers(0, 4, (h, w), dtype=np.uint8) * 60).astype(np.uint8)
        else:
            img = rng.choice(np.array([0, 255, 17, 200], dtype=np.uint8), (h, w), p=[0.3, 0.2, 0.3, 0.2])
        n = int(rng.integers(1, max(2, h * w // 8)))
        flat = np.sort(rng.choice(h * w, size=min(n, h * w), replace=False))
        if case % 4 == 1:
            flat = flat[rng.permutation(len(flat))]
        seeds = np.stack([flat // w, flat % w], axis=1).astype(np.uint64)
        max_level = int(rng.choice([254, 120, 7]))
        edge = bool(rng.integers(0, 2))
        merging = case % 2 == 0
        b = pkg.TransformBuilder.new().set_max_water_lvl(max_level)
        if edge:
            b.enable_edge_correction()
        want = []
        if merging:
            ol.merge(img, seeds, max_level=max_level, edge=edge,
                     hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(ol.canonicalise(c, seeds)[0])))
            got = b.build_merging().transform_to_list(img, seeds)
        else:
            ol.segment(img, seeds, max_level=max_level, edge=edge, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(c)))
            got = b.build_segmenting().transform_to_list(img, seeds)
        assert [l for l, _ in got] == list(range(max_level + 1)), (case, merging)
        for (lvl, hist), wnt in zip(got, want):
            assert hist.shape == wnt.shape and (hist == wnt).all(), (case, lvl, h, w, merging, edge, max_level)


def test_random_smooth_maps_across_the_three_schedules():
    """Smooth fields of random shape and correlation with seed lists thinned at random: the default's three regimes by seeds
    per tile -- the tile queue in flood order, the passes on the early schedule, the passes as they were -- each also
    forced (modes 2, 4, 0), on ONE context so that captured graphs of one mode meet transforms of another; segmenting and,
    now and then, the merging transform's final labels, against the arrival-form oracle."""
    import cases
    ge.build_hip()
    pkg = ge.load_package()
    rng = np.random.default_rng(4242 + int(os.environ.get("WS_TEST_SEED_OFFSET", "0")))
    ws = pkg.TransformBuilder.new().set_engine(pkg.ENGINE_FUSED).build_segmenting()
    set_mode = pkg._ffi.lib().ws_ctx_set_persistent_pass
    picked = {0: 0, 2: 0, 4: 0}
    for case in range(24):
        h = int(rng.integers(300, 1400))
        w = int(rng.integers(80, 400)) * 4 if case % 5 else int(rng.integers(300, 1500))      # now and then a width that is no multiple of 4
        img = cases.smooth_field(h, w, int(rng.integers(1, 1000)), octaves=int(rng.integers(4, 8)))
        seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
        keep = int(rng.choice([1, 3, 12, 60, 400]))
        seeds = seeds[:: keep] if len(seeds) > keep else seeds
        if len(seeds) == 0:
            continue
        max_level = int(rng.choice([254, 254, 160, 70]))
        want = ol.segment_arrival(img, seeds, max_level=max_level)
        b = pkg.TransformBuilder.new().set_engine(pkg.ENGINE_FUSED).set_max_water_lvl(max_level)
        ws = b.build_segmenting()
        for mode in (3, int(rng.choice([0, 2, 4]))):
            assert set_mode(ws._ctx().handle, mode) == 0
            got = ws.transform(img, seeds)
            assert (got == want).all(), (case, h, w, len(seeds), max_level, mode, int((got != want).sum()))
        tiles = ((w + 127) // 128 + 1) * ((h + 63) // 64 + 1)
        picked[2 if len(seeds) * 2 <= tiles else (4 if len(seeds) <= 32 * tiles else 0)] += 1
        assert set_mode(ws._ctx().handle, 3) == 0
        if case % 6 == 0:
            mg = pkg.TransformBuilder.new().set_max_water_lvl(max_level).build_merging()
            assert (mg.transform_final(img, seeds) == ol.merge_arrival(img, seeds, max_level=max_level)).all(), (case, "merge")
    assert picked[2] >= 3 and picked[4] >= 3, picked      # (roughly: relax_tiles is a little larger than this count)
