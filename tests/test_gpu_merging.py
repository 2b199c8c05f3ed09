"""Parity of the merging transform and of transform_to_list with the CPU oracle (-m gpu).

The reference's merged-lake ids are arbitrary (sort/dedup order, lib.rs:440-443, 508-541), so
planes are compared after canonicalisation (smallest seed colour in the lake) and lake-size
lists as sorted multisets."""
import numpy as np
import pytest

import __graft_entry__ as ge
import cases
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    ge.build_hip()
    return ge.load_package()


def _merging(pkg, max_level=254, edge=False, hook=None):
    b = pkg.TransformBuilder.new().set_max_water_lvl(max_level)
    if edge:
        b.enable_edge_correction()
    if hook is not None:
        b.set_wlvl_hook(hook)
    return b.build_merging()


def _oracle_levels(img, seeds, **kw):
    snaps = []
    ol.merge(img, seeds, hook=lambda l, m, i, c: snaps.append(ol.canonicalise(c, seeds)[0]), **kw)
    return snaps


@pytest.mark.parametrize("shape,seed,edge", [((24, 24), 1, False), ((50, 70), 2, False), ((96, 96), 3, True),
                                              ((130, 67), 4, False), ((200, 300), 5, True)])
def test_merge_history_matches_oracle_every_level(pkg, shape, seed, edge):
    img = cases.field(*shape, seed)
    seeds = ol.find_local_minima(img)
    want = _oracle_levels(img, seeds, edge=edge)
    hist = _merging(pkg, edge=edge).transform_history(img, seeds)
    assert [l for l, _ in hist] == list(range(255))
    for (lvl, got), w in zip(hist, want):
        assert got.shape == w.shape
        assert (got == w).all(), lvl


def test_merge_partition_matches_randomised_faithful_reference_shape(pkg):
    # the oracle run with the reference's own random tie-break and its order-dependent
    # representatives must give the same partition and the same sorted lake sizes
    img = cases.field(80, 90, 7)
    seeds = ol.find_local_minima(img)
    snaps = []
    ol.merge(img, seeds, tie=ol.TIE_RANDOM, rng_seed=11, mode=ol.MAP_FAITHFUL, hook=lambda l, m, i, c: snaps.append(c.copy()))
    hist = _merging(pkg).transform_history(img, seeds)
    for lvl in range(0, 255, 5):
        got = hist[lvl][1]
        assert (ol.canonicalise(snaps[lvl], seeds)[0] == got).all(), lvl
        assert sorted(ol.find_lake_sizes(snaps[lvl])[1:].tolist()) == sorted(ol.find_lake_sizes(got)[1:].tolist())


def test_merge_adversarial_cases(pkg):
    for name, img, seeds in cases.adversarial_cases():
        seeds = cases.seeds_or_maxima(img, seeds)
        if img.size == 0:
            continue
        for edge in (False, True):
            want = _oracle_levels(img, seeds, edge=edge)
            hist = _merging(pkg, edge=edge).transform_history(img, seeds)
            for lvl in (0, 1, 7, 9, 100, 254):
                assert (hist[lvl][1] == want[lvl]).all(), (name, edge, lvl)


@pytest.mark.parametrize("maxlvl", [1, 60, 254])
def test_merge_to_list_matches_oracle_lake_sizes(pkg, maxlvl):
    img = cases.field(72, 88, 9)
    seeds = ol.find_local_minima(img)
    want = []
    ol.merge(img, seeds, max_level=maxlvl, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(ol.canonicalise(c, seeds)[0])))
    got = _merging(pkg, max_level=maxlvl).transform_to_list(img, seeds)
    assert [l for l, _ in got] == list(range(maxlvl + 1))
    for (lvl, hist), w in zip(got, want):
        assert hist.shape == w.shape == (72 * 88 + 1,)              # lib.rs:630: pixels + 1
        assert (hist == w).all(), lvl


def test_segmenting_to_list_matches_oracle(pkg):
    img = cases.field(64, 80, 10)
    seeds = ol.find_local_minima(img)
    want = []
    ol.segment(img, seeds, max_level=90, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(c)))
    ws = pkg.TransformBuilder.new().set_max_water_lvl(90).build_segmenting()
    got = ws.transform_to_list(img, seeds)
    for (lvl, hist), w in zip(got, want):
        assert (hist == w).all(), lvl


def test_merge_to_list_sparse_core_bench_shape(pkg):
    # tests/core_bench.rs:27-61: merging transform_to_list on a 1024x1024 Uniform(0,254) field
    img = cases.field(1024, 1024, 1)
    ws = _merging(pkg)
    seeds = ws.find_local_minima(img)
    res = ws.transform_to_list_sparse(img, seeds)
    assert len(res) == 255
    interior = 1022 * 1022
    # conservation: areas + uncoloured = pixels, at every level; lakes only ever merge
    prev_lakes = None
    for lvl, unc, colours, areas in res:
        assert int(areas.sum()) + unc == 1024 * 1024
        assert len(set(colours.tolist())) == len(colours)
        if prev_lakes is not None:
            assert len(colours) <= prev_lakes
        prev_lakes = len(colours)
    assert res[-1][1] == 1024 * 1024 - interior                     # only the border stays uncoloured
    # spot-check three levels against the arrival-form oracle (canonical partition per level)
    want = {}
    ol.merge_arrival(img, seeds, hook=lambda l, m, i, c: want.__setitem__(l, ol.find_lake_sizes(c)) if l in (3, 120, 254) else None)
    for lvl in (3, 120, 254):
        _, unc, colours, areas = res[lvl]
        dense = np.zeros(1024 * 1024 + 1, np.uint64)
        dense[colours.astype(np.int64)] = areas
        dense[0] = unc
        assert (dense == want[lvl]).all(), lvl


@pytest.mark.parametrize("shape", [(600, 700), (1500, 1400)])      # the second: the host plane is widened from u32 chunks (ws_hostcopy.hip)
def test_merge_final_labels_device_and_host(pkg, shape):
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    img = eng.random_field(shape[0], shape[1], 4)
    seeds = eng.find_local_minima(img)
    got = eng.merge(img, seeds, max_level=100)
    torch.cuda.synchronize()
    himg = img.cpu().numpy()
    hseeds = seeds.cpu().numpy().astype(np.uint64)
    want = ol.merge_arrival(himg, hseeds, max_level=100)
    assert (got.cpu().numpy().view(np.uint32) == want).all()
    host = _merging(pkg, max_level=100).transform_final(himg, hseeds)
    assert (host == want).all()


def test_merge_config3_8192_properties(pkg):
    # BASELINE config C3: 8192x8192 merging.  The oracle cannot run this size; check the defining
    # properties on the device: the merged plane is constant on every segmenting lake, two adjacent
    # coloured pixels (one of them interior) always share a merged id, and ids are the smallest
    # colour of their class.
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    size = 8192
    img = eng.random_field(size, size, 1)
    seeds = eng.find_local_minima(img)
    seg = eng.segment(img, seeds)
    mer = eng.merge(img, seeds)
    torch.cuda.synchronize()
    seg64 = seg.to(torch.int64) & 0xFFFFFFFF
    mer64 = mer.to(torch.int64) & 0xFFFFFFFF
    assert bool(((seg64 == 0) == (mer64 == 0)).all())
    assert bool((mer64 <= seg64).all())                       # canonical id = smallest colour of the class
    # constant on segmenting lakes: map colour -> merged id must be a function
    S = seeds.shape[0]
    lo = torch.full((S + 1,), 1 << 40, dtype=torch.int64, device=img.device)
    hi = torch.zeros((S + 1,), dtype=torch.int64, device=img.device)
    lo.scatter_reduce_(0, seg64.flatten(), mer64.flatten(), reduce="amin")
    hi.scatter_reduce_(0, seg64.flatten(), mer64.flatten(), reduce="amax")
    used = hi > 0
    assert bool((lo[used] == hi[used]).all())
    # adjacency closure (interior-centre rule of find_merge, lib.rs:411-434)
    inter = torch.zeros((size, size), dtype=torch.bool, device=img.device)
    inter[1:-1, 1:-1] = True
    a, b = mer64[:, :-1], mer64[:, 1:]
    ok = (a == 0) | (b == 0) | (a == b) | ~(inter[:, :-1] | inter[:, 1:])
    assert bool(ok.all())
    a, b = mer64[:-1, :], mer64[1:, :]
    ok = (a == 0) | (b == 0) | (a == b) | ~(inter[:-1, :] | inter[1:, :])
    assert bool(ok.all())
    # on a uniform random field the whole interior ends up as one lake with id 1... unless a border seed
    ids = torch.unique(mer64[1:-1, 1:-1])
    assert ids.numel() == 1


@pytest.mark.parametrize("shape,edge", [((600, 700), False), ((512, 512), False), ((130, 70), False), ((64, 64), False),
                                        ((65, 129), False), ((257, 193), True), ((66, 66), False), ((2, 300), False)])
def test_merge_final_labels_full_level_border_and_corner_seeds(pkg, shape, edge):
    # final level of a fully flooded field: nearly every tile is one lake (the one-lake tile path), seeds on
    # the image border join through their inward neighbour, seeds on the four corners never join anything
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    h, w = shape
    himg = cases.field(h, w, 31)
    base = ol.find_local_minima(himg)
    ph, pw = (h + 2, w + 2) if edge else (h, w)
    extra = [(0, 0), (0, pw - 1), (ph - 1, 0), (ph - 1, pw - 1), (0, pw // 2), (ph - 1, pw // 3), (ph // 2, 0), (ph // 3, pw - 1),
             (0, 1), (1, 0)]
    extra = [(r, c) for r, c in extra if r < (ph if edge else h) and c < (pw if edge else w)]
    hseeds = np.concatenate([base.reshape(-1, 2), np.array(extra, np.uint64).reshape(-1, 2)]).astype(np.uint64)
    img = torch.from_numpy(himg).to(eng.device)
    seeds = torch.from_numpy(hseeds.astype(np.int64)).to(eng.device).to(torch.int32)
    got = eng.merge(img, seeds.contiguous(), edge=edge)
    torch.cuda.synchronize()
    want = ol.merge_arrival(himg, hseeds, edge=edge)
    assert (got.cpu().numpy().view(np.uint32) == want).all()


def test_merge_final_colour_enters_one_lake_tile_through_a_general_tile(pkg):
    # tiles are 64 wide: tile 0 and tile 2 flood completely (one-lake tiles), tile 1 is a wall with a one-pixel
    # corridor (a general tile).  Colour 2 (seeded in tile 0) reaches tile 2 only through the corridor, as a
    # same-colour crossing pair, and meets colour 1 (seeded in tile 2) inside tile 2: one lake, id 1.
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    for h, w, wall in ((64, 192, (64, 128)), (60, 256, (64, 192)), (64, 320, (128, 192))):      # one tile row: no detour through a tile below
        himg = np.full((h, w), 7, np.uint8)
        himg[:, wall[0]:wall[1]] = 255
        himg[30, wall[0]:wall[1]] = 7                                   # the corridor
        himg[6:15, wall[1] + 16:wall[1] + 25] = 100                      # colour 1 sits in a pocket that floods late:
        hseeds = np.array([[10, wall[1] + 20], [30, 10]], np.uint64)     # colour 2 (seeded before the wall) fills tile 2 first
        img = torch.from_numpy(himg).to(eng.device)
        seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
        got = eng.merge(img, seeds).cpu().numpy().view(np.uint32)
        want = ol.merge_arrival(himg, hseeds)
        assert (got == want).all(), (h, w)
        assert set(np.unique(want)) == {0, 1}


def test_merge_final_random_mazes_of_open_and_walled_tiles(pkg):
    # mixtures of one-lake tiles (fully flooded 64x64 blocks) and general tiles (walls with corridors, pockets that
    # flood late), few seeds in random order: every way a colour can reach a tile without its seed
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    rng = np.random.default_rng(int(__import__("os").environ.get("WS_TEST_SEED_OFFSET", "0")) + 4242)
    for case in range(40):
        ty, tx = int(rng.integers(1, 5)), int(rng.integers(1, 6))
        h, w = ty * 64 - int(rng.integers(0, 5)), tx * 64 - int(rng.integers(0, 5))
        himg = np.full((h, w), 9, np.uint8)
        for by in range(ty):
            for bx in range(tx):
                kind = rng.integers(0, 4)
                y0, x0, y1, x1 = by * 64, bx * 64, min(by * 64 + 64, h), min(bx * 64 + 64, w)
                if kind == 0:                                             # walled tile with a few corridors
                    himg[y0:y1, x0:x1] = 255
                    for _ in range(int(rng.integers(1, 4))):
                        if rng.integers(0, 2):
                            himg[int(rng.integers(y0, y1)), x0:x1] = 9
                        else:
                            himg[y0:y1, int(rng.integers(x0, x1))] = 9
                elif kind == 1:                                           # open tile with late-flooding pockets
                    for _ in range(int(rng.integers(1, 4))):
                        py, px = int(rng.integers(y0, y1)), int(rng.integers(x0, x1))
                        himg[py:py + 9, px:px + 9] = int(rng.integers(60, 250))
        n_seeds = int(rng.integers(1, 9))
        hseeds = np.stack([rng.integers(0, h, n_seeds), rng.integers(0, w, n_seeds)], axis=1).astype(np.uint64)
        img = torch.from_numpy(himg).to(eng.device)
        seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
        got = eng.merge(img, seeds).cpu().numpy().view(np.uint32)
        want = ol.merge_arrival(himg, hseeds)
        assert (got == want).all(), (case, h, w, n_seeds)


def test_to_list_call_sequences_on_one_context_replay_the_right_graphs(pkg):
    # the level loop of transform_to_list / transform_final is replayed as hipGraphs when a call repeats the previous
    # one's shape; a call of ANOTHER shape (levels, seeds, pixels, capacity) must never replay them.  One context,
    # shapes interleaved and repeated, every result against the oracle.
    cases_ = [((72, 88), 9, 1), ((72, 88), 9, 60), ((128, 96), 3, 254), ((72, 88), 9, 60), ((40, 52), 5, 17), ((72, 88), 9, 254)]
    for shape, seed, maxlvl in cases_:
        img = cases.field(*shape, seed)
        seeds = ol.find_local_minima(img)
        want = []
        final = ol.merge_arrival(img, seeds, max_level=maxlvl, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(c)))
        ws = _merging(pkg, max_level=maxlvl)
        for rep in range(3):                      # 2nd call captures, 3rd replays
            got = ws.transform_to_list_sparse(img, seeds)
            assert len(got) == maxlvl + 1
            for (lvl, unc, cols, areas), w in zip(got, want):
                nz = np.nonzero(w[1:])[0] + 1
                assert unc == w[0] and (np.sort(cols) == nz).all() and (areas[np.argsort(cols)] == w[nz]).all(), (shape, maxlvl, rep, lvl)
            assert (ws.transform_final(img, seeds) == final).all(), (shape, maxlvl, rep)
            seg = pkg.TransformBuilder.new().set_max_water_lvl(maxlvl).build_segmenting().transform_to_list_sparse(img, seeds)
            assert sum(int(a.sum()) for _, _, _, a in seg[-1:]) + seg[-1][1] == img.size


def test_transform_to_list_device_resident(pkg):
    # ws_transform_to_list_device: image, seeds and lake records stay in HBM; only offsets and uncoloured counts come back
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    for shape, seed, maxlvl, merging, edge in (((96, 132), 4, 254, True, False), ((70, 64), 5, 90, True, True), ((128, 100), 6, 254, False, False)):
        himg = cases.field(*shape, seed)
        hseeds = ol.find_local_minima(himg)
        want = []
        if merging:
            ol.merge_arrival(himg, hseeds, max_level=maxlvl, edge=edge, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(c)))
        else:
            ol.segment(himg, hseeds, max_level=maxlvl, edge=edge, hook=lambda l, m, i, c: want.append(ol.find_lake_sizes(c)))
        img = torch.from_numpy(himg).to(eng.device)
        seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
        buf = None
        for rep in range(3):
            lakes, offsets, unc = eng.transform_to_list(img, seeds, merging=merging, max_level=maxlvl, edge=edge, lakes=buf)
            buf = lakes if rep == 0 else buf
            rec = lakes.cpu().numpy()
            assert len(offsets) == maxlvl + 2 and int(offsets[-1]) == len(rec)
            for lvl, w in enumerate(want):
                r = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
                nz = np.nonzero(w[1:])[0] + 1
                assert unc[lvl] == w[0] and (np.sort(r[:, 0]) == nz).all() and (r[np.argsort(r[:, 0]), 1] == w[nz]).all(), (shape, lvl, rep)


def test_host_lists_of_a_plane_whose_records_cross_the_bus_as_u32(pkg):
    # ws_transform_to_list with host buffers at 2048^2 (39 M records, 2.4 M a group of levels): groups of two million words and
    # more are narrowed to u32 on the device and widened into the caller's ws_lake records by the host's threads (ws_lists.hip,
    # ws_hostcopy.hip).  Against the device-resident call's records, level by level, and the oracle's lake sizes at five levels.
    import ctypes
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    himg = ol.random_field(2048, 2048, 21)
    hseeds = np.ascontiguousarray(np.asarray(ol.find_local_minima(himg), dtype=np.uint64).reshape(-1, 2))
    img = torch.from_numpy(himg).to(eng.device)
    seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
    d_lakes, d_off, d_unc = eng.transform_to_list(img, seeds, merging=True)
    d_rec = d_lakes.cpu().numpy().astype(np.uint64)
    ws = _merging(pkg)
    c, L = ws._ctx(), pkg._ffi.lib()
    cap = int(d_off[-1]) + 10
    rec = np.zeros((cap, 2), dtype=np.uint64)
    n_lakes = ctypes.c_size_t(0)
    offsets, unc = np.zeros(256, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
    for threads in (4, 0):      # widened by host threads; one 16-byte copy per group, as before
        assert L.ws_ctx_set_host_threads(c.handle, threads) == 0
        rec[:] = 0
        assert L.ws_transform_to_list(c.handle, 1, himg.ctypes.data, 2048, 2048, 2048, hseeds.ctypes.data, len(hseeds), ctypes.byref(ws._opt),
                                      rec.ctypes.data, cap, ctypes.byref(n_lakes), offsets.ctypes.data, unc.ctypes.data) == 0
        assert n_lakes.value == int(d_off[-1]) and (offsets == np.asarray(d_off, dtype=np.uint64)).all() and (unc == np.asarray(d_unc, dtype=np.uint64)).all()
        for lvl in range(255):
            a = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
            b = d_rec[int(offsets[lvl]):int(offsets[lvl + 1])]
            assert (a[np.argsort(a[:, 0])] == b[np.argsort(b[:, 0])]).all(), (threads, lvl)
    want = {}
    levels = (0, 40, 128, 200, 254)
    ol.merge_arrival(himg, hseeds, hook=lambda l, m, i, c: want.__setitem__(l, ol.find_lake_sizes(c)) if l in levels else None)
    for lvl in levels:
        r = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
        nz = np.nonzero(want[lvl][1:])[0] + 1
        assert unc[lvl] == want[lvl][0] and (np.sort(r[:, 0]) == nz).all() and (r[np.argsort(r[:, 0]), 1] == want[lvl][nz]).all(), lvl
