"""Parity of the HIP engine with the CPU oracle, through the C ABI (run with -m gpu on MI355X).

Bar: bit-exact labels (integer work).  Sizes the sweep oracle finishes in seconds are compared
directly; larger fields against the arrival-form oracle (itself proven equal to the sweep form
in test_oracle_golden.py); BASELINE.json's full sizes through size-independent properties."""
import ctypes

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


def _seg(pkg, engine=None, max_level=254, edge=False, hook=None):
    b = pkg.TransformBuilder.new().set_max_water_lvl(max_level)
    if edge:
        b.enable_edge_correction()
    if engine is not None:
        b.set_engine(engine)
    if hook is not None:
        b.set_wlvl_hook(hook)
    return b.build_segmenting()


# ---- find_local_minima (lib.rs:1178-1197) -------------------------------------------------

@pytest.mark.parametrize("shape,seed", [((16, 16), 1), ((64, 64), 2), ((47, 93), 3), ((512, 512), 1), ((300, 1030), 4),
                                         ((1025, 2050), 5), ((3, 3), 6), ((2, 50), 7), ((50, 2), 8)])
def test_find_local_minima_matches_oracle(pkg, shape, seed):
    img = cases.field(*shape, seed)
    got = _seg(pkg).find_local_minima(img)
    want = ol.find_local_minima(img)
    assert got.shape == want.shape and (got == want).all()


def test_find_local_minima_plateaus_and_strided_input(pkg):
    ws = _seg(pkg)
    assert len(ws.find_local_minima(np.full((40, 40), 7, np.uint8))) == 0       # no strict maximum on a plateau
    big = cases.field(200, 300, 9)
    view = big[10:150, 20:220]                                                   # row stride != width
    assert (ws.find_local_minima(view) == ol.find_local_minima(np.ascontiguousarray(view))).all()


# ---- segmenting transform -----------------------------------------------------------------

@pytest.mark.parametrize("shape,seed", [((16, 16), 1), ((64, 64), 2), ((47, 93), 3), ((256, 256), 4), ((512, 512), 1),
                                         ((300, 210), 5), ((1024, 1024), 6)])
def test_segment_fused_bit_exact_vs_sweep_oracle(pkg, shape, seed):
    img = cases.field(*shape, seed)
    seeds = ol.find_local_minima(img)
    got = _seg(pkg, pkg.ENGINE_FUSED).transform(img, seeds)
    want = ol.segment_par(img, seeds)[0]
    assert got.dtype == np.uint64 and (got == want).all()


@pytest.mark.parametrize("shape,seed", [((16, 16), 1), ((64, 64), 2), ((130, 67), 3), ((512, 512), 1)])
def test_segment_sweep_engine_bit_exact(pkg, shape, seed):
    img = cases.field(*shape, seed)
    seeds = ol.find_local_minima(img)
    got = _seg(pkg, pkg.ENGINE_SWEEP).transform(img, seeds)
    assert (got == ol.segment_par(img, seeds)[0]).all()
    st = pkg.default_context().stats()
    assert st["sweep_steps"] == ol.segment(img, seeds, want_stats=True)[1].scans   # same scan count as the reference loop


def test_segment_config2_2048_bit_exact(pkg):
    # BASELINE config C2: 2048x2048 random field, segmenting, bit-exact vs the CPU oracle
    img = cases.field(2048, 2048, 1)
    ws = _seg(pkg)
    seeds = ws.find_local_minima(img)
    assert (seeds == ol.find_local_minima(img)).all()
    got = ws.transform(img, seeds)
    want = ol.segment_arrival(img, seeds)
    assert (got == want).all()
    assert (got[1:-1, 1:-1] != 0).all()


def test_segment_adversarial_cases_both_engines(pkg):
    for name, img, seeds in cases.adversarial_cases():
        seeds = cases.seeds_or_maxima(img, seeds)
        for edge in (False, True):
            want = ol.segment(img, seeds, edge=edge)
            for engine in (pkg.ENGINE_FUSED, pkg.ENGINE_SWEEP):
                got = _seg(pkg, engine, edge=edge).transform(img, seeds)
                assert got.shape == want.shape, (name, edge, engine)
                assert (got == want).all(), (name, edge, engine)


@pytest.mark.parametrize("maxlvl", [1, 2, 127, 253, 254])
def test_segment_max_water_level(pkg, maxlvl):
    img = cases.smooth_field(150, 170, 3)
    seeds = ol.find_local_minima(img)[::2]
    got = _seg(pkg, max_level=maxlvl).transform(img, seeds)
    assert (got == ol.segment(img, seeds, max_level=maxlvl)).all()


def test_segment_smooth_fields_many_rings(pkg):
    for seed in (1, 2):
        img = cases.smooth_field(400, 520, seed)
        seeds = ol.find_local_minima(img)
        want, st = ol.segment_par(img, seeds)
        assert st.max_rings > 20                 # long plateaus: far more rings per level than iid noise
        assert (_seg(pkg).transform(img, seeds) == want).all()


def test_segment_long_corridor_crosses_many_tiles(pkg):
    # serpentine corridor 1 px wide through a 200x200 wall: arrival rings in the thousands,
    # the path crosses tile borders hundreds of times
    img = np.full((200, 200), 255, np.uint8)
    for r in range(1, 199, 2):
        img[r, 1:199] = 5
        img[r + 1, 198 if (r // 2) % 2 == 0 else 1] = 5
    seeds = [(1, 1)]
    want, al, ar = ol.segment(img, seeds, want_arrival=True)
    assert ar.max() > 5000
    assert (_seg(pkg).transform(img, seeds) == want).all()


def test_segment_output_is_a_reachable_sample_of_the_reference(pkg):
    img = cases.field(200, 260, 12)
    seeds = ol.find_local_minima(img)
    got = _seg(pkg).transform(img, seeds)
    assert ol.check_reachable(img, seeds, got)[0] == 0


def test_segment_is_deterministic_across_runs(pkg):
    img = cases.field(700, 900, 13)
    ws = _seg(pkg)
    seeds = ws.find_local_minima(img)
    a = ws.transform(img, seeds)
    for _ in range(3):
        assert (ws.transform(img, seeds) == a).all()


def test_seed_out_of_bounds_is_an_error(pkg):
    img = cases.field(32, 32, 1)
    for engine in (pkg.ENGINE_FUSED, pkg.ENGINE_SWEEP):
        with pytest.raises(pkg.SeedOutOfBounds):
            _seg(pkg, engine).transform(img, [(5, 5), (32, 0)])
    # with edge correction the plane is 34x34 and seeds are not shifted (lib.rs:1675-1677)
    got = _seg(pkg, edge=True).transform(img, [(33, 33), (5, 5)])
    assert got.shape == (34, 34) and got[33, 33] == 1


def test_no_seeds_and_empty_images(pkg):
    img = cases.field(40, 40, 2)
    assert _seg(pkg).transform(img, np.zeros((0, 2), np.uint64)).sum() == 0
    for shape in ((0, 0), (0, 5), (5, 0), (1, 1), (2, 2)):
        out = _seg(pkg).transform(np.zeros(shape, np.uint8), [])
        assert out.shape == shape
    # edge correction of an EMPTY image with a long side: the plane is two border rows (columns) of that length, and the
    # kernels read a clamped address of the stand-in block for every one of its pixels (the block is sized by the long side)
    for shape in ((0, 5000), (5000, 0), (0, 100003)):
        out = _seg(pkg, edge=True).transform(np.zeros(shape, np.uint8), [])
        assert out.shape == (shape[0] + 2, shape[1] + 2) and out.sum() == 0
        out = _seg(pkg, edge=True).transform(np.zeros(shape, np.uint8), [(0, 1), (1, 0)] if shape[1] else [(0, 0)])
        assert out.shape == (shape[0] + 2, shape[1] + 2) and int((out != 0).sum()) == (2 if shape[1] else 1)      # seeds keep their colour, nothing floods


# ---- hooks / history (lib.rs:1796-1804, 1824-1835) ----------------------------------------

@pytest.mark.parametrize("engine_name", ["ENGINE_FUSED", "ENGINE_SWEEP"])
def test_transform_history_matches_oracle_per_level(pkg, engine_name):
    img = cases.field(96, 120, 5)
    seeds = ol.find_local_minima(img)
    want = []
    ol.segment(img, seeds, max_level=60, hook=lambda l, m, i, c: want.append(c.copy()))
    hist = _seg(pkg, getattr(pkg, engine_name), max_level=60).transform_history(img, seeds)
    assert [l for l, _ in hist] == list(range(61))
    for (lvl, got), w in zip(hist, want):
        assert (got == w).all(), lvl


def test_transform_history_of_a_plane_widened_by_host_threads(pkg):
    # from 2^21 pixels on a hook's plane reaches the host as u32 chunks widened by host threads (ws_hostcopy.hip): segmenting
    # (a level's snapshot of the stamps) and merging (the level's relabelled plane), every level against the oracle's
    img = cases.field(1500, 1400, 8)
    seeds = ol.find_local_minima(img)
    want = []
    ol.segment(img, seeds, max_level=9, hook=lambda l, m, i, c: want.append(c.copy()))
    hist = _seg(pkg, max_level=9).transform_history(img, seeds)
    assert [l for l, _ in hist] == list(range(10))
    for (lvl, got), w in zip(hist, want):
        assert (got == w).all(), lvl
    wantm = []
    ol.merge(img, seeds, max_level=9, hook=lambda l, m, i, c: wantm.append(ol.canonicalise(c, seeds)[0]))
    b = pkg.TransformBuilder.new().set_max_water_lvl(9)
    histm = b.build_merging().transform_history(img, seeds)
    assert len(histm) == 10
    for (lvl, got), w in zip(histm, wantm):
        assert (got == w).all(), lvl


def test_transform_with_hook_receives_hookctx(pkg):
    img = cases.field(50, 60, 6)
    seeds = ol.find_local_minima(img)

    def hook(ctx):
        assert ctx.image.shape == (52, 62) and ctx.colours.shape == (52, 62)        # padded: edge correction on
        assert ctx.seeds[0] == (1, (int(seeds[0][0]), int(seeds[0][1])))            # (colour, (row, col)) lib.rs:1671
        return ctx.water_level, ctx.max_water_level, int((ctx.colours != 0).sum())

    res = _seg(pkg, max_level=9, edge=True, hook=hook).transform_with_hook(img, seeds)
    want = []
    ol.segment(img, seeds, max_level=9, edge=True, hook=lambda l, m, i, c: want.append((l, m, int((c != 0).sum()))))
    assert res == want
    assert _seg(pkg, max_level=9).transform_with_hook(img, seeds) == []                # no hook: empty Vec


# ---- device-resident path and full-size properties ---------------------------------------------

def _torch_engine(pkg):
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    return dev.DeviceEngine(0)


def _verify_fixpoint_on_device(img, seeds, labels, keys, max_level=254):
    """Size-independent proof of correctness with plain torch ops: the arrival stamps satisfy
    key = max(base, 1 + min4(key)) (unique fixpoint of the flood) and every
    flooded pixel carries the label of its first earlier-arrived neighbour in D,R,L,U order."""
    import torch
    H, W = img.shape
    INF = 0xFF000000
    k = keys.to(torch.int64) & 0xFFFFFFFF
    lab = labels.to(torch.int64) & 0xFFFFFFFF
    big = torch.full((H + 2, W + 2), INF, dtype=torch.int64, device=img.device)
    big[1:-1, 1:-1] = k
    d, r, l, u = big[2:, 1:-1], big[1:-1, 2:], big[1:-1, :-2], big[:-2, 1:-1]
    m = torch.minimum(torch.minimum(d, r), torch.minimum(l, u))
    base = torch.full((H, W), INF, dtype=torch.int64, device=img.device)
    inter = torch.zeros((H, W), dtype=torch.bool, device=img.device)
    inter[1:-1, 1:-1] = True
    v = img.to(torch.int64)
    ok = inter & (v <= max_level)
    base[ok] = (v[ok] << 24) | 1
    want = torch.minimum(torch.maximum(base, m + 1), torch.full_like(base, INF))
    seedmask = torch.zeros((H, W), dtype=torch.bool, device=img.device)
    s = seeds.to(torch.int64)
    seedmask[s[:, 0], s[:, 1]] = True
    assert bool((k[seedmask] == 0).all())
    assert bool((k[~seedmask] == want[~seedmask]).all()), "arrival stamps are not the flood fixpoint"
    # labels: seeds carry index+1 (last duplicate wins); others copy their parent
    lbig = torch.zeros((H + 2, W + 2), dtype=torch.int64, device=img.device)
    lbig[1:-1, 1:-1] = lab
    ld, lr, ll, lu = lbig[2:, 1:-1], lbig[1:-1, 2:], lbig[1:-1, :-2], lbig[:-2, 1:-1]
    parent = torch.where(d < k, ld, torch.where(r < k, lr, torch.where(l < k, ll, lu)))
    flooded = (~seedmask) & (k != INF)
    assert bool((lab[flooded] == parent[flooded]).all()), "a label is not its parent's label"
    assert bool((lab[(~seedmask) & (k == INF)] == 0).all())
    idx = torch.arange(1, s.shape[0] + 1, device=img.device, dtype=torch.int64)
    assert bool((lab[s[:, 0], s[:, 1]] == idx).all())          # maxima are distinct pixels: no duplicates
    return int(flooded.sum())


def test_device_random_field_matches_oracle_generator(pkg):
    eng = _torch_engine(pkg)
    for (h, w, seed) in ((37, 53, 0), (256, 300, 5)):
        assert (eng.random_field(h, w, seed).cpu().numpy() == ol.random_field(h, w, seed)).all()


def test_device_path_matches_host_path(pkg):
    import torch
    eng = _torch_engine(pkg)
    img = eng.random_field(777, 1234, 3)
    seeds = eng.find_local_minima(img)
    labels = eng.segment(img, seeds)
    torch.cuda.synchronize()
    himg = img.cpu().numpy()
    hseeds = seeds.cpu().numpy().astype(np.uint64)
    assert (hseeds == ol.find_local_minima(himg)).all()
    assert (labels.cpu().numpy().view(np.uint32) == ol.segment_arrival(himg, hseeds)).all()


def test_segment_u32_host_labels_equal_the_usize_plane(pkg):
    # ws_segment_u32: the same transform with the labels as the device holds them (half the bytes over PCIe)
    import ctypes
    img = cases.field(300, 420, 6)
    seeds = np.ascontiguousarray(np.asarray(ol.find_local_minima(img), dtype=np.uint64))
    want = ol.segment(img, seeds)
    for edge in (False, True):
        ws = _seg(pkg, edge=edge)
        c, opt = ws._ctx(), ws._opt
        e = 2 if edge else 0
        out = np.zeros((300 + e, 420 + e), dtype=np.uint32)
        rc = pkg._ffi.lib().ws_segment_u32(c.handle, img.ctypes.data, 300, 420, 420, seeds.ctypes.data, len(seeds), ctypes.byref(opt), out.ctypes.data)
        assert rc == 0, rc
        assert (out == (ol.segment(img, seeds, edge=True) if edge else want)).all()
        assert pkg._ffi.lib().ws_segment_u32(c.handle, img.ctypes.data, 300, 420, 420, seeds.ctypes.data, len(seeds), ctypes.byref(opt), None) == pkg._ffi.WS_ERR_BAD_ARG


@pytest.mark.parametrize("shape", [(1500, 1401), (2048, 3072), (1024, 2048), (4096, 4096), (4099, 5003)])
def test_usize_host_labels_of_planes_that_cross_the_bus_in_chunks(pkg, shape):
    # ws_hostcopy.hip: from 2^21 pixels on the u64 plane is not copied but WIDENED by host threads out of chunks of the u32 plane
    # (a quarter of the plane each, 2^19 .. 2^22 labels: four chunks and a few pixels, four exactly, the smallest plane that goes
    # this way, four of the largest chunks, five of them and a rest); the caller's plane may sit on any 8-byte boundary.  Against ws_segment_u32 (one plain copy of the same device labels) and, at the smallest size, the oracle.
    import ctypes
    h, w = shape
    ws = _seg(pkg)
    c, opt = ws._ctx(), ws._opt
    img = np.ascontiguousarray(_torch_engine(pkg).random_field(h, w, 5).cpu().numpy())
    seeds = np.ascontiguousarray(np.asarray(ol.find_local_minima(img), dtype=np.uint64))
    L = pkg._ffi.lib()
    out32 = np.zeros((h, w), dtype=np.uint32)
    assert L.ws_segment_u32(c.handle, img.ctypes.data, h, w, w, seeds.ctypes.data, len(seeds), ctypes.byref(opt), out32.ctypes.data) == 0
    for shift in (0, 1):
        buf = np.full(h * w + 3, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
        out = buf[1 + shift:1 + shift + h * w]
        assert L.ws_segment(c.handle, img.ctypes.data, h, w, w, seeds.ctypes.data, len(seeds), ctypes.byref(opt), out.ctypes.data) == 0
        assert (out.reshape(h, w) == out32).all()
        assert buf[shift] == 0xDEADBEEFDEADBEEF and buf[1 + shift + h * w] == 0xDEADBEEFDEADBEEF      # nothing beside the plane
        n_found = ctypes.c_size_t(0)
        out[:] = 7
        assert L.ws_segment_minima(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), out.ctypes.data, None, 0, ctypes.byref(n_found)) == 0
        assert n_found.value == len(seeds) and (out.reshape(h, w) == out32).all()
    if h * w < 3_000_000:
        assert (out32 == ol.segment_arrival(img, seeds)).all()
    # the (usize, usize) list (from 2^21 words on: the two largest planes here) takes the same road (2 x n words): ws_find_local_minima, ws_segment_minima with the list wanted, a
    # list cut short by its capacity (the status survives the copy)
    got = np.zeros((len(seeds) + 5, 2), dtype=np.uint64)
    assert L.ws_find_local_minima(c.handle, img.ctypes.data, h, w, w, got.ctypes.data, len(got), ctypes.byref(n_found)) == 0
    assert n_found.value == len(seeds) and (got[:len(seeds)] == seeds).all() and not got[len(seeds):].any()
    got[:] = 0
    out = np.zeros(h * w, dtype=np.uint64)
    assert L.ws_segment_minima(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), out.ctypes.data, got.ctypes.data, len(got), ctypes.byref(n_found)) == 0
    assert n_found.value == len(seeds) and (got[:len(seeds)] == seeds).all() and (out.reshape(h, w) == out32).all()
    short = np.zeros((len(seeds) - 3, 2), dtype=np.uint64)
    assert L.ws_find_local_minima(c.handle, img.ctypes.data, h, w, w, short.ctypes.data, len(short), ctypes.byref(n_found)) == pkg._ffi.WS_ERR_CAPACITY
    assert n_found.value == len(seeds) and (short == seeds[:len(short)]).all()
    # ws_ctx_set_host_threads: none (the plane widened on the device, one copy), one, more than there are chunks' worth of rows
    out = np.zeros(h * w, dtype=np.uint64)
    for threads in (0, 1, 7):
        assert L.ws_ctx_set_host_threads(c.handle, threads) == 0
        out[:] = 9
        assert L.ws_segment(c.handle, img.ctypes.data, h, w, w, seeds.ctypes.data, len(seeds), ctypes.byref(opt), out.ctypes.data) == 0
        assert (out.reshape(h, w) == out32).all(), threads
    assert L.ws_ctx_set_host_threads(c.handle, -1) == pkg._ffi.WS_ERR_BAD_ARG and L.ws_ctx_set_host_threads(c.handle, 65) == pkg._ffi.WS_ERR_BAD_ARG


def test_usize_planes_of_random_shapes_equal_the_u32_planes(pkg):
    # the chunk arithmetic of ws_hostcopy.hip on odd sizes: planes of 2.1 .. 9 M pixels of random shape (chunks that are not a
    # multiple of the row, a last chunk of any length, threads 1 .. 6), the usize plane against the device's own u32 plane
    import ctypes
    rng = np.random.default_rng(99)
    eng = _torch_engine(pkg)
    ws = _seg(pkg)
    c, opt, L = ws._ctx(), ws._opt, pkg._ffi.lib()
    n_found = ctypes.c_size_t(0)
    for case in range(10):
        h = int(rng.integers(700, 3000))
        w = int(rng.integers((1 << 21) // h + 1, 9_000_000 // h))
        img = np.ascontiguousarray(eng.random_field(h, w, 300 + case).cpu().numpy())
        out32 = np.zeros((h, w), dtype=np.uint32)
        out64 = np.full(h * w + 1, 5, dtype=np.uint64)
        assert L.ws_ctx_set_host_threads(c.handle, 1 + case % 6) == 0
        assert L.ws_segment_minima_u32(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), out32.ctypes.data, None, 0, ctypes.byref(n_found)) == 0
        assert L.ws_segment_minima(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), out64[case % 2:].ctypes.data, None, 0, ctypes.byref(n_found)) == 0
        assert (out64[case % 2:case % 2 + h * w].reshape(h, w) == out32).all(), (h, w)
        assert out64[h * w if case % 2 == 0 else 0] == 5


@pytest.mark.parametrize("edge", [False, True])
def test_long_usize_seed_lists_through_the_host_abi(pkg, edge):
    # 2.5 M (usize, usize) pairs through ws_segment (copied whole, checked and narrowed by k_narrow_seeds); with the seed_shift option
    # (edge correction's padded plane, the caller's coordinates moved by (+1, +1)) they move on the way.  Against the device-resident
    # call with the same list, and the reference's panic for a seed outside the plane (lib.rs:1675-1677) wherever in the list it sits.
    import ctypes
    import torch
    h, w = 2200, 2304
    eng = _torch_engine(pkg)
    img_t = eng.random_field(h, w, 11)
    img = np.ascontiguousarray(img_t.cpu().numpy())
    rr, cc = np.meshgrid(np.arange(1, h - 1, dtype=np.uint64), np.arange(1, w - 1, dtype=np.uint64), indexing="ij")
    pick = ((rr + cc) % 2 == 0)
    seeds = np.ascontiguousarray(np.stack([rr[pick], cc[pick]], axis=1))      # row-major, strictly increasing
    assert len(seeds) > (1 << 21) + 1000
    want = eng.segment(img_t, torch.from_numpy(seeds.astype(np.int32)).to(img_t.device), edge=edge, seed_shift=edge)
    torch.cuda.synchronize()
    ws = _seg(pkg, edge=edge)
    c = ws._ctx()
    opt = pkg._ffi.Options(254, int(edge), 0, 0, int(edge))
    L = pkg._ffi.lib()
    e = 2 if edge else 0
    out = np.zeros((h + e, w + e), dtype=np.uint64)
    assert L.ws_segment(c.handle, img.ctypes.data, h, w, w, seeds.ctypes.data, len(seeds), ctypes.byref(opt), out.ctypes.data) == 0
    assert (out == want.cpu().numpy().view(np.uint32)).all()
    for where in (5, (1 << 21) - 1, (1 << 21) + 7, len(seeds) - 1):
        bad = seeds.copy()
        bad[where] = (h - 1, w + 1) if where % 2 else (1 << 40, 3)      # (with the shift the caller's plane is h x w: column w + 1 is outside either way)
        assert L.ws_segment(c.handle, img.ctypes.data, h, w, w, bad.ctypes.data, len(bad), ctypes.byref(opt), out.ctypes.data) == pkg._ffi.WS_ERR_SEED_OOB


@pytest.mark.parametrize("edge", [False, True])
def test_host_cube_of_slices_equals_the_loop_over_them(pkg, edge):
    # ws_segment_batch: what tests/integration.rs:267,356 loops over (find_local_minima + transform per slice of a cube) as one
    # call whose slices take turns on three internal contexts.  Every slice against the oracle, with its own minima and with
    # lists of the caller's (an empty one, unsorted ones); cubes of fewer slices than lanes.
    import ctypes
    rng = np.random.default_rng(5)
    cube = np.stack([cases.field(130, 96, 40 + k) if k % 3 else cases.smooth_field(130, 96, 40 + k) for k in range(7)])
    ws = _seg(pkg, edge=edge)
    got, counts = ws.transform_cube(cube)
    for k in range(7):
        seeds = ol.find_local_minima(cube[k])
        assert counts[k] == len(seeds)
        assert (got[k] == ol.segment(cube[k], seeds, edge=edge)).all(), k
    lists = []
    for k in range(7):
        s = np.asarray(ol.find_local_minima(cube[k]), dtype=np.uint64).reshape(-1, 2)
        s = s[rng.permutation(len(s))[: len(s) // (k + 1)]] if k != 3 else s[:0]
        lists.append(s)
    got = ws.transform_cube(cube, lists)
    for k in range(7):
        assert (got[k] == ol.segment(cube[k], lists[k], edge=edge)).all(), k
    for n in (0, 1, 2):
        got, counts = ws.transform_cube(cube[:n])
        assert got.shape[0] == n
        for k in range(n):
            assert (got[k] == ol.segment(cube[k], ol.find_local_minima(cube[k]), edge=edge)).all()
    # the LOWEST failing slice is the one reported, whatever the lanes' pace (lib.rs:1675-1677: the reference panics on a seed
    # outside the plane, in the first slice that has one)
    c, L = ws._ctx(), pkg._ffi.lib()
    e = 2 if edge else 0
    out = np.zeros((7, 130 + e, 96 + e), dtype=np.uint64)
    for bad_slices in ((2, 5), (6, 4), (0,), (3, 1, 2)):
        bad = [l.copy() for l in lists]
        for k in bad_slices:
            bad[k] = np.array([[1, 1], [4000 + k, 3]], dtype=np.uint64)
        flat = np.ascontiguousarray(np.concatenate(bad, axis=0))
        offs = np.zeros(8, dtype=np.uintp)
        offs[1:] = np.cumsum([len(l) for l in bad])
        failed = ctypes.c_size_t(99)
        rc = L.ws_segment_batch(c.handle, cube.ctypes.data, 7, 130, 96, 96, 130 * 96, flat.ctypes.data, offs.ctypes.data_as(pkg._ffi.szp),
                                ctypes.byref(ws._opt), out.ctypes.data, None, ctypes.byref(failed))
        assert rc == pkg._ffi.WS_ERR_SEED_OOB and failed.value == min(bad_slices), (bad_slices, failed.value)
        assert ("slice %d" % min(bad_slices)).encode() in L.ws_last_error(c.handle)
    got = ws.transform_cube(cube, lists)      # ... and the context works on
    assert (got[6] == ol.segment(cube[6], lists[6], edge=edge)).all()


def test_host_cube_of_large_slices(pkg):
    # slices of 2^21 pixels and more: the lanes' label planes are widened by their host threads at once (ws_hostcopy.hip)
    cube = np.stack([cases.field(1500, 1400, 70 + k) for k in range(5)])
    ws = _seg(pkg)
    got, counts = ws.transform_cube(cube)
    for k in range(5):
        seeds = ol.find_local_minima(cube[k])
        assert counts[k] == len(seeds) and (got[k] == ol.segment_arrival(cube[k], seeds)).all(), k


@pytest.mark.parametrize("shape,kind", [((96, 128), "noise"), ((257, 512), "noise"), ((300, 420), "noise"), ((130, 96), "smooth"), ((64, 1056), "smooth"),
                                        ((3, 32), "noise"), ((2, 64), "noise"), ((40, 64), "flat")])
@pytest.mark.parametrize("edge", [False, True])
def test_transform_from_minima_is_the_call_pair(pkg, shape, kind, edge):
    # ws_segment_minima: lib.rs:73-86's `find_local_minima` + `transform` as one call.  With w % 32 == 0 and no edge correction the
    # seed tables come out of the minima kernels (no list); every other shape runs the two calls inside the library.
    h, w = shape
    img = cases.field(h, w, 31) if kind == "noise" else (cases.smooth_field(h, w, 9) if kind == "smooth" else np.full(shape, 17, dtype=np.uint8))
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    want = ol.segment(img, seeds, edge=edge) if h * w else None
    ws = _seg(pkg, edge=edge)
    got, got_seeds = ws.transform_from_minima(img, want_seeds=True)
    assert got_seeds.shape == seeds.shape and (got_seeds == seeds).all()
    assert got.dtype == np.uint64 and (got == want).all()
    assert (ws.transform_from_minima(img) == want).all()                 # no list asked for: none is written
    got32 = ws.transform_from_minima(img, labels_u32=True)
    assert got32.dtype == np.uint32 and (got32 == want).all()
    # a seed buffer that is too small: the labels are complete, the list is cut, the status says so
    if len(seeds) > 1:
        c, opt = ws._ctx(), ws._opt
        e = 2 if edge else 0
        out = np.zeros((h + e, w + e), dtype=np.uint64)
        few = np.zeros((len(seeds) - 1, 2), dtype=np.uint64)
        n = ctypes.c_size_t(0)
        rc = pkg._ffi.lib().ws_segment_minima(c.handle, img.ctypes.data, h, w, w, ctypes.byref(opt), out.ctypes.data, few.ctypes.data, len(few), ctypes.byref(n))
        assert rc == pkg._ffi.WS_ERR_CAPACITY and n.value == len(seeds)
        assert (out == want).all() and (few == seeds[:-1]).all()


def test_transform_from_minima_on_the_device_and_after_other_transforms(pkg):
    # the device form, fast path (graph capture and replay on the third call), interleaved with list-seeded transforms of other planes
    import torch
    eng = _torch_engine(pkg)
    himg = cases.field(512, 768, 77)
    hseeds = np.asarray(ol.find_local_minima(himg), dtype=np.uint64).reshape(-1, 2)
    want = ol.segment_arrival(himg, hseeds)
    img = torch.from_numpy(himg).to(eng.device)
    out = torch.empty((512, 768), dtype=torch.int32, device=eng.device)
    for k in range(4):
        labels, n, seeds = eng.segment_minima(img, out=out, want_seeds=True)
        assert n == len(hseeds) and (seeds.cpu().numpy().astype(np.uint64) == hseeds).all()
        assert (labels.cpu().numpy().view(np.uint32) == want).all()
        if k == 1:      # something else in between: the tables must be rebuilt, not remembered
            other = cases.smooth_field(512, 768, 5)
            os_ = np.asarray(ol.find_local_minima(other), dtype=np.int32).reshape(-1, 2)
            got = eng.segment(torch.from_numpy(other).to(eng.device), torch.from_numpy(os_).to(eng.device))
            assert (got.cpu().numpy().view(np.uint32) == ol.segment_arrival(other, os_.astype(np.uint64))).all()
    labels, n = eng.segment_minima(img)
    assert n == len(hseeds) and (labels.cpu().numpy().view(np.uint32) == want).all()
    # a different image in the SAME buffer: a replayed graph reads the image again
    himg2 = cases.smooth_field(512, 768, 12)
    img.copy_(torch.from_numpy(himg2).to(eng.device))
    labels, n = eng.segment_minima(img, out=out)
    s2 = np.asarray(ol.find_local_minima(himg2), dtype=np.uint64).reshape(-1, 2)
    assert n == len(s2) and (labels.cpu().numpy().view(np.uint32) == ol.segment_arrival(himg2, s2)).all()


def test_transform_from_minima_at_the_headline_size_equals_the_two_calls(pkg):
    # 8192^2 (BASELINE's field): ws_segment_minima_device against ws_find_local_minima_device + ws_segment_device, which
    # test_headline_field_8192_equals_oracle compares with the oracle label by label
    import torch
    eng = _torch_engine(pkg)
    img = eng.random_field(8192, 8192, 1)
    seeds = eng.find_local_minima(img)
    want = eng.segment(img, seeds).clone()
    labels, n, got_seeds = eng.segment_minima(img, want_seeds=True)
    assert n == seeds.shape[0] and bool((got_seeds == seeds).all())
    assert bool((labels == want).all())
    labels2, n2 = eng.segment_minima(img)      # (no list)
    assert n2 == n and bool((labels2 == want).all())


@pytest.mark.parametrize("size", [4096, 8192])
def test_full_size_fixpoint_properties(pkg, size):
    # BASELINE.json headline size (8192^2) and the C4 slice size (4096^2): the oracle cannot run
    # these in seconds, so verify the defining equations of the result on the device instead
    import torch
    eng = _torch_engine(pkg)
    img = eng.random_field(size, size, 1)
    seeds = eng.find_local_minima(img)
    assert 0.105 < seeds.shape[0] / (size * size) < 0.113
    lin = seeds[:, 0].to(torch.int64) * size + seeds[:, 1].to(torch.int64)
    assert bool((lin[1:] > lin[:-1]).all())                     # row-major order
    labels = eng.segment(img, seeds)
    keys = eng.last_arrival()
    torch.cuda.synchronize()
    flooded = _verify_fixpoint_on_device(img, seeds, labels, keys)
    assert flooded + seeds.shape[0] == (size - 2) * (size - 2)   # every interior pixel ends up coloured
    st = eng.stats()
    assert st["relax_passes"] >= 2 and st["resolve_passes"] >= 2
    # a 512x512 crop-independent spot check against the oracle on the same generator stream is done
    # in test_device_path_matches_host_path; here also check idempotence of a second run
    again = eng.segment(img, seeds)
    assert bool((again == labels).all())


# ---- seed painting: one-pass path for sorted lists, fix-up for everything else (lib.rs:1672-1677) ----

@pytest.mark.parametrize("shape", [(700, 1100), (1023, 517), (64, 8)])
def test_seed_painting_sorted_duplicated_and_shuffled_lists(pkg, shape):
    img = cases.field(*shape, 21)
    base = ol.find_local_minima(img)                       # row-major, strictly increasing
    rng = np.random.default_rng(5)
    ws = _seg(pkg, pkg.ENGINE_FUSED)
    variants = {
        "sorted": base,
        "sorted_adjacent_duplicates": np.repeat(base, 1 + (np.arange(len(base)) % 5 == 0), axis=0),
        "shuffled": base[rng.permutation(len(base))],
        "one_swap_at_the_end": np.concatenate([base[:-2], base[-1:], base[-2:-1]]),
        "shuffled_with_duplicates": np.concatenate([base, base[::7]])[rng.permutation(len(base) + len(base[::7]))],
        "border_and_corner_seeds": np.concatenate([np.array([[0, 0], [0, shape[1] - 1]], np.uint64), base,
                                                   np.array([[shape[0] - 1, 0], [shape[0] - 1, shape[1] - 1]], np.uint64)]),
    }
    for name, seeds in variants.items():
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        got = ws.transform(img, seeds)
        want = ol.segment_arrival(img, seeds)
        assert got.shape == want.shape and (got == want).all(), name
    # a seed outside the plane anywhere in a sorted list is still reported (the reference panics: lib.rs:1676)
    bad = np.concatenate([base[: len(base) // 2], np.array([[shape[0], 0]], np.uint64), base[len(base) // 2:]])
    with pytest.raises(IndexError):
        ws.transform(img, bad)
    assert (ws.transform(img, base) == ol.segment_arrival(img, base)).all()         # and the context recovers


# ---- long-range floods: the whole-row scan of the late relaxation passes ---------------------------------

@pytest.mark.parametrize("shape,octaves,few_seeds", [((1100, 1600), 6, False), ((900, 2048), 7, True), ((700, 1030), 5, True)])
def test_segment_large_smooth_fields_long_range(pkg, shape, octaves, few_seeds):
    # correlation lengths of 64-256 pixels: floods cross many tiles, dozens to hundreds of passes; with only
    # a handful of seeds a single flood crosses the whole plane
    img = cases.smooth_field(*shape, 17, octaves=octaves)
    seeds = ol.find_local_minima(img)
    if few_seeds:
        seeds = seeds[:: max(len(seeds) // 5, 1)][:5]
    ws = _seg(pkg, pkg.ENGINE_FUSED)
    set_mode = pkg._ffi.lib().ws_ctx_set_persistent_pass
    assert set_mode(ws._ctx().handle, 0) == 0      # the passes themselves (with sparse seeds the default takes the tile queue)
    got = ws.transform(img, seeds)
    assert set_mode(ws._ctx().handle, 3) == 0
    want = ol.segment_arrival(img, seeds)
    assert got.shape == want.shape and (got == want).all()
    st = ws._ctx().stats()
    assert st["relax_passes"] >= 8           # the scan-capable kernel variant ran (passes >= 4)
    assert (ws.transform(img, seeds) == want).all()      # ... and whatever the default picks


@pytest.mark.parametrize("mode", [1, 2, 4])
@pytest.mark.parametrize("shape,octaves,few_seeds", [((1100, 1600), 6, False), ((900, 2048), 7, True), ((1500, 640), 7, True), ((2100, 300), 5, False)])
def test_persistent_tile_queue_pass_gives_the_same_labels(pkg, shape, octaves, few_seeds, mode):
    # ws_ctx_set_persistent_pass: the first same-grid pass of a long-range flood as ONE launch with a device-side tile queue
    # (k_relax, PERSIST), first come (1) or in flood order (2: buckets by level, a run hands itself the tile it announced).
    # Opt-in; the labels must be the oracle's whatever order the queue runs the tiles in.
    img = cases.smooth_field(shape[0], shape[1], 11 + octaves, octaves=octaves)
    seeds = ol.find_local_minima(img)
    if few_seeds:
        seeds = seeds[::max(len(seeds) // 3, 1)]
    ws = _seg(pkg)
    c = ws._ctx()
    assert pkg._ffi.lib().ws_ctx_set_persistent_pass(c.handle, 5) == pkg._ffi.WS_ERR_BAD_ARG
    assert pkg._ffi.lib().ws_ctx_set_persistent_pass(c.handle, mode) == 0
    got = ws.transform(img, seeds)
    st = c.stats()
    assert pkg._ffi.lib().ws_ctx_set_persistent_pass(c.handle, 3) == 0      # (the default: auto)
    assert st["relax_passes"] >= (9 if mode != 2 else 5)      # first come: pass 7 was the queue, pass 8 looked at every tile again; flood order: passes 3 and 4
    assert (got == ol.segment_arrival(img, seeds)).all()
    corridor = np.full((600, 1400), 255, dtype=np.uint8)      # one long winding corridor: a chain of tile runs, nothing in parallel
    corridor[5:595:10, 3:-3] = 7
    corridor[5:595:20, -6:-3] = 7
    for r in range(5, 585, 10):
        x = slice(-6, -3) if (r // 10) % 2 == 0 else slice(3, 6)
        corridor[r:r + 11, x] = 7
    cs = np.array([[5, 4]], dtype=np.uint64)
    assert pkg._ffi.lib().ws_ctx_set_persistent_pass(c.handle, mode) == 0
    got = ws.transform(corridor, cs)
    assert pkg._ffi.lib().ws_ctx_set_persistent_pass(c.handle, 3) == 0
    assert (got == ol.segment_arrival(corridor, cs)).all()


def test_segment_batch_of_independent_slices(pkg):
    # config C4 in miniature: a cube of slices, one call; equal to slice-by-slice calls and to the oracle
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    s, h, w = 5, 96, 200
    himgs = [cases.field(h, w, 40 + k) if k % 2 == 0 else cases.smooth_field(h, w, 40 + k) for k in range(s)]
    hseeds = [ol.find_local_minima(a) for a in himgs]
    hseeds[3] = hseeds[3][:0]                                            # a slice without seeds
    offs = np.concatenate([[0], np.cumsum([len(x) for x in hseeds])])
    cube = torch.from_numpy(np.stack(himgs)).to(eng.device)
    allseeds = torch.from_numpy(np.concatenate(hseeds).astype(np.int64).reshape(-1, 2)).to(torch.int32).to(eng.device).contiguous()
    got = eng.segment_batch(cube, allseeds, offs).cpu().numpy().view(np.uint32)
    for k in range(s):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k


def _batch_case(s, h, w, first_seed):
    """s slices, the local minima of each plus seeds on its first / last row and in a corner (border seeds keep their
    colour but never flood: in a stacked batch they sit right next to the neighbouring slice)."""
    himgs, hseeds = [], []
    for k in range(s):
        a = cases.field(h, w, first_seed + k) if k % 2 == 0 else cases.smooth_field(h, w, first_seed + k)
        extra = np.array([[0, 0], [0, w // 2], [h - 1, 1], [h - 1, w - 1]], dtype=np.int64)
        sd = np.unique(np.concatenate([ol.find_local_minima(a).astype(np.int64).reshape(-1, 2), extra]), axis=0)      # sorted row-major
        himgs.append(a)
        hseeds.append(sd)
    return himgs, hseeds


def _run_batch(eng, himgs, hseeds, edge=False):
    import torch
    offs = np.concatenate([[0], np.cumsum([len(x) for x in hseeds])])
    cube = torch.from_numpy(np.stack(himgs)).to(eng.device)
    allseeds = torch.from_numpy(np.concatenate(hseeds).astype(np.int64).reshape(-1, 2)).to(torch.int32).to(eng.device).contiguous()
    return eng.segment_batch(cube, allseeds, offs, edge=edge).cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("s,h,w,edge", [
    (4, 40, 64, False),      # slices end inside relaxation tiles (32 rows) and resolve tiles (64 rows)
    (3, 64, 64, False),      # slices = resolve tiles
    (6, 30, 128, False),
    (9, 8, 16, False),       # many thin slices inside one tile
    (3, 62, 62, True),       # edge correction: padded slices of 64 x 64
    (2, 33, 36, False),      # 33 * 36 % 128 != 0: the slice-by-slice loop
    (3, 40, 66, False),      # w % 4 != 0: the loop
])
def test_segment_batch_stacked_forms(pkg, s, h, w, edge):
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    himgs, hseeds = _batch_case(s, h, w, 300 + s)
    got = _run_batch(eng, himgs, hseeds, edge)
    for k in range(s):
        want = ol.segment_arrival(himgs[k], hseeds[k], edge=edge)
        assert got[k].shape == want.shape and (got[k] == want).all(), k


def test_segment_batch_unsorted_list_and_bad_seed(pkg):
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    himgs, hseeds = _batch_case(4, 32, 64, 77)
    hseeds[2] = hseeds[2][::-1].copy()                      # one list in decreasing order: the painted form, slice by slice
    got = _run_batch(eng, himgs, hseeds)
    for k in range(4):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k
    hseeds[2] = hseeds[2][::-1].copy()
    got = _run_batch(eng, himgs, hseeds)                    # sorted again: the context goes back to the stacked form
    for k in range(4):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k
    # a seed below its own slice would land in the next slice of the stack: it must be an error, as in a single call
    bad = [x.copy() for x in hseeds]
    bad[1] = np.concatenate([bad[1], [[32, 5]]])
    with pytest.raises(Exception) as ei:
        _run_batch(eng, himgs, bad)
    assert "seed" in str(ei.value).lower()
    got = _run_batch(eng, himgs, hseeds)                    # and the context is usable afterwards
    for k in range(4):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k


def test_graph_replay_of_repeated_transforms(pkg):
    """A transform that repeats the previous one's buffers, sizes and seed count replays its first passes as one hipGraph
    (third call on).  The CONTENTS of the buffers change between the calls: every replay must compute afresh, and a flood
    that needs more passes than the graph holds must go on with the ordinary loop."""
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    with torch.cuda.stream(torch.cuda.Stream()):      # capture is not allowed on the legacy null stream
        eng = dev.DeviceEngine(0)
        h, w = 96, 256
        imgs = [cases.field(h, w, 500), cases.smooth_field(h, w, 501), cases.field(h, w, 502)]
        lists = [ol.find_local_minima(a).astype(np.int64).reshape(-1, 2) for a in imgs]
        # a long corridor (one seed at its end, walls elsewhere): hundreds of rings, far more passes than the graph holds
        maze = np.full((h, w), 255, np.uint8)
        for k, y in enumerate(range(2, h - 2, 4)):
            maze[y, 2:w - 2] = 7
            maze[y:y + 5, (w - 3) if k % 2 == 0 else 2] = 7
        imgs.append(maze)
        lists.append(np.array([[2, 2]], dtype=np.int64))
        n = min(len(x) for x in lists[:3])
        d_img = torch.empty((h, w), dtype=torch.uint8, device=eng.device)
        d_seeds = torch.empty((n, 2), dtype=torch.int32, device=eng.device)
        out = torch.empty((h, w), dtype=torch.int32, device=eng.device)
        replays = 0
        for rep in range(9):
            k = rep % 3
            seeds = lists[k][:n]                      # a prefix of a strictly increasing list is strictly increasing
            d_img.copy_(torch.from_numpy(imgs[k]))
            d_seeds.copy_(torch.from_numpy(seeds).to(torch.int32))
            got = eng.segment(d_img, d_seeds, out=out).cpu().numpy().view(np.uint32)
            replays += eng.stats()["graph_launches"]
            assert (got == ol.segment_arrival(imgs[k], seeds)).all(), rep
            if rep in (4, 7):                         # same buffers, the merging transform in between (another key)
                m = eng.merge(d_img, d_seeds).cpu().numpy().view(np.uint32)
                assert (m == ol.merge_arrival(imgs[k], seeds)).all(), rep
        assert replays >= 3, replays
        # the corridor through the same buffers and seed count 1: replayed graph, gate closed, ordinary loop afterwards
        d_one = torch.empty((1, 2), dtype=torch.int32, device=eng.device)
        d_one.copy_(torch.from_numpy(lists[3]).to(torch.int32))
        d_img.copy_(torch.from_numpy(maze))
        replays = 0
        set_mode = pkg._ffi.lib().ws_ctx_set_persistent_pass
        assert set_mode(eng.ctx.handle, 0) == 0      # (the passes: with one seed the default would end inside the graph, through the tile queue)
        for rep in range(4):
            got = eng.segment(d_img, d_one, out=out).cpu().numpy().view(np.uint32)
            replays += eng.stats()["graph_launches"]
            assert eng.stats()["relax_passes"] > 8
            assert (got == ol.segment_arrival(maze, lists[3])).all(), rep
        assert replays >= 2, replays
        assert set_mode(eng.ctx.handle, 3) == 0
        for rep in range(3):                         # ... and the default: the queue inside the replayed graph
            got = eng.segment(d_img, d_one, out=out).cpu().numpy().view(np.uint32)
            assert (got == ol.segment_arrival(maze, lists[3])).all(), rep
        # merging and batches repeat as well (their own keys)
        d_img.copy_(torch.from_numpy(imgs[0]))
        d_seeds.copy_(torch.from_numpy(lists[0][:n]).to(torch.int32))
        mout = torch.empty((h, w), dtype=torch.int32, device=eng.device)
        for rep in range(4):
            m = eng.merge(d_img, d_seeds, out=mout).cpu().numpy().view(np.uint32)
            assert (m == ol.merge_arrival(imgs[0], lists[0][:n])).all(), rep
        assert eng.stats()["graph_launches"] == 1
        # ... with its unions queued behind the graph before the host has looked (ws_merge_device): contents that change
        # between replays, and a flood the graph's passes do not finish (the unions then ran on stale labels and are redone)
        for rep in range(6):
            k = rep % 3
            d_img.copy_(torch.from_numpy(imgs[k]))
            d_seeds.copy_(torch.from_numpy(lists[k][:n]).to(torch.int32))
            m = eng.merge(d_img, d_seeds, out=mout).cpu().numpy().view(np.uint32)
            assert (m == ol.merge_arrival(imgs[k], lists[k][:n])).all(), rep
        d_img.copy_(torch.from_numpy(maze))
        assert set_mode(eng.ctx.handle, 0) == 0      # (as above: the passes, so that the flood outlasts the graph)
        for rep in range(4):
            m = eng.merge(d_img, d_one, out=mout).cpu().numpy().view(np.uint32)
            assert eng.stats()["relax_passes"] > 8
            assert (m == ol.merge_arrival(maze, lists[3])).all(), rep
        assert eng.stats()["graph_launches"] == 1
        assert set_mode(eng.ctx.handle, 3) == 0
        for rep in range(3):
            m = eng.merge(d_img, d_one, out=mout).cpu().numpy().view(np.uint32)
            assert (m == ol.merge_arrival(maze, lists[3])).all(), rep
        himgs, hseeds = _batch_case(4, 32, 64, 900)
        offs = np.concatenate([[0], np.cumsum([len(x) for x in hseeds])])
        cube = torch.from_numpy(np.stack(himgs)).to(eng.device)
        allseeds = torch.from_numpy(np.concatenate(hseeds).astype(np.int64).reshape(-1, 2)).to(torch.int32).to(eng.device).contiguous()
        bout = torch.empty((4, 32, 64), dtype=torch.int32, device=eng.device)
        for rep in range(4):
            got = eng.segment_batch(cube, allseeds, offs, out=bout).cpu().numpy().view(np.uint32)
            for k in range(4):
                assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), (rep, k)
        assert eng.stats()["graph_launches"] == 1


def test_segment_batch_in_several_groups(pkg):
    # a stack may hold at most 2^31 pixels; larger batches are cut into groups.  ws_ctx_set_batch_pixel_limit lowers the
    # limit so that a small batch exercises the grouping (3 + 3 + 1 slices, the last group a single slice)
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    himgs, hseeds = _batch_case(7, 32, 64, 4100)
    eng.ctx.set_batch_pixel_limit(3 * 32 * 64)
    got = _run_batch(eng, himgs, hseeds)
    for k in range(7):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k
    eng.ctx.set_batch_pixel_limit(0)
    got = _run_batch(eng, himgs, hseeds)          # one stack again
    for k in range(7):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k


def test_seed_tables_walk_rejects_every_list_that_is_not_strictly_increasing(pkg):
    """The side-table builder reads the list once and proves on the way that it is strictly increasing and in bounds
    (every wave checks the list range between the lower bounds of its 8192-pixel chunk's two ends).  Lists that are
    sorted almost everywhere must still be caught -- and then give the labels of the general path."""
    shape = (700, 1100)                                    # 94 chunks of 8192 pixels
    img = cases.field(*shape, 23)
    base = np.ascontiguousarray(ol.find_local_minima(img), dtype=np.uint64)
    pos = base[:, 0] * shape[1] + base[:, 1]
    cut = int(np.searchsorted(pos, 8192 * 40))             # first seed of chunk 40
    ws = _seg(pkg, pkg.ENGINE_FUSED)
    variants = {
        "halves_swapped": np.concatenate([base[len(base) // 2:], base[: len(base) // 2]]),
        "far_duplicate": np.concatenate([base[: len(base) // 2], base[10:11], base[len(base) // 2:]]),
        "swap_across_a_chunk_boundary": np.concatenate([base[: cut - 1], base[cut:cut + 1], base[cut - 1:cut], base[cut + 1:]]),
        "swap_inside_a_chunk": np.concatenate([base[: cut + 3], base[cut + 4:cut + 5], base[cut + 3:cut + 4], base[cut + 5:]]),
        "first_two_swapped": np.concatenate([base[1:2], base[0:1], base[2:]]),
        "one_chunk_repeated": np.concatenate([base[:cut], base[cut - 50:cut], base[cut:]]),
        "reversed": base[::-1],
    }
    for name, seeds in variants.items():
        assert (ws.transform(img, base) == ol.segment_arrival(img, base)).all()          # the context expects a sorted list again
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        got = ws.transform(img, seeds)
        assert (got == ol.segment_arrival(img, seeds)).all(), name
    for name, bad in {
        "column_outside_in_the_middle": np.concatenate([base[:cut], np.array([[base[cut][0], shape[1] + 3]], np.uint64), base[cut:]]),
        "row_outside_at_the_end": np.concatenate([base, np.array([[shape[0] + 5, 1]], np.uint64)]),
        "row_outside_at_the_start": np.concatenate([np.array([[shape[0], 0]], np.uint64), base]),
    }.items():
        assert (ws.transform(img, base) == ol.segment_arrival(img, base)).all()
        with pytest.raises(IndexError):
            ws.transform(img, np.ascontiguousarray(bad, dtype=np.uint64))


def test_segment_batch_stacked_with_an_unsorted_list_does_not_chase_garbage(pkg):
    # Regression: in the stacked form the side tables subtract every slice's first list index; built from a list that is
    # not strictly increasing that difference can be any word, also one with the reference bit set, and the reference
    # chase followed it out of the plane (a GPU memory fault) before the transform was repeated slice by slice.
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    rng = np.random.default_rng(9)
    h, w = 32, 184
    himgs, hseeds = [], []
    for k, n in enumerate((289, 5, 28)):
        himgs.append(rng.integers(0, 254, (h, w), dtype=np.uint8))
        flat = np.sort(rng.choice(h * w, size=n, replace=False))
        hseeds.append(np.stack([flat // w, flat % w], axis=1).astype(np.int64))
    hseeds[1] = hseeds[1][::-1].copy()
    for _ in range(2):
        got = _run_batch(eng, himgs, hseeds)
        for k in range(3):
            assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k].astype(np.uint64))).all(), k
        # the prediction flipped to "unsorted"; a sorted batch in between flips it back, so that the stacked form is
        # tried on the unsorted list again
        sorted_lists = [hseeds[0], hseeds[1][::-1].copy(), hseeds[2]]
        got = _run_batch(eng, himgs, sorted_lists)
        for k in range(3):
            assert (got[k] == ol.segment_arrival(himgs[k], sorted_lists[k].astype(np.uint64))).all(), k


@pytest.mark.parametrize("shape,octaves", [((1000, 1300), 6), ((777, 1001), 6), ((1500, 640), 7), ((2100, 300), 5),
                                           ((193, 1290), 5), ((1290, 193), 5), ((449, 385), 6), ((129, 2050), 5), ((640, 1024), 6)])
def test_long_range_passes_on_one_grid_odd_shapes(pkg, shape, octaves):
    # the late passes (compacted tile lists from pass 6, ONE grid of 128 x 64 tiles from pass 7 -- two half-width bands per
    # wave --, round caps) on shapes whose last tile row / column is partial in either geometry (one pixel past a multiple of
    # 64 rows / 128 columns, exact multiples, a plane of one tile and a bit), on a width that is not a multiple of 4
    # (scalar loads), and on a tall narrow plane; few seeds, so that single floods cross the whole plane, and all maxima,
    # so that hundreds of fronts meet
    img = cases.smooth_field(*shape, 23, octaves=octaves)
    allseeds = ol.find_local_minima(img)
    ws = _seg(pkg, pkg.ENGINE_FUSED)
    set_mode = pkg._ffi.lib().ws_ctx_set_persistent_pass
    for seeds in (allseeds[:: max(len(allseeds) // 3, 1)][:3], allseeds):
        want = ol.segment_arrival(img, seeds)
        assert set_mode(ws._ctx().handle, 0) == 0      # the passes themselves (with sparse seeds the default would take the tile queue)
        got = ws.transform(img, seeds)
        assert set_mode(ws._ctx().handle, 3) == 0
        assert got.shape == want.shape and (got == want).all(), (shape, len(seeds))
        assert ws._ctx().stats()["relax_passes"] >= 8
        got = ws.transform(img, seeds)                  # ... and whatever the default picks
        assert (got == want).all(), (shape, len(seeds), "default")
    # the same through the merging transform (final labels) and with a low maximum level (large never-flooded regions)
    mg = pkg.TransformBuilder.new().set_max_water_lvl(120).build_merging()
    assert (mg.transform_final(img, allseeds) == ol.merge_arrival(img, allseeds, max_level=120)).all()


def test_long_range_stacked_slices(pkg):
    # a stack of smooth slices: hundreds of passes with slice walls inside the tiles of the one-grid passes
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    himgs = [cases.smooth_field(200, 512, 60 + k, octaves=6) for k in range(3)]
    hseeds = [ol.find_local_minima(a)[:: 5] for a in himgs]
    got = _run_batch(eng, himgs, hseeds)
    assert eng.stats()["relax_passes"] >= 8
    for k in range(3):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k


def test_sparse_seeds_take_the_tile_queue_in_stacks_low_levels_and_merging(pkg):
    # The default picks the persistent pass in flood order when seeds are sparse (fewer than one per two tiles).  Here: a stack
    # of two smooth slices (slice walls inside the queue's 128 x 128 tiles), a low maximum level (few buckets: the queue's
    # order has one or two levels per bucket, and most of the plane is never flooded), an odd-sized plane, the merging
    # transform's final labels -- every label against the oracle.
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    himgs = [cases.smooth_field(1024, 2048, 70 + k, octaves=7) for k in range(2)]
    hseeds = [np.asarray(ol.find_local_minima(a), dtype=np.uint64).reshape(-1, 2) for a in himgs]
    hseeds = [x[:: max(len(x) // 2, 1)][:2] for x in hseeds]
    got = _run_batch(eng, himgs, hseeds)
    assert eng.stats()["relax_passes"] <= 12      # (the queue stood for the passes)
    for k in range(2):
        assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k
    img = cases.smooth_field(1990, 2052, 33, octaves=7)
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    seeds = seeds[:: max(len(seeds) // 3, 1)][:3]
    for max_level in (254, 90, 20):
        ws = _seg(pkg, pkg.ENGINE_FUSED, max_level=max_level)
        got = ws.transform(img, seeds)
        assert (got == ol.segment_arrival(img, seeds, max_level=max_level)).all(), max_level
        assert ws._ctx().stats()["relax_passes"] <= 12, max_level
    mg = pkg.TransformBuilder.new().set_max_water_lvl(200).build_merging()
    assert (mg.transform_final(img, seeds) == ol.merge_arrival(img, seeds, max_level=200)).all()
    # seeds = the image's own minima (ws_segment_minima): a map with a handful of them; the second call predicts "sparse" from
    # the first one's count and takes the queue, the third replays the graph that holds it
    import torch
    few = np.full((2048, 2048), 200, dtype=np.uint8)
    yy, xx = np.mgrid[0:2048, 0:2048]
    for k, (cy, cx) in enumerate([(300, 400), (1500, 1700), (900, 1100), (1800, 200), (200, 1800)]):
        d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        few = np.minimum(few, np.clip(d / 6 + 3 * k, 0, 200)).astype(np.uint8)
    few = (253 - few).astype(np.uint8)      # five peaks: find_local_minima returns strict maxima
    for cy, cx in [(300, 400), (1500, 1700), (900, 1100), (1800, 200), (200, 1800)]:
        few[cy, cx] += 1                    # (a strict one on top of each plateau)
    fs = np.asarray(ol.find_local_minima(few), dtype=np.uint64).reshape(-1, 2)
    assert 1 <= len(fs) <= 16
    want = ol.segment_arrival(few, fs)
    d_img = torch.from_numpy(few).to(eng.device)
    out = torch.empty((2048, 2048), dtype=torch.int32, device=eng.device)
    passes = []
    for _ in range(3):
        labels, n = eng.segment_minima(d_img, out=out)
        assert n == len(fs) and (labels.cpu().numpy().view(np.uint32) == want).all()
        passes.append(eng.stats()["relax_passes"])
    assert passes[1] <= 12 and passes[2] <= 12, passes


def test_level_snapshots_on_the_device_match_transform_history(pkg):
    # ws_level_snapshot_device: the plane transform_history reports for a level, without leaving the device
    import torch
    eng = _torch_engine(pkg)
    himg = cases.smooth_field(150, 212, 8)
    hseeds = ol.find_local_minima(himg)
    levels = {}
    ol.segment(himg, hseeds, hook=lambda lvl, mx, im, lab: levels.__setitem__(lvl, lab.copy()))
    img = torch.from_numpy(himg).to(eng.device)
    seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(eng.device).contiguous()
    labels = eng.segment(img, seeds)
    for lvl in (0, 1, 37, 128, 200, 254):
        got = eng.level_snapshot(labels, lvl).cpu().numpy().view(np.uint32)
        assert (got == levels[lvl]).all(), lvl
    assert (eng.level_snapshot(labels, 254, out=labels).cpu().numpy().view(np.uint32) == levels[254]).all()      # in place


# ---- seam repair: pass 1 as bands and strips along the tile borders of pass 0 ---------------------------------------------

@pytest.mark.parametrize("shape,kind", [((96, 520), "noise"), ((257, 1028), "noise"), ((300, 772), "smooth"), ((1000, 1300), "smooth5"),
                                        ((64, 260), "noise"), ((33, 512), "noise"), ((640, 1024), "noise"), ((129, 2048), "smooth")])
def test_seam_repair_flow_on_small_planes(pkg, shape, kind):
    # planes of 2^24 pixels and more take this flow by default (test_gpu_fullsize.py, the fix-point tests above); here the
    # threshold is lowered so that the oracle can check it: widths that are and are not multiples of the 256-pixel tile,
    # heights that are not multiples of 32 (a last seam a few rows above the plane's end), a single seam row, one band of
    # strips, noise and smooth fields (floods that cross many seams), several water levels, and the merging transform
    # (whose segmenting part takes the same passes)
    if kind == "noise":
        img = cases.field(*shape, 77)
    else:
        img = cases.smooth_field(*shape, 41, octaves=5 if kind == "smooth5" else 4)
    seeds = ol.find_local_minima(img)
    ws = _seg(pkg, pkg.ENGINE_FUSED)
    ws._ctx().set_seam_repair_min_pixels(1)
    try:
        for _ in range(2):      # the second call replays a captured graph
            got = ws.transform(img, seeds)
            assert (got == ol.segment_arrival(img, seeds)).all(), (shape, kind)
            st = ws._ctx().stats()
            assert st["launches_relax"] == st["relax_passes"] + 1      # pass 1 was two launches: bands, strips
        few = seeds[:: max(len(seeds) // 3, 1)][:3]
        assert (ws.transform(img, few) == ol.segment_arrival(img, few)).all()
        lo = pkg.TransformBuilder.new().set_max_water_lvl(100).build_segmenting()
        lo._ctx().set_seam_repair_min_pixels(1)
        assert (lo.transform(img, seeds) == ol.segment_arrival(img, seeds, max_level=100)).all()
        mg = pkg.TransformBuilder.new().build_merging()
        mg._ctx().set_seam_repair_min_pixels(1)
        assert (mg.transform_final(img, seeds) == ol.merge_arrival(img, seeds)).all()
    finally:
        ws._ctx().set_seam_repair_min_pixels(0)


@pytest.mark.parametrize("s,h,w", [(3, 40, 520), (2, 64, 512), (5, 33, 772), (2, 200, 1028)])
def test_seam_repair_flow_on_stacked_slices(pkg, s, h, w):
    # a stack of slices through the seam repair: slice walls inside bands (a slice height that is no multiple of 32), on a
    # seam (64), and several to a strip slice; noise and smooth slices
    import importlib
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    eng.ctx.set_seam_repair_min_pixels(1)
    try:
        himgs, hseeds = _batch_case(s, h, w, 500 + s)
        got = _run_batch(eng, himgs, hseeds)
        for k in range(s):
            assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k].astype(np.uint64))).all(), k
        st = eng.stats()
        assert st["launches_relax"] == st["relax_passes"] + 1
        himgs = [cases.smooth_field(h, w, 90 + k, octaves=4) for k in range(s)]
        hseeds = [ol.find_local_minima(a) for a in himgs]
        got = _run_batch(eng, himgs, hseeds)
        for k in range(s):
            assert (got[k] == ol.segment_arrival(himgs[k], hseeds[k])).all(), k
    finally:
        eng.ctx.set_seam_repair_min_pixels(0)


def test_segment_begin_end_pipelined_on_two_contexts(pkg):
    # ws_segment_device_begin / _end: two engines on two streams take turns, every transform queued behind the other
    # engine's one before the host has waited for anything.  Same labels as the one-call form; a flood that needs more
    # passes than the replayed graph holds is finished inside _end; misuse is an error, not a hang.
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    for mode in ("events", "one_stream", "concurrent"):
        _pipelined_begin_end(pkg, dev, torch, mode)


def test_segment_begin_end_with_the_tile_queue_inside_the_graph(pkg):
    # sparse seeds on smooth maps: the replayed graph of _begin then holds the tile queue (pass 3), the check (pass 4) and the
    # gated resolve; two contexts on two streams, their queue launches overlap on the GPU
    import importlib
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    engines = []
    for s in (torch.cuda.Stream(0), torch.cuda.Stream(0)):
        with torch.cuda.stream(s):
            engines.append(dev.DeviceEngine(0))
    dv = engines[0].device
    ss, sw, sl = [], [], []
    for k in range(2):
        h = cases.smooth_field(1536, 2048, 80 + k, octaves=7)
        s = np.asarray(ol.find_local_minima(h), dtype=np.uint64).reshape(-1, 2)
        s = s[:: max(len(s) // 3, 1)][:3]
        ss.append((torch.from_numpy(h).to(dv), torch.from_numpy(s.astype(np.int64).astype(np.int32)).to(dv).contiguous()))
        sw.append(ol.segment_arrival(h, s))
        sl.append(torch.zeros((1536, 2048), dtype=torch.int32, device=dv))
    torch.cuda.synchronize()
    for rnd in range(4):
        for k, eng in enumerate(engines):
            eng.segment_begin(ss[k][0], ss[k][1], sl[k])
        for k, eng in enumerate(engines):
            eng.segment_end()
            assert (sl[k].cpu().numpy().view(np.uint32) == sw[k]).all(), (rnd, k)
            sl[k].zero_()
        torch.cuda.synchronize()
    st = engines[0].stats()
    assert st["graph_launches"] == 1 and st["relax_passes"] <= 12, st


def _pipelined_begin_end(pkg, dev, torch, mode):
    # "events": two streams, the next transform waits for the other context's by an event; "one_stream": stream order does
    # it, and _end must wait for its own graph only (the other context's transform is queued behind it); "concurrent": two
    # streams and no order between them -- the two contexts' transforms overlap on the GPU (what bench.py does)
    same_stream = mode == "one_stream"
    streams = [torch.cuda.Stream(0), torch.cuda.Stream(0)]
    if same_stream:
        streams[1] = streams[0]
    engines = []
    for s in streams:
        with torch.cuda.stream(s):
            engines.append(dev.DeviceEngine(0))
    for himg in (cases.field(300, 772, 5), cases.smooth_field(520, 1028, 31, octaves=5)):
        hseeds = ol.find_local_minima(himg)
        want = ol.segment_arrival(himg, hseeds)
        img = torch.from_numpy(himg).to(engines[0].device)
        seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(engines[0].device).contiguous()
        outs = [torch.zeros(himg.shape, dtype=torch.int32, device=engines[0].device) for _ in engines]
        torch.cuda.synchronize()
        events = [torch.cuda.Event(), torch.cuda.Event()]
        pending = [False, False]
        replayed = 0
        for k in range(10):
            i = k & 1
            if pending[i]:
                engines[i].segment_end()
                replayed += engines[i].stats()["graph_launches"]
                assert (outs[i].cpu().numpy().view(np.uint32) == want).all(), (k, himg.shape)
                outs[i].zero_()
                torch.cuda.synchronize()
            if k > 0 and mode == "events":
                streams[i].wait_event(events[1 - i])
            engines[i].segment_begin(img, seeds, outs[i])
            events[i].record(streams[i])
            pending[i] = True
        for i in (0, 1):
            engines[i].segment_end()
            assert (outs[i].cpu().numpy().view(np.uint32) == want).all(), ("tail", i)
        assert replayed >= 4        # from its third call on an engine replays its graph: those were left in flight
    # the CONTENTS of a replayed call's buffers are free to change: a list that is no longer sorted (the graph's tables are
    # then invalid: _end repeats the transform with painted seeds) and a seed outside the plane (_end reports it)
    perm = np.random.default_rng(3).permutation(len(hseeds))
    seeds.copy_(torch.from_numpy(hseeds[perm].astype(np.int64)).to(torch.int32))
    torch.cuda.synchronize()
    for i in (0, 1):
        engines[i].segment_begin(img, seeds, outs[i])
    for i in (0, 1):
        engines[i].segment_end()
        assert (outs[i].cpu().numpy().view(np.uint32) == ol.segment_arrival(himg, hseeds[perm])).all(), ("permuted", i)
    bad = hseeds.copy()
    bad[len(bad) // 2] = (himg.shape[0] + 5, 1)
    seeds.copy_(torch.from_numpy(bad.astype(np.int64)).to(torch.int32))
    torch.cuda.synchronize()
    engines[0].segment_begin(img, seeds, outs[0])
    with pytest.raises(pkg.SeedOutOfBounds):
        engines[0].segment_end()
    seeds.copy_(torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32))
    torch.cuda.synchronize()
    # misuse
    with pytest.raises(pkg.WatershedError):
        engines[0].segment_end()
    engines[0].segment_begin(img, seeds, outs[0])
    with pytest.raises(pkg.WatershedError):
        engines[0].segment_begin(img, seeds, outs[0])
    with pytest.raises(pkg.WatershedError):
        engines[0].segment(img, seeds, out=outs[0])
    with pytest.raises(pkg.WatershedError):
        engines[0].find_local_minima(img)         # any other work on a context that holds a transform
    # ... the row-block entry points and the context setters included: they rewrite stamps, flags, tables and work lists that
    # the transform in flight (and its _end half) still uses
    L, hnd, BAD = pkg._ffi.lib(), engines[0].ctx.handle, pkg._ffi.WS_ERR_BAD_ARG
    hh, ww = int(img.shape[0]), int(img.shape[1])
    keys = torch.empty((hh, ww), dtype=torch.int32, device=img.device)
    chg = ctypes.c_int(0)
    assert L.ws_block_begin(hnd, img.data_ptr(), hh, ww, ww, 254, seeds.data_ptr(), int(seeds.shape[0]), 1, keys.data_ptr()) == BAD
    assert L.ws_block_relax_halo(hnd, img.data_ptr(), hh, ww, ww, 254, 1, 0, keys.data_ptr()) == BAD
    assert L.ws_block_resolve_local(hnd, keys.data_ptr(), outs[1].data_ptr(), hh, ww, 1, 0) == BAD
    assert L.ws_block_init(hnd, hh, ww, seeds.data_ptr(), seeds.data_ptr(), 0, keys.data_ptr(), outs[1].data_ptr()) == BAD
    assert L.ws_block_relax(hnd, img.data_ptr(), hh, ww, ww, 254, keys.data_ptr(), ctypes.byref(chg)) == BAD
    assert L.ws_block_resolve(hnd, keys.data_ptr(), outs[1].data_ptr(), hh, ww, ctypes.byref(chg)) == BAD
    assert L.ws_block_merge_import(hnd, keys.data_ptr(), 0, keys.data_ptr()) == BAD
    assert L.ws_ctx_set_profiling(hnd, 1) == BAD
    assert L.ws_ctx_set_seam_repair_min_pixels(hnd, 4096) == BAD
    assert L.ws_ctx_set_batch_pixel_limit(hnd, 4096) == BAD
    assert L.ws_ctx_set_persistent_pass(hnd, 1) == BAD
    assert L.ws_ctx_set_live_list_min_colours(hnd, 1) == BAD
    engines[0].segment_end()
    assert (engines[0].segment(img, seeds, out=outs[0]).cpu().numpy().view(np.uint32) == want).all()
    # the merging transform in two halves: the replayed graph and the unions queued behind it are left in flight; a flood
    # that needs more passes than the graph holds (the smooth field) has its unions redone inside merge_end
    if mode == "concurrent":
        for himg in (cases.field(300, 772, 6), cases.smooth_field(520, 1028, 32, octaves=5)):
            hseeds = ol.find_local_minima(himg)
            want = ol.merge_arrival(himg, hseeds)
            img = torch.from_numpy(himg).to(engines[0].device)
            seeds = torch.from_numpy(hseeds.astype(np.int64)).to(torch.int32).to(engines[0].device).contiguous()
            outs = [torch.zeros(himg.shape, dtype=torch.int32, device=engines[0].device) for _ in engines]
            torch.cuda.synchronize()
            pending = [False, False]
            for k in range(10):
                i = k & 1
                if pending[i]:
                    engines[i].merge_end()
                    assert (outs[i].cpu().numpy().view(np.uint32) == want).all(), (k, himg.shape)
                    outs[i].zero_()
                    torch.cuda.synchronize()
                engines[i].merge_begin(img, seeds, outs[i])
                pending[i] = True
            for i in (0, 1):
                engines[i].merge_end()
                assert (outs[i].cpu().numpy().view(np.uint32) == want).all(), ("tail", i)
        engines[0].merge_begin(img, seeds, outs[0])
        with pytest.raises(pkg.WatershedError):
            engines[0].segment_end()              # the wrong end
        engines[0].merge_end()
