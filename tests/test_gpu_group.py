"""Several ranks behind one C call: ws_group_* / ws_segment_tiled(_device) / ws_segment_batch_group (csrc/ws_tiled.hip).

The box has ONE GPU, so the LOCAL groups here put 2-4 ranks on device 0 (the protocol is the same: a host thread and a
context per rank, the exchange steps as stream-ordered copies); the RCCL group runs with world == 1, which still sends the
flag word through ncclAllReduce and the tables through ncclAllGather of the real library.  Every result is compared with
the single-domain transform of the same inputs (itself oracle-checked in the other files) and, at small sizes, with the
oracle directly.  Reference seam: the one-address-space loop of lib.rs:1689-1748.
"""
import ctypes
import importlib

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


class Group:
    def __init__(self, pkg, n_ranks=None, rccl=False):
        self.ffi = pkg._ffi
        self.L = self.ffi.lib()
        self.h = ctypes.c_void_p()
        if rccl:
            uid = ctypes.create_string_buffer(self.ffi.WS_RCCL_ID_BYTES)
            rc = self.L.ws_group_rccl_unique_id(uid)
            assert rc == 0, (rc, self.L.ws_group_last_error(None))
            rc = self.L.ws_group_create_rccl(0, 0, 1, uid, ctypes.byref(self.h))
        else:
            rc = self.L.ws_group_create_local(n_ranks, None, ctypes.byref(self.h))
        assert rc == 0, (rc, self.L.ws_group_last_error(None))

    def err(self):
        return self.L.ws_group_last_error(self.h).decode()

    def info(self):
        w, n, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert self.L.ws_group_info(self.h, ctypes.byref(w), ctypes.byref(n), ctypes.byref(f)) == 0
        return w.value, n.value, f.value

    def segment_tiled(self, img, seeds, merging=False, max_level=254, edge=False, seed_shift=False, expect=0):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1, 2))
        h, w = img.shape
        e = 2 if edge else 0
        out = np.zeros((h + e, w + e), dtype=np.uint64)
        opt = self.ffi.Options(max_level, int(edge), 0, 0, int(seed_shift))
        rounds = ctypes.c_uint32(0)
        rc = self.L.ws_segment_tiled(self.h, img.ctypes.data, h, w, w, s.ctypes.data if s.size else None, s.shape[0], ctypes.byref(opt),
                                     int(merging), out.ctypes.data, ctypes.byref(rounds))
        assert rc == expect, (rc, self.err())
        return out, rounds.value

    def close(self):
        if self.h:
            self.L.ws_group_destroy(self.h)
            self.h = ctypes.c_void_p()


def test_tile_rows_partition_matches_distributed_py(pkg):
    L = pkg._ffi.lib()
    wsd = importlib.import_module("rustronomy_watershed_amd.distributed")
    for h, world in ((10, 3), (8192, 8), (7, 7), (4097, 4), (32768, 8)):
        for rank in range(world):
            v = [ctypes.c_size_t() for _ in range(4)]
            assert L.ws_tile_rows(h, rank, world, *[ctypes.byref(x) for x in v]) == 0
            assert tuple(x.value for x in v) == wsd.row_block(h, rank, world)
    v = [ctypes.c_size_t() for _ in range(4)]
    assert L.ws_tile_rows(3, 0, 4, *[ctypes.byref(x) for x in v]) == pkg._ffi.WS_ERR_BAD_ARG      # a rank without rows


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 4])
def test_local_group_selftest_and_info(pkg, n_ranks):
    g = Group(pkg, n_ranks)
    assert g.info() == (n_ranks, n_ranks, 0)
    assert g.L.ws_group_selftest(g.h) == 0, g.err()
    g.close()


def test_rccl_group_of_one_rank_runs_the_real_library(pkg):
    g = Group(pkg, rccl=True)
    assert g.info() == (1, 1, 0)
    assert g.L.ws_group_selftest(g.h) == 0, g.err()          # ncclAllReduce (max, min) and ncclAllGather with world == 1
    img = cases.field(96, 128, 5)
    seeds = ol.find_local_minima(img)
    got, _ = g.segment_tiled(img, seeds)
    assert (got == ol.segment(img, seeds)).all()
    g.close()


@pytest.mark.parametrize("n_ranks", [2, 3, 4])
@pytest.mark.parametrize("shape", [(256, 512), (131, 260), (1000, 1024), (97, 100)])
def test_tiled_host_equals_oracle_segmenting_and_merging(pkg, n_ranks, shape):
    # widths that are multiples of 4 take the fast form (seed tables, one table exchange); 97 x 100 ... also; (131, 260) too;
    # the general form is covered below (odd width, unsorted lists)
    img = cases.field(shape[0], shape[1], 17 + n_ranks)
    seeds = ol.find_local_minima(img)
    g = Group(pkg, n_ranks)
    got, rounds = g.segment_tiled(img, seeds)
    assert (got == ol.segment_arrival(img, seeds)).all()
    assert rounds >= 3                                   # the vote, at least one halo swap, the table
    gotm, _ = g.segment_tiled(img, seeds, merging=True, max_level=120)
    assert (gotm == ol.merge_arrival(img, seeds, max_level=120)).all()
    g.close()


def test_tiled_host_blocks_of_more_than_a_chunk_of_labels(pkg):
    # every rank's rows reach the caller's usize plane as u32 chunks widened by host threads of the rank's context
    # (ws_hostcopy.hip): blocks of 6.3 M and 2.1 M pixels (four chunks of 1.6 M / of 2^19 labels), all ranks at it at once
    img = cases.field(3000, 4200, 23)
    seeds = ol.find_local_minima(img)
    want = ol.segment_arrival(img, seeds)
    for n_ranks in (2, 6):
        g = Group(pkg, n_ranks)
        got, _ = g.segment_tiled(img, seeds)
        assert (got == want).all()
        g.close()


@pytest.mark.parametrize("n_ranks,rccl", [(1, False), (2, False), (3, False), (1, True)])
def test_host_cube_over_the_ranks_of_a_group(pkg, n_ranks, rccl):
    # ws_segment_batch_host: rank r pipelines its block of the slices (ws_segment_batch on its own context); with more ranks than
    # slices some ranks idle.  Every slice against the oracle -- with its own minima, with the caller's lists, with edge correction
    # -- and the lowest failing slice named (lib.rs:1675-1677).
    rng = np.random.default_rng(3)
    cube = np.stack([cases.field(100, 72, 60 + k) if k % 2 else cases.smooth_field(100, 72, 60 + k) for k in range(8)])
    g = Group(pkg, n_ranks, rccl=rccl)
    L = g.L

    def run(n, lists, edge, expect=0):
        e = 2 if edge else 0
        out = np.zeros((max(n, 1), 100 + e, 72 + e), dtype=np.uint64)
        counts = np.zeros(max(n, 1), dtype=np.uintp)
        failed = ctypes.c_size_t(77)
        opt = pkg._ffi.Options(254, int(edge))
        flat = offs = None
        if lists is not None:
            offs = np.zeros(n + 1, dtype=np.uintp)
            offs[1:] = np.cumsum([len(l) for l in lists[:n]])
            flat = np.ascontiguousarray(np.concatenate(list(lists[:n]) + [np.zeros((1, 2), dtype=np.uint64)], axis=0))
        rc = L.ws_segment_batch_host(g.h, cube.ctypes.data, n, 100, 72, 72, 100 * 72, flat.ctypes.data if flat is not None else None,
                                     offs.ctypes.data_as(pkg._ffi.szp) if offs is not None else None, ctypes.byref(opt), out.ctypes.data,
                                     counts.ctypes.data_as(pkg._ffi.szp), ctypes.byref(failed))
        assert rc == expect, (rc, g.err())
        return out, counts, failed.value

    for edge in (False, True):
        for n in (8, 5, 1, 0):
            out, counts, _ = run(n, None, edge)
            for k in range(n):
                seeds = ol.find_local_minima(cube[k])
                assert counts[k] == len(seeds) and (out[k] == ol.segment(cube[k], seeds, edge=edge)).all(), (n, k)
    lists = []
    for k in range(8):
        s = np.asarray(ol.find_local_minima(cube[k]), dtype=np.uint64).reshape(-1, 2)
        lists.append(s[rng.permutation(len(s))[: len(s) // (k + 1)]] if k != 4 else s[:0])
    out, counts, _ = run(8, lists, False)
    for k in range(8):
        assert counts[k] == len(lists[k]) and (out[k] == ol.segment(cube[k], lists[k])).all(), k
    for bad_slices in ((6, 3), (7,), (0, 5)):
        bad = [l.copy() for l in lists]
        for k in bad_slices:
            bad[k] = np.array([[2, 2], [3000 + k, 1]], dtype=np.uint64)
        _, _, failed = run(8, bad, False, expect=pkg._ffi.WS_ERR_SEED_OOB)
        assert failed == min(bad_slices), (bad_slices, failed)
    out, _, _ = run(8, lists, False)      # the group works on
    assert (out[7] == ol.segment(cube[7], lists[7])).all()
    g.close()


def test_tiled_smooth_field_floods_cross_several_blocks(pkg):
    # few seeds, long floods: chains cross the seams many times and in both directions
    img = cases.smooth_field(600, 512, 3, octaves=5)
    seeds = ol.find_local_minima(img)
    g = Group(pkg, 4)
    got, rounds = g.segment_tiled(img, seeds)
    assert (got == ol.segment_arrival(img, seeds)).all()
    gotm, _ = g.segment_tiled(img, seeds, merging=True)
    assert (gotm == ol.merge_arrival(img, seeds)).all()
    g.close()


def test_tiled_general_form_odd_width_unsorted_and_duplicate_seeds(pkg):
    rng = np.random.default_rng(3)
    g = Group(pkg, 3)
    img = cases.field(150, 203, 8)                       # w % 4 != 0: every rank votes for the general form
    seeds = ol.find_local_minima(img)
    got, _ = g.segment_tiled(img, seeds)
    assert (got == ol.segment(img, seeds)).all()
    img = cases.field(160, 256, 9)
    seeds = np.asarray(ol.find_local_minima(img))
    shuffled = seeds[rng.permutation(len(seeds))]        # rows not sorted: seeds dealt out one by one with explicit colours
    got, _ = g.segment_tiled(img, shuffled)
    assert (got == ol.segment(img, shuffled)).all()
    dup = np.concatenate([seeds, seeds[::7]])            # rows sorted within each half only, and duplicates: later entries win
    got, _ = g.segment_tiled(img, dup)
    assert (got == ol.segment(img, dup)).all()
    samerow = seeds.copy()
    samerow[5], samerow[6] = seeds[6].copy(), seeds[5].copy()      # rows still sorted, list not strictly increasing: ws_block_begin refuses, vote
    if samerow[5][0] == samerow[6][0]:
        got, _ = g.segment_tiled(img, samerow)
        assert (got == ol.segment(img, samerow)).all()
    g.close()


def test_tiled_edge_correction_and_seed_shift(pkg):
    img = cases.field(120, 128, 12)
    seeds = ol.find_local_minima(img)
    g = Group(pkg, 3)
    got, _ = g.segment_tiled(img, seeds, edge=True)
    assert got.shape == (122, 130)
    assert (got == ol.segment(img, seeds, edge=True)).all()           # lib.rs:1675-1677: seeds index the padded plane unshifted
    ws = pkg.TransformBuilder.new().enable_edge_correction().shift_seeds_into_padded_plane().build_segmenting() \
        if hasattr(pkg.TransformBuilder, "shift_seeds_into_padded_plane") else None
    got2, _ = g.segment_tiled(img, seeds, edge=True, seed_shift=True)
    moved = np.asarray(seeds, dtype=np.int64) + 1
    pad = np.zeros((122, 130), np.uint8)
    pad[1:-1, 1:-1] = img
    assert (got2 == ol.segment(pad, moved)).all()
    if ws is not None:
        assert (got2 == ws.transform(img, seeds)).all()
    g.close()


def test_tiled_errors(pkg):
    ffi = pkg._ffi
    g = Group(pkg, 4)
    img = cases.field(64, 64, 1)
    seeds = ol.find_local_minima(img)
    bad = np.concatenate([np.asarray(seeds), [[64, 3]]])
    g.segment_tiled(img, bad, expect=ffi.WS_ERR_SEED_OOB)            # the reference panics: lib.rs:1676
    g.segment_tiled(img[:3], np.zeros((0, 2)), expect=ffi.WS_ERR_BAD_ARG)      # fewer rows than ranks
    out, _ = g.segment_tiled(img, seeds)                               # the group is usable after an error
    assert (out == ol.segment(img, seeds)).all()
    opt = ffi.Options(0)
    rc = g.L.ws_segment_tiled(g.h, img.ctypes.data, 64, 64, 64, None, 0, ctypes.byref(opt), 0, out.ctypes.data, None)
    assert rc == ffi.WS_ERR_MAX_TOO_LOW
    g.close()


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_tiled_device_8192_equals_single_domain(pkg, n_ranks):
    # the headline field in row blocks, device resident, against the single-domain transform (oracle-checked at this
    # size in test_gpu_oracle_at_size.py): every owned row, segmenting and merging
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    H = W = 8192
    img = eng.random_field(H, W, 1)
    seeds = eng.find_local_minima(img).clone()
    want = eng.segment(img, seeds)
    want_m = eng.merge(img, seeds, max_level=150)
    torch.cuda.synchronize()
    g = Group(pkg, n_ranks)
    L, ffi = g.L, pkg._ffi
    rows = seeds[:, 0].contiguous()
    blocks = (ffi.TileBlock * n_ranks)()
    keep = []
    spans = []
    for r in range(n_ranks):
        v = [ctypes.c_size_t() for _ in range(4)]
        assert L.ws_tile_rows(H, r, n_ranks, *[ctypes.byref(x) for x in v]) == 0
        r0, r1, lo, hi = (x.value for x in v)
        b = torch.searchsorted(rows, torch.tensor([lo, hi], dtype=rows.dtype, device=rows.device))
        i0, i1 = int(b[0]), int(b[1])
        loc = seeds[i0:i1].clone()
        loc[:, 0] -= lo
        bi = img[lo:hi].contiguous()
        lab = torch.empty((hi - lo, W), dtype=torch.int32, device=eng.device)
        keep += [loc, bi, lab]
        spans.append((r0, r1, lo, lab))
        blocks[r] = ffi.TileBlock(bi.data_ptr(), loc.data_ptr(), None, i1 - i0, i0 + 1, 0, lab.data_ptr())
    torch.cuda.synchronize()
    opt = ffi.Options(254)
    rounds = ctypes.c_uint32(0)
    rc = L.ws_segment_tiled_device(g.h, H, W, int(seeds.shape[0]), blocks, ctypes.byref(opt), 0, ctypes.byref(rounds))
    assert rc == 0, (rc, g.err())
    for r0, r1, lo, lab in spans:
        assert bool((lab[r0 - lo:r1 - lo] == want[r0:r1]).all())
    assert 3 <= rounds.value <= 12
    opt = ffi.Options(150)
    rc = L.ws_segment_tiled_device(g.h, H, W, int(seeds.shape[0]), blocks, ctypes.byref(opt), 1, None)
    assert rc == 0, (rc, g.err())
    for r0, r1, lo, lab in spans:
        assert bool((lab[r0 - lo:r1 - lo] == want_m[r0:r1]).all())
    g.close()


def _blocks_of(pkg, g, n_ranks, img, seeds, eng):
    import torch
    L, ffi = g.L, pkg._ffi
    H, W = img.shape
    rows = seeds[:, 0].contiguous()
    blocks = (ffi.TileBlock * n_ranks)()
    keep, spans = [], []
    for r in range(n_ranks):
        v = [ctypes.c_size_t() for _ in range(4)]
        assert L.ws_tile_rows(H, r, n_ranks, *[ctypes.byref(x) for x in v]) == 0
        r0, r1, lo, hi = (x.value for x in v)
        b = torch.searchsorted(rows, torch.tensor([lo, hi], dtype=rows.dtype, device=rows.device))
        i0, i1 = int(b[0]), int(b[1])
        loc = seeds[i0:i1].clone()
        loc[:, 0] -= lo
        bi = img[lo:hi].contiguous()
        lab = torch.empty((hi - lo, W), dtype=torch.int32, device=eng.device)
        keep += [loc, bi, lab]
        spans.append((r0, r1, lo, lab))
        blocks[r] = ffi.TileBlock(bi.data_ptr(), loc.data_ptr() if i1 > i0 else None, None, i1 - i0, i0 + 1, 0, lab.data_ptr())
    torch.cuda.synchronize()
    return blocks, spans, keep


@pytest.mark.parametrize("n_ranks,rccl", [(1, False), (2, False), (3, False), (4, False), (1, True)])
@pytest.mark.parametrize("merging", [True, False])
def test_transform_to_list_of_a_field_in_row_blocks(pkg, n_ranks, rccl, merging):
    # ws_transform_to_list_tiled_device: the flood on all ranks, the stamps and labels of the owned rows gathered on rank 0, the
    # records of every level from the whole plane there (ws_lists_from_arrival_device).  Lake sizes of every level against the
    # oracle's (lib.rs:628-635), on a noise field and on a smooth one whose lakes span all blocks.
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    ffi = pkg._ffi
    for kind in ("noise", "smooth"):
        H, W = (300, 256) if kind == "noise" else (280, 192)
        himg = cases.field(H, W, 31) if kind == "noise" else cases.smooth_field(H, W, 12, octaves=4)
        hseeds = np.asarray(ol.find_local_minima(himg), dtype=np.int64).reshape(-1, 2)
        want = {}
        run = ol.merge_arrival if merging else ol.segment      # (the segmenting oracle with a hook is the literal sweep form)
        run(himg, hseeds, hook=lambda l, m, i, c: want.__setitem__(l, ol.find_lake_sizes(c)))
        img = torch.from_numpy(himg).to(eng.device)
        seeds = torch.from_numpy(hseeds.astype(np.int32)).to(eng.device)
        g = Group(pkg, n_ranks, rccl=rccl)
        blocks, spans, keep = _blocks_of(pkg, g, n_ranks, img, seeds, eng)
        cap = 255 * (len(hseeds) + 1)      # every colour alive at every level: the most there can be
        lakes = torch.zeros((cap, 2), dtype=torch.int64, device=eng.device)
        n_lakes = ctypes.c_size_t(0)
        offsets = np.zeros(256, dtype=np.uint64)
        uncol = np.zeros(255, dtype=np.uint64)
        opt = ffi.Options(254)
        rounds = ctypes.c_uint32(0)
        rc = g.L.ws_transform_to_list_tiled_device(g.h, H, W, len(hseeds), blocks, ctypes.byref(opt), int(merging), lakes.data_ptr(), cap, ctypes.byref(n_lakes),
                                                   offsets.ctypes.data, uncol.ctypes.data, ctypes.byref(rounds))
        assert rc == 0, (rc, g.err())
        assert n_lakes.value == offsets[255] <= cap
        rec = lakes.cpu().numpy()
        for lvl in range(255):
            dense = np.zeros(H * W + 1, dtype=np.uint64)
            part = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
            dense[part[:, 0]] = part[:, 1].astype(np.uint64)
            dense[0] = uncol[lvl]
            assert (dense == want[lvl]).all(), (kind, lvl)
        # the blocks hold the segmenting labels, as after ws_segment_tiled_device
        seg = ol.segment_arrival(himg, hseeds)
        for r0, r1, lo, lab in spans:
            assert (lab[r0 - lo:r1 - lo].cpu().numpy().view(np.uint32) == seg[r0:r1]).all()
        # a record buffer that is too small: the count is still reported
        rc = g.L.ws_transform_to_list_tiled_device(g.h, H, W, len(hseeds), blocks, ctypes.byref(opt), int(merging), lakes.data_ptr(), 10, ctypes.byref(n_lakes),
                                                   offsets.ctypes.data, uncol.ctypes.data, None)
        assert rc == ffi.WS_ERR_CAPACITY and n_lakes.value == offsets[255] > 10
        g.close()


def test_transform_to_list_of_a_field_in_row_blocks_with_any_seed_list(pkg):
    # the general form of the tiled transform (a list in any order: explicit colours per block) under the lists
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    ffi = pkg._ffi
    H, W, n_ranks = 150, 131, 3
    himg = cases.field(H, W, 77)
    rng = np.random.default_rng(4)
    hseeds = np.asarray(ol.find_local_minima(himg), dtype=np.int64).reshape(-1, 2)
    hseeds = hseeds[rng.permutation(len(hseeds))][: len(hseeds) // 2]
    want = {}
    ol.merge_arrival(himg, hseeds, hook=lambda l, m, i, c: want.__setitem__(l, ol.find_lake_sizes(c)))
    img = torch.from_numpy(himg).to(eng.device)
    g = Group(pkg, n_ranks)
    blocks = (ffi.TileBlock * n_ranks)()
    keep = []
    for r in range(n_ranks):
        v = [ctypes.c_size_t() for _ in range(4)]
        assert g.L.ws_tile_rows(H, r, n_ranks, *[ctypes.byref(x) for x in v]) == 0
        r0, r1, lo, hi = (x.value for x in v)
        inside = (hseeds[:, 0] >= lo) & (hseeds[:, 0] < hi)
        loc = hseeds[inside].copy()
        loc[:, 0] -= lo
        col = (np.nonzero(inside)[0] + 1).astype(np.int32)
        t_loc = torch.from_numpy(loc.astype(np.int32)).to(eng.device).contiguous()
        t_col = torch.from_numpy(col).to(eng.device).contiguous()
        bi = img[lo:hi].contiguous()
        lab = torch.empty((hi - lo, W), dtype=torch.int32, device=eng.device)
        keep += [t_loc, t_col, bi, lab]
        blocks[r] = ffi.TileBlock(bi.data_ptr(), t_loc.data_ptr(), t_col.data_ptr(), len(loc), 0, 0, lab.data_ptr())
    torch.cuda.synchronize()
    cap = 255 * (len(hseeds) + 1)
    lakes = torch.zeros((cap, 2), dtype=torch.int64, device=eng.device)
    n_lakes = ctypes.c_size_t(0)
    offsets, uncol = np.zeros(256, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
    opt = ffi.Options(254)
    rc = g.L.ws_transform_to_list_tiled_device(g.h, H, W, len(hseeds), blocks, ctypes.byref(opt), 1, lakes.data_ptr(), cap, ctypes.byref(n_lakes),
                                               offsets.ctypes.data, uncol.ctypes.data, None)
    assert rc == 0, (rc, g.err())
    rec = lakes.cpu().numpy()
    for lvl in range(255):
        dense = np.zeros(H * W + 1, dtype=np.uint64)
        part = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
        dense[part[:, 0]] = part[:, 1].astype(np.uint64)
        dense[0] = uncol[lvl]
        assert (dense == want[lvl]).all(), lvl
    g.close()


@pytest.mark.parametrize("n_ranks", [1, 2, 4])
@pytest.mark.parametrize("edge", [False, True])
def test_transform_to_list_of_a_host_field_over_a_group(pkg, n_ranks, edge):
    # ws_transform_to_list_tiled: host image and (usize, usize) seeds in, lake records out -- what a Rust caller of transform_to_list
    # (lib.rs:1551-1561) gets from several GPUs.  Every level's lake sizes against the oracle's, with edge correction (the padded
    # plane's lists, as ws_transform_to_list's), a list in any order, and a record buffer that is too small.
    ffi = pkg._ffi
    rng = np.random.default_rng(n_ranks)
    himg = cases.field(97, 120, 5)
    hseeds = np.asarray(ol.find_local_minima(himg), dtype=np.uint64).reshape(-1, 2)
    g = Group(pkg, n_ranks)
    for lists in (hseeds, hseeds[rng.permutation(len(hseeds))][: len(hseeds) // 3]):
        lists = np.ascontiguousarray(lists)
        for merging in (1, 0):
            want = {}
            (ol.merge_arrival if merging else ol.segment)(himg, lists, edge=edge, hook=lambda l, m, i, c: want.__setitem__(l, ol.find_lake_sizes(c)))
            cap = 255 * (len(lists) + 1)
            rec = np.zeros((cap, 2), dtype=np.uint64)
            n_lakes = ctypes.c_size_t(0)
            offsets, uncol = np.zeros(256, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
            opt = ffi.Options(254, int(edge))
            rc = g.L.ws_transform_to_list_tiled(g.h, merging, himg.ctypes.data, 97, 120, 120, lists.ctypes.data, len(lists), ctypes.byref(opt),
                                                rec.ctypes.data, cap, ctypes.byref(n_lakes), offsets.ctypes.data, uncol.ctypes.data, None)
            assert rc == 0, (rc, g.err())
            e = 2 if edge else 0
            npx = (97 + e) * (120 + e)
            for lvl in range(255):
                dense = np.zeros(npx + 1, dtype=np.uint64)
                part = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
                dense[part[:, 0].astype(np.int64)] = part[:, 1]
                dense[0] = uncol[lvl]
                assert (dense == want[lvl]).all(), (merging, lvl)
            total = n_lakes.value
            rc = g.L.ws_transform_to_list_tiled(g.h, merging, himg.ctypes.data, 97, 120, 120, lists.ctypes.data, len(lists), ctypes.byref(opt),
                                                rec.ctypes.data, 7, ctypes.byref(n_lakes), offsets.ctypes.data, uncol.ctypes.data, None)
            assert rc == ffi.WS_ERR_CAPACITY and n_lakes.value == total
    g.close()


def test_host_lists_of_a_tiled_field_with_millions_of_records(pkg):
    # from two million records on, rank 0's records reach the caller as u32 words widened by host threads, 8 M records a piece
    # (ws_tiled.hip): 15 M records here -- two pieces -- against the one-context host call, level by level
    ffi = pkg._ffi
    himg = ol.random_field(1200, 1400, 3)
    hseeds = np.ascontiguousarray(np.asarray(ol.find_local_minima(himg), dtype=np.uint64).reshape(-1, 2))
    L = ffi.lib()
    ws = pkg.api.TransformBuilder().build_merging()
    c = ws._ctx()
    cap = 17_000_000
    a, b = np.zeros((cap, 2), dtype=np.uint64), np.zeros((cap, 2), dtype=np.uint64)
    na, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    oa, ob = np.zeros(256, dtype=np.uint64), np.zeros(256, dtype=np.uint64)
    ua, ub = np.zeros(255, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
    opt = ffi.Options(254)
    assert L.ws_transform_to_list(c.handle, 1, himg.ctypes.data, 1200, 1400, 1400, hseeds.ctypes.data, len(hseeds), ctypes.byref(opt), a.ctypes.data, cap,
                                  ctypes.byref(na), oa.ctypes.data, ua.ctypes.data) == 0
    g = Group(pkg, 2)
    rc = g.L.ws_transform_to_list_tiled(g.h, 1, himg.ctypes.data, 1200, 1400, 1400, hseeds.ctypes.data, len(hseeds), ctypes.byref(opt), b.ctypes.data, cap,
                                        ctypes.byref(nb), ob.ctypes.data, ub.ctypes.data, None)
    assert rc == 0, (rc, g.err())
    assert na.value == nb.value > (1 << 23) and (oa == ob).all() and (ua == ub).all()
    for lvl in range(255):
        pa, pb = a[int(oa[lvl]):int(oa[lvl + 1])], b[int(ob[lvl]):int(ob[lvl + 1])]
        assert (pa[np.argsort(pa[:, 0])] == pb[np.argsort(pb[:, 0])]).all(), lvl
    g.close()


def test_lists_from_the_arrival_planes_of_a_finished_transform(pkg):
    # ws_lists_from_arrival_device: transform_to_list without a second flood, from ws_last_arrival_device's stamps and the labels --
    # the same records as ws_transform_to_list_device on image and seeds, merging and segmenting; a padded plane as it stands
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    ffi, L = pkg._ffi, pkg._ffi.lib()
    for edge in (False, True):
        himg = cases.field(210, 330, 8)
        hseeds = np.asarray(ol.find_local_minima(himg), dtype=np.int64).reshape(-1, 2)
        img = torch.from_numpy(himg).to(eng.device)
        seeds = torch.from_numpy(hseeds.astype(np.int32)).to(eng.device)
        labels = eng.segment(img, seeds, edge=edge)
        keys = eng.last_arrival().clone()
        ph, pw = labels.shape
        for merging in (1, 0):
            cap = 255 * (len(hseeds) + 1)
            a = torch.zeros((cap, 2), dtype=torch.int64, device=eng.device)
            b = torch.zeros((cap, 2), dtype=torch.int64, device=eng.device)
            na, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
            oa, ob = np.zeros(256, dtype=np.uint64), np.zeros(256, dtype=np.uint64)
            ua, ub = np.zeros(255, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
            opt = ffi.Options(254, int(edge))
            assert L.ws_transform_to_list_device(eng.ctx.handle, merging, img.data_ptr(), 210, 330, 330, seeds.data_ptr(), len(hseeds), ctypes.byref(opt),
                                                 a.data_ptr(), cap, ctypes.byref(na), oa.ctypes.data, ua.ctypes.data) == 0
            assert L.ws_lists_from_arrival_device(eng.ctx.handle, merging, keys.data_ptr(), labels.data_ptr(), ph, pw, len(hseeds), ctypes.byref(opt),
                                                  b.data_ptr(), cap, ctypes.byref(nb), ob.ctypes.data, ub.ctypes.data) == 0
            assert na.value == nb.value and (oa == ob).all() and (ua == ub).all()
            ra, rb = a[:na.value].cpu().numpy(), b[:nb.value].cpu().numpy()
            for lvl in range(255):      # (the records of a level come in the order their lakes' waves wrote them)
                pa, pb = ra[int(oa[lvl]):int(oa[lvl + 1])], rb[int(ob[lvl]):int(ob[lvl + 1])]
                assert (pa[np.argsort(pa[:, 0])] == pb[np.argsort(pb[:, 0])]).all(), (edge, merging, lvl)
    n0 = ctypes.c_size_t(5)
    assert L.ws_lists_from_arrival_device(eng.ctx.handle, 1, None, None, 0, 7, 0, ctypes.byref(ffi.Options(254)), None, 0, ctypes.byref(n0), oa.ctypes.data, ua.ctypes.data) == 0
    assert n0.value == 0 and not oa.any()


def test_batch_group_equals_slice_by_slice(pkg):
    import torch
    dev = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev.DeviceEngine(0)
    S, H, W, n_ranks = 6, 512, 512, 3
    g = Group(pkg, n_ranks)
    ffi, L = pkg._ffi, g.L
    parts = (ffi.BatchPart * n_ranks)()
    keep, want, outs = [], {}, []
    for r in range(n_ranks):
        mine = list(range(r, S, n_ranks))                  # slice i -> rank i % world
        cube = torch.empty((len(mine), H, W), dtype=torch.uint8, device=eng.device)
        sl, offs = [], [0]
        for j, k in enumerate(mine):
            cube[j] = eng.random_field(H, W, 40 + k)
            s = eng.find_local_minima(cube[j]).clone()
            want[k] = eng.segment(cube[j], s).clone()
            sl.append(s)
            offs.append(offs[-1] + int(s.shape[0]))
        allseeds = torch.cat(sl).contiguous()
        lab = torch.empty((len(mine), H, W), dtype=torch.int32, device=eng.device)
        co = (ctypes.c_size_t * len(offs))(*offs)
        keep += [cube, allseeds, lab, co]
        outs.append((mine, lab))
        parts[r] = ffi.BatchPart(cube.data_ptr(), allseeds.data_ptr(), co, len(mine), lab.data_ptr())
    torch.cuda.synchronize()
    opt = ffi.Options(254)
    fr, fs = ctypes.c_size_t(), ctypes.c_size_t()
    rc = L.ws_segment_batch_group(g.h, H, W, parts, ctypes.byref(opt), ctypes.byref(fr), ctypes.byref(fs))
    assert rc == 0, (rc, g.err())
    for mine, lab in outs:
        for j, k in enumerate(mine):
            assert bool((lab[j] == want[k]).all()), k
    g.close()


# ---- 2-D tiles (BASELINE config 5's wording): the field cut in both directions -------------------------------------------------

def _tiled2d(pkg, img, seeds, py, px, max_level=254, merging=False):
    import torch
    grp_mod = importlib.import_module("rustronomy_watershed_amd.group")
    g = grp_mod.Group.local(py * px)
    dev = torch.device("cuda", 0)
    field = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
    s = torch.from_numpy(np.asarray(seeds, dtype=np.int64).reshape(-1, 2).astype(np.int32)).to(dev)
    blocks, spans, keep = g.make_blocks2d(field, s, py, px)
    rounds = g.segment_tiled2d_device(img.shape[0], img.shape[1], py, px, blocks, max_level=max_level, n_seeds_total=int(s.shape[0]), merging=merging)
    out = np.zeros(img.shape, dtype=np.uint32)
    halo_ok = True
    for (r0, r1, lo, hi), (c0, c1, clo, chi), lab in spans:
        L = lab.cpu().numpy().view(np.uint32)
        out[r0:r1, c0:c1] = L[r0 - lo:r1 - lo, c0 - clo:c1 - clo]
    # every tile's halo ring holds the owner's labels (the four corner cells of the ring belong to nobody's stencil)
    for (r0, r1, lo, hi), (c0, c1, clo, chi), lab in spans:
        L = lab.cpu().numpy().view(np.uint32)
        if lo < r0:
            halo_ok &= bool((L[0, c0 - clo:c1 - clo] == out[lo, c0:c1]).all())
        if hi > r1:
            halo_ok &= bool((L[-1, c0 - clo:c1 - clo] == out[hi - 1, c0:c1]).all())
        if clo < c0:
            halo_ok &= bool((L[r0 - lo:r1 - lo, 0] == out[r0:r1, clo]).all())
        if chi > c1:
            halo_ok &= bool((L[r0 - lo:r1 - lo, -1] == out[r0:r1, chi - 1]).all())
    g.close()
    return out, rounds, halo_ok


@pytest.mark.parametrize("py,px", [(2, 2), (2, 3), (3, 2), (1, 4), (4, 1)])
@pytest.mark.parametrize("kind", ["noise", "smooth", "few"])
def test_field_in_2d_tiles_equals_the_single_domain_transform(pkg, py, px, kind):
    # ws_segment_tiled2d_device on a local group: halo rows AND columns, the general form's block steps on every tile
    shape = (301, 422) if kind == "noise" else (260, 517)
    img = cases.field(*shape, 41) if kind == "noise" else cases.smooth_field(*shape, 13, octaves=5)
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    if kind == "few":
        seeds = seeds[:: max(len(seeds) // 4, 1)][:4]      # floods that cross every tile boundary, several times
    else:
        seeds = seeds[np.random.default_rng(3).permutation(len(seeds))]      # any order: colours are indices of THIS list
    want = ol.segment_arrival(img, seeds)
    got, rounds, halo_ok = _tiled2d(pkg, img, seeds, py, px)
    assert (got == want).all(), (py, px, kind, int((got != want).sum()))
    assert halo_ok and rounds >= 2


@pytest.mark.parametrize("py,px", [(2, 2), (3, 2), (1, 3)])
@pytest.mark.parametrize("max_level", [254, 120, 60])
def test_merging_final_labels_in_2d_tiles(pkg, py, px, max_level):
    # lakes that span several tiles, in pieces that are disconnected inside a tile: one gather of (colour, local root) pairs of
    # every tile's outermost rows and columns joins them (lib.rs:1328-1522 after the last level, against the oracle)
    img = cases.smooth_field(290, 417, 19, octaves=5)
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    want = ol.merge_arrival(img, seeds, max_level=max_level)
    got, _, halo_ok = _tiled2d(pkg, img, seeds, py, px, max_level=max_level, merging=True)
    assert (got == want).all(), (py, px, max_level, int((got != want).sum()))
    rnd = cases.field(200, 263, 8)
    rs = np.asarray(ol.find_local_minima(rnd), dtype=np.uint64).reshape(-1, 2)
    got, _, _ = _tiled2d(pkg, rnd, rs, py, px, max_level=max_level, merging=True)
    assert (got == ol.merge_arrival(rnd, rs, max_level=max_level)).all()


def test_2d_tiles_argument_errors_and_low_levels(pkg):
    import torch
    grp_mod = importlib.import_module("rustronomy_watershed_amd.group")
    g = grp_mod.Group.local(4)
    field = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    s = torch.zeros((0, 2), dtype=torch.int32, device="cuda")
    blocks, _, _keep = g.make_blocks2d(field, s, 2, 2)
    with pytest.raises(RuntimeError):
        g.segment_tiled2d_device(64, 64, 3, 2, blocks)          # 3 x 2 tiles on a group of four ranks
    with pytest.raises(ValueError):
        grp_mod.Group.tile_grid(1, 64, 0, 2, 2)                 # fewer rows than tile rows
    g.close()
    img = cases.smooth_field(200, 333, 5, octaves=5)
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    got, _, halo_ok = _tiled2d(pkg, img, seeds, 2, 2, max_level=90)
    assert halo_ok and (got == ol.segment_arrival(img, seeds, max_level=90)).all()


def test_2d_tiles_at_size_and_through_an_rccl_group_of_one(pkg):
    import torch
    grp_mod = importlib.import_module("rustronomy_watershed_amd.group")
    dev_mod = importlib.import_module("rustronomy_watershed_amd.device")
    eng = dev_mod.DeviceEngine(0)
    # 2048 x 3072 random field in 2 x 3 tiles against the single-domain transform (oracle-checked at size elsewhere)
    img = eng.random_field(2048, 3072, 17)
    seeds = eng.find_local_minima(img)
    want = eng.segment(img, seeds).clone()
    g = grp_mod.Group.local(6)
    blocks, spans, keep = g.make_blocks2d(img, seeds, 2, 3)
    g.segment_tiled2d_device(2048, 3072, 2, 3, blocks, n_seeds_total=int(seeds.shape[0]))
    for (r0, r1, lo, hi), (c0, c1, clo, chi), lab in spans:
        assert bool((lab[r0 - lo:r1 - lo, c0 - clo:c1 - clo] == want[r0:r1, c0:c1]).all())
    g.close()
    # a group of ONE rank made by RCCL: 1 x 1 tiles -- no neighbour, but the flag goes through ncclAllReduce of the real library
    g = grp_mod.Group.rccl(0, 0, 1, lambda raw: raw)
    small = eng.random_field(256, 384, 3)
    ss = eng.find_local_minima(small)
    blocks, spans, keep = g.make_blocks2d(small, ss, 1, 1)
    g.segment_tiled2d_device(256, 384, 1, 1, blocks, n_seeds_total=int(ss.shape[0]))
    assert bool((spans[0][2] == eng.segment(small, ss)).all())
    g.close()


@pytest.mark.parametrize("py,px", [(2, 2), (2, 3), (1, 4), (3, 1)])
@pytest.mark.parametrize("merging", [True, False])
def test_transform_to_list_of_a_field_in_2d_tiles(pkg, py, px, merging):
    # ws_transform_to_list_tiled2d_device: the flood in py x px tiles, the owned rectangles of stamps and labels gathered on rank 0,
    # the records of all levels from the whole plane there.  Lake sizes of every level against the oracle's.
    import torch
    grp_mod = importlib.import_module("rustronomy_watershed_amd.group")
    ffi = pkg._ffi
    dev = torch.device("cuda", 0)
    for kind in ("noise", "smooth"):
        H, W = (190, 230) if kind == "noise" else (170, 150)
        himg = cases.field(H, W, 13) if kind == "noise" else cases.smooth_field(H, W, 5, octaves=4)
        hseeds = np.asarray(ol.find_local_minima(himg), dtype=np.int64).reshape(-1, 2)
        want = {}
        (ol.merge_arrival if merging else ol.segment)(himg, hseeds, hook=lambda l, m, i, c: want.__setitem__(l, ol.find_lake_sizes(c)))
        g = grp_mod.Group.local(py * px)
        field = torch.from_numpy(himg).to(dev)
        s = torch.from_numpy(hseeds.astype(np.int32)).to(dev)
        blocks, spans, keep = g.make_blocks2d(field, s, py, px)
        cap = 255 * (len(hseeds) + 1)
        lakes = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
        n_lakes = ctypes.c_size_t(0)
        offsets, uncol = np.zeros(256, dtype=np.uint64), np.zeros(255, dtype=np.uint64)
        opt = ffi.Options(254)
        rc = ffi.lib().ws_transform_to_list_tiled2d_device(g._h, H, W, py, px, len(hseeds), blocks, ctypes.byref(opt), int(merging), lakes.data_ptr(), cap,
                                                           ctypes.byref(n_lakes), offsets.ctypes.data, uncol.ctypes.data, None)
        assert rc == 0, (rc, ffi.lib().ws_group_last_error(g._h))
        rec = lakes.cpu().numpy()
        for lvl in range(255):
            dense = np.zeros(H * W + 1, dtype=np.uint64)
            part = rec[int(offsets[lvl]):int(offsets[lvl + 1])]
            dense[part[:, 0]] = part[:, 1].astype(np.uint64)
            dense[0] = uncol[lvl]
            assert (dense == want[lvl]).all(), (kind, lvl)
        g.close()


def test_2d_tiles_host_planes_of_more_than_two_million_pixels_a_tile(pkg):
    # ws_segment_tiled2d with host buffers: a tile's owned rectangle is packed on the device, crosses the bus as u32 chunks and is
    # widened into its rows of the caller's usize plane by host threads (ws_hostcopy.hip: rows of the rectangle to rows of the
    # plane's pitch).  3000 x 2900 in 2 x 2 tiles (2.2 M pixels each) with edge correction, against the one-context host call.
    ffi = pkg._ffi
    L = ffi.lib()
    himg = ol.random_field(3000, 2900, 9)
    hseeds = np.ascontiguousarray(np.asarray(ol.find_local_minima(himg), dtype=np.uint64).reshape(-1, 2))
    for edge in (False, True):
        e = 2 if edge else 0
        opt = ffi.Options(254, int(edge))
        ws = pkg.api.TransformBuilder().build_segmenting()
        want = np.zeros((3000 + e, 2900 + e), dtype=np.uint64)
        assert L.ws_segment(ws._ctx().handle, himg.ctypes.data, 3000, 2900, 2900, hseeds.ctypes.data, len(hseeds), ctypes.byref(opt), want.ctypes.data) == 0
        g = Group(pkg, 4)
        got = np.full((3000 + e, 2900 + e), 7, dtype=np.uint64)
        rc = g.L.ws_segment_tiled2d(g.h, himg.ctypes.data, 3000, 2900, 2900, hseeds.ctypes.data, len(hseeds), ctypes.byref(opt), 2, 2, 0, got.ctypes.data, None)
        assert rc == 0, (rc, g.err())
        assert (got == want).all()
        g.close()


@pytest.mark.parametrize("shape,py,px", [((2, 2), 2, 2), ((4, 4), 2, 2), ((3, 9), 1, 3), ((9, 3), 3, 1), ((5, 7), 2, 3), ((16, 1), 4, 1), ((1, 12), 1, 4), ((40, 33), 3, 2)])
def test_2d_tiles_of_a_pixel_or_two(pkg, shape, py, px):
    # tiles whose planes are all halo, single rows and columns, fields smaller than a relaxation tile: segmenting and merging
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.integers(0, 200, shape, dtype=np.uint8)
    n = max(shape[0] * shape[1] // 5, 1)
    seeds = np.stack([rng.integers(0, shape[0], n), rng.integers(0, shape[1], n)], axis=1).astype(np.uint64)
    seeds = np.unique(seeds, axis=0)
    got, _, halo_ok = _tiled2d(pkg, img, seeds, py, px)
    assert halo_ok and (got == ol.segment_arrival(img, seeds)).all(), (shape, py, px)
    got, _, _ = _tiled2d(pkg, img, seeds, py, px, max_level=120, merging=True)
    assert (got == ol.merge_arrival(img, seeds, max_level=120)).all(), (shape, py, px, "merging")
