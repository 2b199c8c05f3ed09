"""World-size-2 (and 3) rehearsal of the multi-GPU drivers on CPU with the gloo backend:
the tiled single-field path (row blocks + halo exchange + 1-word all-reduce) must be bit-exact
with the single-domain oracle, and slice sharding must cover a batch exactly once."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge
import cases
import oracle_lib as ol


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiled_worker(rank, world, port, img, seeds, max_level, outdir, force_general=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        ge.load_package()
        wd = importlib.import_module("rustronomy_watershed_amd.distributed")
        from numpy_engine import NumpyBlockEngine
        r0, r1, lo, hi = wd.row_block(img.shape[0], rank, world)
        loc, col = wd.local_seeds(seeds.astype(np.int64), lo, hi)
        block = NumpyBlockEngine(img[lo:hi], loc.numpy(), col.numpy(), max_level)
        block.force_general = force_general and rank == world - 1      # ONE rank that cannot: every rank must follow
        owned, rounds = wd.segment_tiled(block, rank, world)
        np.save(os.path.join(outdir, f"fast{rank}.npy"), np.array([block.fast and not block.force_general]))
        assert owned.shape[0] == r1 - r0
        np.save(os.path.join(outdir, f"part{rank}.npy"), owned.numpy().view(np.uint32))
        np.save(os.path.join(outdir, f"rounds{rank}.npy"), np.array([rounds]))
    finally:
        dist.destroy_process_group()


def _arrival_worker(rank, world, port, img, seeds, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        ge.load_package()
        wd = importlib.import_module("rustronomy_watershed_amd.distributed")
        from numpy_engine import NumpyBlockEngine
        r0, r1, lo, hi = wd.row_block(img.shape[0], rank, world)
        loc, col = wd.local_seeds(seeds.astype(np.int64), lo, hi)
        block = NumpyBlockEngine(img[lo:hi], loc.numpy(), col.numpy(), 254)
        wd.segment_tiled(block, rank, world)
        planes = wd.gather_arrival_planes(block, rank, world, img.shape[0])
        assert (planes is not None) == (rank == 0)
        if rank == 0:
            np.save(os.path.join(outdir, "keys.npy"), planes[0].numpy().view(np.uint32))
            np.save(os.path.join(outdir, "labels.npy"), planes[1].numpy().view(np.uint32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_arrival_planes_gathered_on_rank_0_are_the_single_domain_planes(world):
    # transform_to_list of a tiled field (ws_transform_to_list_tiled_device) makes its lake records on rank 0 from the gathered
    # arrival stamps and labels of every rank's owned rows: here the gather over gloo, against the oracle's planes of the whole field
    img = cases.smooth_field(57, 44, 6)
    seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_arrival_worker, args=(world, _free_port(), img, seeds, d), nprocs=world, join=True)
        keys, labels = np.load(os.path.join(d, "keys.npy")), np.load(os.path.join(d, "labels.npy"))
    want_labels, want_keys = ol.segment_arrival(img, seeds, want_keys=True)
    assert (labels == want_labels).all()
    # the oracle's stamps are (level << 32 | ring) in 64 bits, ~0 where no flood arrives; the engine's (level << 24 | ring) in 32
    # bits, level 255 and up for "never" (csrc/ws_common.hpp: KEY_INF)
    never = want_keys == np.uint64(0xFFFFFFFFFFFFFFFF)
    packed = ((want_keys >> np.uint64(32)) << np.uint64(24)) | (want_keys & np.uint64(0xFFFFFF))
    assert (keys[~never] == packed[~never]).all() and (keys[never] >= 0xFF000000).all()


def _run_tiled(img, seeds, world, max_level=254, force_general=False):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_tiled_worker, args=(world, _free_port(), img, np.asarray(seeds, dtype=np.uint64).reshape(-1, 2), max_level, d, force_general),
                 nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"part{r}.npy")) for r in range(world)]
        rounds = [int(np.load(os.path.join(d, f"rounds{r}.npy"))[0]) for r in range(world)]
    assert len(set(rounds)) == 1                     # every rank takes part in every exchange
    return np.concatenate(parts, axis=0), rounds[0]


def _tiled2d_worker(rank, py, px, port, img, seeds, max_level, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=py * px)
    try:
        import importlib
        ge.load_package()
        wd = importlib.import_module("rustronomy_watershed_amd.distributed")
        from numpy_engine import NumpyBlockEngine
        (r0, r1, lo, hi), (c0, c1, clo, chi) = wd.tile_grid(img.shape[0], img.shape[1], rank, py, px)
        loc, col = wd.local_seeds2d(seeds.astype(np.int64), lo, hi, clo, chi)
        block = NumpyBlockEngine(img[lo:hi, clo:chi], loc.numpy(), col.numpy(), max_level)
        labels, rounds = wd.segment_tiled2d(block, rank, py, px)
        np.save(os.path.join(outdir, f"part{rank}.npy"), labels.numpy().view(np.uint32)[r0 - lo:r1 - lo, c0 - clo:c1 - clo])
        np.save(os.path.join(outdir, f"rounds{rank}.npy"), np.array([rounds]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("py,px", [(2, 2), (1, 3)])
def test_field_in_2d_tiles_over_gloo_is_bit_exact_with_single_domain(py, px):
    # the field cut in BOTH directions (csrc/ws_tiled.hip: tiled2d_rank; here distributed.segment_tiled2d over gloo with the
    # numpy stand-in for the block steps): halo rows and columns, any seed list, every rank in every exchange
    wd_rows = lambda n, k, parts: (n // parts * k + min(k, n % parts), n // parts * (k + 1) + min(k + 1, n % parts))
    for img, few in ((cases.field(61, 90, 3), False), (cases.smooth_field(70, 83, 5), True)):
        seeds = np.asarray(ol.find_local_minima(img), dtype=np.uint64).reshape(-1, 2)
        if few:
            seeds = seeds[:: max(len(seeds) // 3, 1)][:3]
        else:
            seeds = seeds[np.random.default_rng(1).permutation(len(seeds))]
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_tiled2d_worker, args=(py, px, _free_port(), img, seeds, 254, d), nprocs=py * px, join=True)
            got = np.zeros(img.shape, dtype=np.uint32)
            rounds = set()
            for rank in range(py * px):
                (r0, r1), (c0, c1) = wd_rows(img.shape[0], rank // px, py), wd_rows(img.shape[1], rank % px, px)
                got[r0:r1, c0:c1] = np.load(os.path.join(d, f"part{rank}.npy"))
                rounds.add(int(np.load(os.path.join(d, f"rounds{rank}.npy"))[0]))
        assert len(rounds) == 1
        assert (got == ol.segment_arrival(img, seeds)).all(), (py, px, few)


@pytest.mark.parametrize("world", [2, 3])
def test_tiled_field_is_bit_exact_with_single_domain(world):
    img = cases.field(61, 48, 3)
    seeds = ol.find_local_minima(img)
    got, rounds = _run_tiled(img, seeds, world)
    assert (got == ol.segment(img, seeds)).all()
    assert 3 <= rounds <= 8                          # form vote + stamp exchanges (a handful) + ONE table exchange for the labels
    # the general form (one rank cannot use the fast one, so all ranks take it): stamp rounds AND label rounds
    got, rounds_general = _run_tiled(img, seeds, world, force_general=True)
    assert (got == ol.segment(img, seeds)).all()
    assert rounds_general >= 5
    # a seed list that is not sorted: no rank can use the fast form
    rng = np.random.default_rng(world)
    shuffled = seeds[rng.permutation(len(seeds))]
    got, _ = _run_tiled(img, shuffled, world)
    assert (got == ol.segment(img, shuffled)).all()


def test_tiled_field_long_paths_cross_the_seam_many_times():
    # a serpentine corridor that crosses the block boundary on every lap: many exchange rounds
    img = np.full((40, 30), 255, np.uint8)
    for c in range(1, 29, 2):
        img[1:39, c] = 4
        img[38 if (c // 2) % 2 == 0 else 1, c + 1] = 4
    seeds = [(1, 1)]
    want = ol.segment(img, seeds)
    got, rounds = _run_tiled(img, seeds, 2)
    assert (got == want).all()
    assert rounds > 10                               # the stamps need a round per crossing ...
    got, rounds_general = _run_tiled(img, seeds, 2, force_general=True)
    assert (got == want).all()
    assert rounds_general > rounds + 5               # ... the labels only in the general form: the table exchange is one


def test_tiled_field_seeds_on_halo_rows_and_low_max_level():
    img = cases.smooth_field(50, 40, 5)
    r0, r1, lo, hi = 25, 50, 24, 50                   # world 2: rank 1 owns rows 25.., halo row 24
    seeds = [(24, 7), (25, 20), (23, 30), (26, 31), (10, 10), (40, 5)]
    for maxlvl in (60, 254):
        got, _ = _run_tiled(img, seeds, 2, max_level=maxlvl)
        assert (got == ol.segment(img, seeds, max_level=maxlvl)).all()


def test_tiled_field_more_ranks_than_a_chain_is_long():
    # world 4 on a smooth field: label chains run through several blocks (a boundary-table entry that refers to an
    # entry of the next rank, which refers on), blocks of 8-9 rows
    img = cases.smooth_field(34, 44, 9)
    seeds = ol.find_local_minima(img)[::3]
    got, _ = _run_tiled(img, seeds, 4)
    assert (got == ol.segment(img, seeds)).all()
    with pytest.raises(Exception):
        ge.load_package()
        import importlib
        importlib.import_module("rustronomy_watershed_amd.distributed").row_block(3, 0, 4)      # fewer rows than ranks


def _merge_worker(rank, world, port, img, seeds, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        ge.load_package()
        wd = importlib.import_module("rustronomy_watershed_amd.distributed")
        from numpy_engine import NumpyBlockEngine
        r0, r1, lo, hi = wd.row_block(img.shape[0], rank, world)
        loc, col = wd.local_seeds(seeds.astype(np.int64), lo, hi)
        block = NumpyBlockEngine(img[lo:hi], loc.numpy(), col.numpy())
        owned, _ = wd.merge_tiled(block, rank, world, lo, img.shape[0], len(seeds))
        np.save(os.path.join(outdir, f"merged{rank}.npy"), owned.numpy().view(np.uint32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,kind", [(2, (40, 36), "noise"), (3, (45, 40), "smooth"), (4, (34, 44), "smooth")])
def test_merging_final_labels_across_row_blocks(world, shape, kind):
    # the merging transform's final canonical labels (smallest seed colour of every lake) of a tiled field: lakes that
    # span several blocks, some of them in locally disconnected pieces, against the single-domain oracle
    img = cases.field(*shape, 11) if kind == "noise" else cases.smooth_field(*shape, 4)
    seeds = ol.find_local_minima(img)
    if kind == "smooth":
        seeds = seeds[::2]
    # walls (NEVER_FILL columns with gaps) so that not everything ends up in one lake
    img = img.copy()
    img[:, shape[1] // 2] = 255
    img[shape[0] // 3, shape[1] // 2] = 3
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_merge_worker, args=(world, _free_port(), img, np.asarray(seeds, dtype=np.uint64).reshape(-1, 2), d), nprocs=world, join=True)
        got = np.concatenate([np.load(os.path.join(d, f"merged{r}.npy")) for r in range(world)], axis=0)
    want = ol.merge_arrival(img, seeds)
    assert (got == want).all()
    assert len(np.unique(want)) >= 2


def test_row_blocks_and_slice_sharding_partition_exactly():
    pkg = ge.load_package()
    import importlib
    wd = importlib.import_module("rustronomy_watershed_amd.distributed")
    for h, world in ((61, 2), (64, 8), (7, 3), (32768, 8)):
        rows = []
        for r in range(world):
            r0, r1, lo, hi = wd.row_block(h, r, world)
            assert lo == (r0 - 1 if r > 0 else r0) and hi == (r1 + 1 if r < world - 1 else r1)
            rows += list(range(r0, r1))
        assert rows == list(range(h))
    for n, world in ((64, 8), (5, 2), (3, 4)):
        got = sorted(i for r in range(world) for i in wd.shard_slices(n, r, world))
        assert got == list(range(n))
    assert pkg is not None


def _batch_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        ge.load_package()
        wd = importlib.import_module("rustronomy_watershed_amd.distributed")
        mine = wd.shard_slices(5, rank, world)
        # independent slices: no collective on the data path; only the timing reduction of bench.py
        px = 0
        for i in mine:
            img = cases.field(24, 24, 100 + i)
            out = ol.segment(img, ol.find_local_minima(img))
            np.save(os.path.join(outdir, f"slice{i}.npy"), out)
            px += img.size
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # bench.py takes the max time over ranks
        tot = torch.tensor([px], dtype=torch.int64)
        dist.all_reduce(tot)
        if rank == 0:
            np.save(os.path.join(outdir, "summary.npy"), np.array([t.item(), tot.item()]))
    finally:
        dist.destroy_process_group()


def test_independent_slices_shard_without_data_collective():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_batch_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        tmax, total = np.load(os.path.join(d, "summary.npy"))
        assert tmax == 2.0 and total == 5 * 24 * 24
        for i in range(5):
            img = cases.field(24, 24, 100 + i)
            assert (np.load(os.path.join(d, f"slice{i}.npy")) == ol.segment(img, ol.find_local_minima(img))).all()
