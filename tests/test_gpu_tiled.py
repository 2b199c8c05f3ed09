"""One field tiled over 2 ranks, HIP kernels on every rank (both ranks share the single GPU of the
test box; halo rows travel over gloo here -- on a real node the same code runs over RCCL)."""
import os
import tempfile

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge
import cases
import oracle_lib as ol
from test_distributed_cpu import _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, img, seeds, max_level, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        import torch
        ge.load_package()
        wd = importlib.import_module("rustronomy_watershed_amd.distributed")
        dev = importlib.import_module("rustronomy_watershed_amd.device")
        eng = dev.DeviceEngine(0)
        r0, r1, lo, hi = wd.row_block(img.shape[0], rank, world)
        loc, col = wd.local_seeds(seeds.astype(np.int64), lo, hi)
        block = wd.HipBlockEngine(eng, torch.from_numpy(img[lo:hi].copy()).cuda(), loc, col, max_level)
        owned, rounds = wd.segment_tiled(block, rank, world)
        torch.cuda.synchronize()
        np.save(os.path.join(outdir, f"part{rank}.npy"), owned.cpu().numpy().view(np.uint32))
        np.save(os.path.join(outdir, f"rounds{rank}.npy"), np.array([rounds]))
    finally:
        dist.destroy_process_group()


def _run(img, seeds, world, max_level=254):
    ge.build_hip()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), img, np.asarray(seeds, dtype=np.uint64).reshape(-1, 2), max_level, d),
                 nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"part{r}.npy")) for r in range(world)]
        rounds = int(np.load(os.path.join(d, "rounds0.npy"))[0])
    return np.concatenate(parts, axis=0), rounds


def test_tiled_hip_two_ranks_bit_exact_random_field():
    img = cases.field(700, 520, 3)
    seeds = ol.find_local_minima(img)
    got, rounds = _run(img, seeds, 2)
    assert (got == ol.segment_arrival(img, seeds)).all()
    assert rounds < 20            # a handful of exchanges, not one per flood ring


def test_tiled_hip_three_ranks_smooth_field_and_corridor():
    img = cases.smooth_field(300, 260, 2)
    seeds = ol.find_local_minima(img)
    got, _ = _run(img, seeds, 3, max_level=200)
    assert (got == ol.segment(img, seeds, max_level=200)).all()
    cor = np.full((130, 90), 255, np.uint8)
    for c in range(1, 89, 2):
        cor[1:129, c] = 4
        cor[128 if (c // 2) % 2 == 0 else 1, c + 1] = 4
    got, rounds = _run(cor, [(1, 1)], 2)
    assert (got == ol.segment(cor, [(1, 1)])).all()
    assert rounds > 20
