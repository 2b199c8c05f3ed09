"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on GPUs; "gloo" for CPU rehearsal).

* Independent slices (BASELINE config C4) shard with NO collective on the data path:
  `shard_slices` assigns slice i to rank i % world, each rank runs its own transforms.
* One field tiled over ranks (config C5): row blocks with a 1-row halo.  The flood is the unique
  fixpoint of the arrival-stamp recurrence, so the tiled run is bit-exact with
  the single-domain run: every rank relaxes its block to local convergence, neighbours swap halo
  rows, a 1-word all-reduce says whether anyone changed; repeat until quiet, then the same loop for
  the labels.  Messages are one image row of u32 (128 KiB at 32768 columns): latency-bound, so the
  exchange count (a handful, not ~2000 as in a per-ring sweep) is what matters on xGMI.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _ffi


def shard_slices(n_slices, rank, world):
    """Indices of the slices rank `rank` processes (round robin, SURVEY 8e: slice i -> GPU i mod world)."""
    return list(range(rank, n_slices, world))


def row_block(height, rank, world):
    """(r0, r1, lo, hi): owned global rows [r0, r1) and local-plane rows [lo, hi) including halos.
    Every rank must own at least one row: a rank without rows would hand a halo row on as if it were its own."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of {world}")
    if height < world:
        raise ValueError(f"a field of {height} rows cannot be split into {world} row blocks: use at most {height} ranks")
    base, extra = divmod(height, world)
    r0 = rank * base + min(rank, extra)
    r1 = r0 + base + (1 if rank < extra else 0)
    lo = r0 - 1 if rank > 0 else r0
    hi = r1 + 1 if rank < world - 1 else r1
    return r0, r1, lo, hi


def local_seeds(seeds, lo, hi):
    """Seeds (n, 2) in global coordinates -> (local (m, 2) int32, colours (m,) int32) for the rows
    [lo, hi) of one rank; colour = index in the caller's slice + 1 (lib.rs:1670-1672)."""
    seeds = torch.as_tensor(seeds).to(torch.int64).reshape(-1, 2)
    idx = torch.nonzero((seeds[:, 0] >= lo) & (seeds[:, 0] < hi)).flatten()
    loc = seeds[idx].clone()
    loc[:, 0] -= lo
    return loc.to(torch.int32).contiguous(), (idx + 1).to(torch.int32).contiguous()


class HipBlockEngine:
    """A rank's row block on its GPU: ws_block_* of include/ws_hip.h on torch tensors."""

    def __init__(self, device_engine, img_block, seeds_local, colours, max_level=254):
        self.eng = device_engine
        self.img = img_block.contiguous()
        assert self.img.is_cuda and self.img.dtype == torch.uint8
        self.h, self.w = self.img.shape
        dev = self.img.device
        self.seeds = seeds_local.to(dev)
        self.colours = colours.to(dev)
        self.max_level = max_level
        self.keys = torch.empty((self.h, self.w), dtype=torch.int32, device=dev)
        self.labels = torch.empty((self.h, self.w), dtype=torch.int32, device=dev)

    def init(self):
        n = int(self.seeds.shape[0])
        self.eng.ctx.check(_ffi.lib().ws_block_init(self.eng.ctx.handle, self.h, self.w,
                                                    self.seeds.data_ptr() if n else None,
                                                    self.colours.data_ptr() if n else None, n,
                                                    self.keys.data_ptr(), self.labels.data_ptr()))

    def relax(self):
        ch = ctypes.c_int(0)
        self.eng.ctx.check(_ffi.lib().ws_block_relax(self.eng.ctx.handle, self.img.data_ptr(), self.h, self.w, self.w,
                                                     self.max_level, self.keys.data_ptr(), ctypes.byref(ch)))
        return bool(ch.value)

    def resolve(self):
        ch = ctypes.c_int(0)
        self.eng.ctx.check(_ffi.lib().ws_block_resolve(self.eng.ctx.handle, self.keys.data_ptr(), self.labels.data_ptr(),
                                                       self.h, self.w, ctypes.byref(ch)))
        return bool(ch.value)


def _comm_device(plane):
    """Tensors handed to the backend: RCCL moves device memory, gloo needs host memory."""
    return plane.device if dist.get_backend() == "nccl" else torch.device("cpu")


def exchange_halos(plane, rank, world, group=None):
    """Swap halo rows with the neighbour ranks.  `plane` is the local (h, w) tensor whose first row
    is a halo iff rank > 0 and whose last row is a halo iff rank < world - 1."""
    if world == 1:
        return
    dev = _comm_device(plane)
    ops, recvs = [], []
    if rank > 0:                      # upper neighbour: send my first owned row, receive my top halo
        send = plane[1].to(dev).contiguous()
        recv = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, rank - 1, group), dist.P2POp(dist.irecv, recv, rank - 1, group)]
        recvs.append((0, recv))
    if rank < world - 1:              # lower neighbour: send my last owned row, receive my bottom halo
        send = plane[-2].to(dev).contiguous()
        recv = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, rank + 1, group), dist.P2POp(dist.irecv, recv, rank + 1, group)]
        recvs.append((plane.shape[0] - 1, recv))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for row, recv in recvs:
        plane[row].copy_(recv)


def _any_rank(flag, like, group=None):
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_comm_device(like))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())


def segment_tiled(block, rank, world, group=None, max_rounds=1 << 20):
    """Segmenting transform of one field tiled over `world` ranks.  `block` is this rank's engine
    (HipBlockEngine, or any object with init/relax/resolve and .keys/.labels planes).
    Returns (owned label rows, exchange rounds)."""
    block.init()
    rounds = 0
    for plane_name, step in (("keys", block.relax), ("labels", block.resolve)):
        for _ in range(max_rounds):
            changed = step()
            exchange_halos(getattr(block, plane_name), rank, world, group)
            rounds += 1
            if world == 1 or not _any_rank(changed, getattr(block, plane_name), group):
                break
    top = 1 if rank > 0 else 0
    bot = block.labels.shape[0] - (1 if rank < world - 1 else 0)
    return block.labels[top:bot], rounds
