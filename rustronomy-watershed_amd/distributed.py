"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on GPUs; "gloo" for CPU rehearsal).

* Independent slices (BASELINE config C4) shard with NO collective on the data path:
  `shard_slices` assigns slice i to rank i % world, each rank runs its own (stacked) transforms.
* One field tiled over ranks (config C5): row blocks with a 1-row halo.  The flood is the unique
  fixpoint of the arrival-stamp recurrence, so the tiled run is bit-exact with
  the single-domain run.
    - Stamps: every rank relaxes its block to local convergence with the single-GPU kernels (seed side
      tables, fused passes), neighbours swap halo rows, and a 1-word all-reduce says whether any rank
      RECEIVED a row that differs from the one it held; if so the ranks re-relax from the tile rows that
      hold the halo rows (`ws_block_relax_halo`) and swap again.  A flood crosses a seam a handful of
      times at most, so this is 2-4 exchanges, not one per flood ring.  One host read per exchange.
    - Labels: no rounds at all.  After the local two-launch resolve every label is a colour or a
      reference to a neighbour's boundary-row pixel; the boundary rows of all ranks form a closed table
      (2 x width words per rank), all-gathered once, resolved by every rank for itself
      (`ws_block_export_boundary` / `ws_block_import_boundary`, csrc/ws_block.hip).
  Seed lists that are not strictly increasing, or widths with w % 4 != 0, take the general form
  (`ws_block_init` / `_relax` / `_resolve`: painted seeds, iterative label rounds), chosen by all ranks
  together.  Messages are image rows of u32 (128 KiB at 32768 columns): latency-bound on xGMI.
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
    seeds = torch.as_tensor(seeds).reshape(-1, 2)
    rows = seeds[:, 0].contiguous()
    if rows.numel() > 1 and bool((rows[1:] >= rows[:-1]).all()):
        # rows in non-decreasing order (every row-major sorted list, find_local_minima's in particular): the rank's seeds
        # are ONE contiguous range of the list, found by two binary searches -- no 100-M-element masks or index lists
        bounds = torch.searchsorted(rows, torch.tensor([lo, hi], dtype=rows.dtype, device=rows.device))
        i0, i1 = int(bounds[0]), int(bounds[1])
        loc = seeds[i0:i1].to(torch.int32).clone()
        loc[:, 0] -= lo
        colours = torch.arange(i0 + 1, i1 + 1, dtype=torch.int32, device=seeds.device)
        return loc.contiguous(), colours
    seeds = seeds.to(torch.int64)
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
        self.seeds = seeds_local.to(dev).contiguous()
        self.colours = colours.to(dev).contiguous()
        self.max_level = max_level
        self.keys = torch.empty((self.h, self.w), dtype=torch.int32, device=dev)
        self.labels = torch.empty((self.h, self.w), dtype=torch.int32, device=dev)
        self.halo_top = self.halo_bot = False
        n = int(self.seeds.shape[0])
        # the fast form wants the rank's seeds to be one contiguous, strictly increasing range of the caller's list (the
        # order is checked by ws_block_begin on the device; that the colours count up by one, here)
        consecutive = n < 2 or bool((self.colours[1:] - self.colours[:-1] == 1).all().item())
        self.first_colour = int(self.colours[0].item()) if n else 1
        self.fast = consecutive and self.w % 4 == 0 and self.h >= 2 and self.h * self.w < 2 ** 31

    def set_halos(self, top, bottom):
        self.halo_top, self.halo_bot = bool(top), bool(bottom)

    def _check(self, rc):
        self.eng.ctx.check(rc)

    # ---- fast form -------------------------------------------------------------------------------------------------
    def try_begin(self):
        """Seed tables + relaxation to local convergence.  False: this block needs the general form."""
        if not self.fast:
            return False
        n = int(self.seeds.shape[0])
        rc = _ffi.lib().ws_block_begin(self.eng.ctx.handle, self.img.data_ptr(), self.h, self.w, self.w, self.max_level,
                                       self.seeds.data_ptr() if n else None, n, self.first_colour, self.keys.data_ptr())
        if rc == _ffi.WS_ERR_UNSUPPORTED:
            self.fast = False
            self.why_not_fast = _ffi.lib().ws_last_error(self.eng.ctx.handle).decode()
            return False
        self._check(rc)
        return True

    def relax_halo(self):
        self._check(_ffi.lib().ws_block_relax_halo(self.eng.ctx.handle, self.img.data_ptr(), self.h, self.w, self.w, self.max_level,
                                                   int(self.halo_top), int(self.halo_bot), self.keys.data_ptr()))

    def resolve_local(self):
        self._check(_ffi.lib().ws_block_resolve_local(self.eng.ctx.handle, self.keys.data_ptr(), self.labels.data_ptr(), self.h, self.w,
                                                      int(self.halo_top), int(self.halo_bot)))

    def export_boundary(self, rank):
        rows = torch.empty((2, self.w), dtype=torch.int32, device=self.img.device)
        self._check(_ffi.lib().ws_block_export_boundary(self.eng.ctx.handle, self.labels.data_ptr(), self.h, self.w,
                                                        int(self.halo_top), int(self.halo_bot), rank, rows.data_ptr()))
        return rows

    def import_boundary(self, table, rank, world):
        table = table.to(self.img.device).contiguous()
        assert table.numel() == world * 2 * self.w and table.dtype == torch.int32
        self._check(_ffi.lib().ws_block_import_boundary(self.eng.ctx.handle, table.data_ptr(), world, rank, self.labels.data_ptr(),
                                                        self.h, self.w, int(self.halo_top), int(self.halo_bot)))

    def single(self):
        """world == 1: the block is the whole field -- the ordinary single-device transform."""
        return self.eng.segment(self.img, self.seeds, max_level=self.max_level, out=self.labels)

    def single_merge(self):
        return self.eng.merge(self.img, self.seeds, max_level=self.max_level)

    # ---- merging across blocks: final canonical labels -------------------------------------------------------------
    def merge_local(self, row0, field_rows, n_colours_total):
        """A union-find over all seed colours of the field, this block's touching colours joined."""
        self.n_colours_total = int(n_colours_total)
        self.parent = torch.empty(self.n_colours_total + 1, dtype=torch.int32, device=self.img.device)
        self._check(_ffi.lib().ws_block_merge_local(self.eng.ctx.handle, self.labels.data_ptr(), self.h, self.w, row0, field_rows,
                                                    self.n_colours_total, self.parent.data_ptr()))

    def merge_export(self):
        pairs = torch.empty((4 * self.w, 2), dtype=torch.int32, device=self.img.device)
        self._check(_ffi.lib().ws_block_merge_export(self.eng.ctx.handle, self.labels.data_ptr(), self.h, self.w,
                                                     self.parent.data_ptr(), pairs.data_ptr()))
        return pairs

    def merge_import(self, pairs):
        pairs = pairs.to(self.img.device).contiguous()
        self._check(_ffi.lib().ws_block_merge_import(self.eng.ctx.handle, pairs.data_ptr(), pairs.numel() // 2, self.parent.data_ptr()))

    def merge_relabel(self):
        out = torch.empty_like(self.labels)
        self._check(_ffi.lib().ws_block_merge_relabel(self.eng.ctx.handle, self.labels.data_ptr(), self.labels.numel(),
                                                      self.parent.data_ptr(), self.n_colours_total, out.data_ptr()))
        return out

    # ---- general form (any seed list, any width) -------------------------------------------------------------------
    def init(self):
        n = int(self.seeds.shape[0])
        self._check(_ffi.lib().ws_block_init(self.eng.ctx.handle, self.h, self.w,
                                             self.seeds.data_ptr() if n else None,
                                             self.colours.data_ptr() if n else None, n,
                                             self.keys.data_ptr(), self.labels.data_ptr()))

    def relax(self):
        ch = ctypes.c_int(0)
        self._check(_ffi.lib().ws_block_relax(self.eng.ctx.handle, self.img.data_ptr(), self.h, self.w, self.w,
                                              self.max_level, self.keys.data_ptr(), ctypes.byref(ch)))
        return bool(ch.value)

    def resolve(self):
        ch = ctypes.c_int(0)
        self._check(_ffi.lib().ws_block_resolve(self.eng.ctx.handle, self.keys.data_ptr(), self.labels.data_ptr(),
                                                self.h, self.w, ctypes.byref(ch)))
        return bool(ch.value)


def _comm_device(plane):
    """Tensors handed to the backend: RCCL moves device memory, gloo needs host memory."""
    return plane.device if dist.get_backend() == "nccl" else torch.device("cpu")


def swap_halo_rows(plane, rank, world, group=None):
    """Sends the first / last OWNED row to the upper / lower neighbour and receives their boundary rows.
    Returns [(halo row index, received row on plane.device)]; the plane is not written."""
    if world == 1:
        return []
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
    return [(row, recv.to(plane.device)) for row, recv in recvs]


def exchange_halos(plane, rank, world, group=None):
    """Swap halo rows with the neighbour ranks, in place.  `plane` is the local (h, w) tensor whose first row
    is a halo iff rank > 0 and whose last row is a halo iff rank < world - 1."""
    for row, recv in swap_halo_rows(plane, rank, world, group):
        plane[row].copy_(recv)


def _reduce_flag(flag, like, op, group=None):
    """One-word all-reduce of a Python bool or a 0-d / 1-element tensor; ONE host read."""
    if isinstance(flag, torch.Tensor):
        t = flag.to(torch.int32).reshape(1).to(_comm_device(like))
    else:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_comm_device(like))
    dist.all_reduce(t, op=op, group=group)
    return bool(t.item())


def _any_rank(flag, like, group=None):
    return _reduce_flag(flag, like, dist.ReduceOp.MAX, group)


def _all_ranks(flag, like, group=None):
    return _reduce_flag(flag, like, dist.ReduceOp.MIN, group)


def _gather_rows(rows, world, group=None):
    """(2, w) per rank -> (world, 2, w) on every rank."""
    dev = _comm_device(rows)
    mine = rows.to(dev).contiguous()
    if dist.get_backend() == "nccl":
        out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=dev)
        dist.all_gather_into_tensor(out, mine, group=group)
        return out
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.stack(parts)


def _segment_tiled_general(block, rank, world, group, max_rounds):
    """Any seed list: painted seeds, iterative relaxation AND iterative label rounds (one halo swap + one flag each)."""
    block.init()
    rounds = 0
    for plane_name, step in (("keys", block.relax), ("labels", block.resolve)):
        for _ in range(max_rounds):
            changed = step()
            exchange_halos(getattr(block, plane_name), rank, world, group)
            rounds += 1
            if not _any_rank(changed, getattr(block, plane_name), group):
                break
    return rounds


def segment_tiled(block, rank, world, group=None, max_rounds=1 << 20):
    """Segmenting transform of one field tiled over `world` ranks.  `block` is this rank's engine
    (HipBlockEngine, or any object with the same steps and .keys / .labels planes).
    Returns (owned label rows, number of collective exchanges)."""
    if world == 1:
        return block.single(), 0
    block.set_halos(rank > 0, rank < world - 1)
    fast = block.try_begin()
    if not _all_ranks(fast, block.keys, group):          # decided together: the two forms exchange different things
        rounds = 1 + _segment_tiled_general(block, rank, world, group, max_rounds)
    else:
        rounds = 1
        for _ in range(max_rounds):
            got = swap_halo_rows(block.keys, rank, world, group)
            rounds += 1
            new = None
            for row, recv in got:
                d = (recv != block.keys[row]).any()
                new = d if new is None else (new | d)
            if not _any_rank(new if new is not None else False, block.keys, group):
                break                                      # every halo row already equals its neighbour's boundary row
            for row, recv in got:
                block.keys[row].copy_(recv)
            block.relax_halo()
        block.resolve_local()
        table = _gather_rows(block.export_boundary(rank), world, group)
        block.import_boundary(table.reshape(-1), rank, world)
        rounds += 1
    top = 1 if rank > 0 else 0
    bot = block.labels.shape[0] - (1 if rank < world - 1 else 0)
    return block.labels[top:bot], rounds


def gather_arrival_planes(block, rank, world, field_rows, group=None):
    """After segment_tiled: every rank's OWNED rows of the arrival stamps and of the labels -> rank 0's whole planes (one message
    per rank and plane; csrc/ws_tiled.hip: Exchange::gather_rows).  Returns (keys, labels) of field_rows x w on rank 0, None
    elsewhere.  What transform_to_list of a tiled field reads (ws_transform_to_list_tiled_device: rank 0 then makes the lake
    records of all levels from these two planes, ws_lists_from_arrival_device)."""
    top = 1 if rank > 0 else 0
    out = []
    for plane in (block.keys, block.labels):
        bot = plane.shape[0] - (1 if rank < world - 1 else 0)
        mine = plane[top:bot].to(_comm_device(plane)).contiguous()
        if world == 1:
            out.append(mine)
            continue
        if rank == 0:
            full = torch.empty((field_rows, plane.shape[1]), dtype=plane.dtype, device=mine.device)
            full[: mine.shape[0]] = mine
            for r in range(1, world):
                r0, r1, _, _ = row_block(field_rows, r, world)
                if r1 > r0:
                    part = torch.empty((r1 - r0, plane.shape[1]), dtype=plane.dtype, device=mine.device)
                    dist.recv(part, src=r, group=group)
                    full[r0:r1] = part
            out.append(full)
        else:
            if mine.shape[0]:
                dist.send(mine, dst=0, group=group)
            out.append(None)
    return (out[0], out[1]) if rank == 0 else None


# ---- the field cut in both directions (csrc/ws_tiled.hip: tiled2d_rank) ---------------------------------------------------------

def tile_grid(height, width, rank, py, px):
    """Rows and columns of rank = ty * px + tx: ((r0, r1, lo, hi), (c0, c1, clo, chi)) -- row_block in each direction."""
    return row_block(height, rank // px, py), row_block(width, rank % px, px)


def local_seeds2d(seeds, lo, hi, clo, chi):
    """The seeds on a tile's plane (halo ring included), local coordinates, with their colours (index + 1, lib.rs:1670-1672)."""
    seeds = torch.as_tensor(seeds).reshape(-1, 2)
    inside = (seeds[:, 0] >= lo) & (seeds[:, 0] < hi) & (seeds[:, 1] >= clo) & (seeds[:, 1] < chi)
    loc = seeds[inside].clone()
    loc[:, 0] -= lo
    loc[:, 1] -= clo
    return loc, torch.arange(1, seeds.shape[0] + 1, dtype=torch.int64)[inside]


def exchange_halos2d(plane, rank, py, px, group=None):
    """Halo rows AND columns of a tile's plane with its four neighbours, in place (the ring's corner cells are nobody's
    neighbours in a 4-connected stencil)."""
    ty, tx = divmod(rank, px)
    up, down, left, right = ty > 0, ty < py - 1, tx > 0, tx < px - 1
    h, w = plane.shape
    dev = _comm_device(plane)
    ops, recvs = [], []

    def pair(send_view, peer, put):
        send = send_view.to(dev).contiguous()
        recv = torch.empty_like(send)
        ops.extend([dist.P2POp(dist.isend, send, peer, group), dist.P2POp(dist.irecv, recv, peer, group)])
        recvs.append((put, recv))
    if up:
        pair(plane[1], rank - px, lambda r: plane[0].copy_(r))
    if down:
        pair(plane[h - 2], rank + px, lambda r: plane[h - 1].copy_(r))
    if left:
        pair(plane[:, 1], rank - 1, lambda r: plane[:, 0].copy_(r))
    if right:
        pair(plane[:, w - 2], rank + 1, lambda r: plane[:, w - 1].copy_(r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for put, recv in recvs:          # rows first, then columns (the order pair() was called in)
        put(recv.to(plane.device))


def segment_tiled2d(block, rank, py, px, group=None, max_rounds=1 << 20):
    """Segmenting transform of one field in py x px tiles: the general form's steps on every tile (painted seeds with their
    global colours, relaxation rounds, label rounds), halo rows and columns swapped after each.  `block`: this rank's engine
    over its tile plane.  Returns (the labels of the tile's plane, number of collective exchanges)."""
    block.init()
    rounds = 0
    for plane_name, step in (("keys", block.relax), ("labels", block.resolve)):
        for _ in range(max_rounds):
            changed = step()
            exchange_halos2d(getattr(block, plane_name), rank, py, px, group)
            rounds += 1
            if not _any_rank(changed, getattr(block, plane_name), group):
                break
    return block.labels, rounds


def merge_tiled(block, rank, world, row0, field_rows, n_colours_total, group=None):
    """Final canonical labels of the MERGING transform of one field tiled over `world` ranks (lib.rs:1328-1522 after the
    last level: every lake carries the smallest seed colour in it).  `row0`: field row of the block's first LOCAL row
    (its halo row, if it has one); `n_colours_total`: seeds of the whole field.  Runs the tiled segmenting transform,
    joins the touching colours of the block, and exchanges ONE table: the (colour, local root) pairs of every rank's
    boundary and halo rows (4 x width pairs per rank).  Returns (owned label rows, number of collective exchanges)."""
    if world == 1:
        return block.single_merge(), 0
    _, rounds = segment_tiled(block, rank, world, group)
    block.merge_local(row0, field_rows, n_colours_total)
    pairs = _gather_rows(block.merge_export(), world, group)          # (world, 4 w, 2)
    block.merge_import(pairs.reshape(-1, 2))
    out = block.merge_relabel()
    top = 1 if rank > 0 else 0
    bot = out.shape[0] - (1 if rank < world - 1 else 0)
    return out[top:bot], rounds + 1
