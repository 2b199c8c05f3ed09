"""ws_group_* of include/ws_hip.h on torch tensors: several ranks behind one C call (csrc/ws_tiled.hip).

A `Group` is LOCAL (all ranks driven by this process, rank r on `devices[r]`; ranks may share a device) or RCCL (this
process is one rank of `world`, one GPU per process: the 128-byte id comes from rank 0 and travels through whatever the
job already has -- here torch.distributed.broadcast).  The loop that distributed.py spells out over torch.distributed
runs inside the library; torch only allocates the tensors.
"""
import ctypes

import torch

from . import _ffi


class Group:
    def __init__(self, handle, world, n_local, first_local):
        self._h = handle
        self.world, self.n_local, self.first_local = world, n_local, first_local

    # ---- construction ----------------------------------------------------------------------------------------------------
    @classmethod
    def local(cls, n_ranks, devices=None):
        h = ctypes.c_void_p()
        dev = (ctypes.c_int * n_ranks)(*devices) if devices is not None else None
        rc = _ffi.lib().ws_group_create_local(n_ranks, dev, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"ws_group_create_local: {_ffi.lib().ws_strerror(rc).decode()}")
        return cls._wrap(h)

    @classmethod
    def rccl(cls, device, rank, world, broadcast_bytes):
        """`broadcast_bytes(buf_or_None) -> bytes`: hands rank 0's 128 id bytes to every rank (rank 0 passes them in)."""
        L = _ffi.lib()
        uid = ctypes.create_string_buffer(_ffi.WS_RCCL_ID_BYTES)
        if rank == 0:
            rc = L.ws_group_rccl_unique_id(uid)
            if rc != 0:
                raise RuntimeError(f"ws_group_rccl_unique_id: {L.ws_strerror(rc).decode()}: {L.ws_group_last_error(None).decode()}")
        raw = broadcast_bytes(bytes(uid.raw) if rank == 0 else None)
        uid = ctypes.create_string_buffer(raw, _ffi.WS_RCCL_ID_BYTES)
        h = ctypes.c_void_p()
        rc = L.ws_group_create_rccl(device, rank, world, uid, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"ws_group_create_rccl: {L.ws_strerror(rc).decode()}: {L.ws_group_last_error(None).decode()}")
        return cls._wrap(h)

    @classmethod
    def rccl_over_torch(cls, device, rank, world, comm_device):
        """The id travels as a 128-byte tensor through torch.distributed (any backend)."""
        import torch.distributed as dist

        def bcast(raw):
            t = torch.zeros(_ffi.WS_RCCL_ID_BYTES, dtype=torch.uint8, device=comm_device)
            if raw is not None:
                t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            return bytes(t.cpu().numpy().tobytes())
        return cls.rccl(device, rank, world, bcast)

    @classmethod
    def _wrap(cls, h):
        w, n, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _ffi.lib().ws_group_info(h, ctypes.byref(w), ctypes.byref(n), ctypes.byref(f))
        return cls(h, w.value, n.value, f.value)

    def close(self):
        if self._h:
            _ffi.lib().ws_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {_ffi.lib().ws_strerror(rc).decode()}: {_ffi.lib().ws_group_last_error(self._h).decode()}")

    def selftest(self):
        self._check(_ffi.lib().ws_group_selftest(self._h), "ws_group_selftest")

    # ---- one field in row blocks ---------------------------------------------------------------------------------------------
    @staticmethod
    def tile_rows(h, rank, world):
        v = [ctypes.c_size_t() for _ in range(4)]
        rc = _ffi.lib().ws_tile_rows(h, rank, world, *[ctypes.byref(x) for x in v])
        if rc != 0:
            raise ValueError(f"a field of {h} rows cannot be split into {world} row blocks")
        return tuple(x.value for x in v)

    def make_blocks(self, field_rows, img_rows_of, seeds, devices=None):
        """Descriptors of the LOCAL ranks' blocks.  `img_rows_of(lo, hi, rank)`: the rank's rows [lo, hi) of the field as a
        contiguous uint8 tensor on the rank's device; `seeds`: the field's strictly increasing (row, col) list, int32 (n, 2),
        on any device.  Returns (ctypes array, [(r0, r1, lo, labels)], keep-alive list)."""
        rows = seeds[:, 0].contiguous()
        blocks = (_ffi.TileBlock * self.n_local)()
        spans, keep = [], []
        for i in range(self.n_local):
            rank = self.first_local + i
            r0, r1, lo, hi = self.tile_rows(field_rows, rank, self.world)
            b = torch.searchsorted(rows, torch.tensor([lo, hi], dtype=rows.dtype, device=rows.device))
            i0, i1 = int(b[0]), int(b[1])
            img = img_rows_of(lo, hi, rank)
            loc = seeds[i0:i1].to(img.device).clone()
            loc[:, 0] -= lo
            lab = torch.empty((hi - lo, img.shape[1]), dtype=torch.int32, device=img.device)
            keep += [img, loc, lab]
            spans.append((r0, r1, lo, lab))
            blocks[i] = _ffi.TileBlock(img.data_ptr(), loc.data_ptr() if i1 > i0 else None, None, i1 - i0, i0 + 1, 0, lab.data_ptr())
        torch.cuda.synchronize()      # the blocks were cut on torch's streams; the group's contexts run on streams of their own
        return blocks, spans, keep

    def segment_tiled_device(self, field_rows, width, n_seeds_total, blocks, max_level=254, merging=False):
        opt = _ffi.Options(max_level)
        rounds = ctypes.c_uint32(0)
        self._check(_ffi.lib().ws_segment_tiled_device(self._h, field_rows, width, n_seeds_total, blocks, ctypes.byref(opt), int(merging),
                                                       ctypes.byref(rounds)), "ws_segment_tiled_device")
        return rounds.value

    def transform_to_list_tiled_device(self, field_rows, width, n_seeds_total, blocks, lakes, max_level=254, merging=True):
        """transform_to_list of a field in row blocks (ws_transform_to_list_tiled_device): `blocks` as for segment_tiled_device,
        `lakes` an int64 (cap, 2) tensor on rank 0's device for the (colour, area) records.  Returns (n_lakes, offsets numpy
        (levels + 1), uncoloured numpy (levels), rounds) -- meaningful in the process that holds rank 0."""
        import numpy as np
        opt = _ffi.Options(max_level)
        rounds = ctypes.c_uint32(0)
        n = ctypes.c_size_t(0)
        offsets = np.zeros(max_level + 2, dtype=np.uint64)
        uncoloured = np.zeros(max_level + 1, dtype=np.uint64)
        self._check(_ffi.lib().ws_transform_to_list_tiled_device(self._h, field_rows, width, n_seeds_total, blocks, ctypes.byref(opt), int(merging),
                                                                 lakes.data_ptr() if lakes is not None and lakes.numel() else None,
                                                                 lakes.shape[0] if lakes is not None else 0, ctypes.byref(n),
                                                                 offsets.ctypes.data, uncoloured.ctypes.data, ctypes.byref(rounds)),
                    "ws_transform_to_list_tiled_device")
        return n.value, offsets, uncoloured, rounds.value

    # ---- one field in py x px tiles (both directions) ------------------------------------------------------------------------------
    @staticmethod
    def tile_grid(h, w, rank, py, px):
        rows, cols = (ctypes.c_size_t * 4)(), (ctypes.c_size_t * 4)()
        rc = _ffi.lib().ws_tile_grid(h, w, rank, py, px, rows, cols)
        if rc != 0:
            raise ValueError(f"a {h} x {w} field cannot be cut into {py} x {px} tiles")
        return tuple(rows), tuple(cols)

    def make_blocks2d(self, field, seeds, py, px):
        """Descriptors of the LOCAL ranks' tiles of `field` (uint8 (H, W) tensor on the ranks' device: a tile's image is a view
        into it) with the global seed list `seeds` (int32 (n, 2), any order).  Returns (ctypes array, [(rows, cols, labels)], keep)."""
        H, W = field.shape
        blocks = (_ffi.TileBlock2D * self.n_local)()
        spans, keep = [], []
        colours_all = torch.arange(1, seeds.shape[0] + 1, dtype=torch.int32, device=seeds.device)
        for i in range(self.n_local):
            rows, cols = self.tile_grid(H, W, self.first_local + i, py, px)
            (r0, r1, lo, hi), (c0, c1, clo, chi) = rows, cols
            view = field[lo:hi, clo:chi]
            inside = (seeds[:, 0] >= lo) & (seeds[:, 0] < hi) & (seeds[:, 1] >= clo) & (seeds[:, 1] < chi)
            loc = seeds[inside].to(field.device).clone()
            loc[:, 0] -= lo
            loc[:, 1] -= clo
            col = colours_all[inside].to(field.device).contiguous()
            lab = torch.empty((hi - lo, chi - clo), dtype=torch.int32, device=field.device)
            keep += [view, loc, col, lab]
            spans.append((rows, cols, lab))
            ns = int(loc.shape[0])
            blocks[i] = _ffi.TileBlock2D(view.data_ptr(), field.stride(0), loc.data_ptr() if ns else None, col.data_ptr() if ns else None, ns, lab.data_ptr())
        torch.cuda.synchronize()
        return blocks, spans, keep

    def segment_tiled2d_device(self, field_h, field_w, py, px, blocks, max_level=254, n_seeds_total=0, merging=False):
        opt = _ffi.Options(max_level)
        rounds = ctypes.c_uint32(0)
        self._check(_ffi.lib().ws_segment_tiled2d_device(self._h, field_h, field_w, py, px, n_seeds_total, blocks, ctypes.byref(opt), int(merging),
                                                         ctypes.byref(rounds)), "ws_segment_tiled2d_device")
        return rounds.value

    # ---- a batch of independent slices ------------------------------------------------------------------------------------------
    def segment_batch(self, h, w, parts, max_level=254):
        """parts: one (cube (S, h, w) uint8, seeds (n, 2) int32, offsets list of S + 1, labels (S, h, w) int32) per LOCAL rank."""
        arr = (_ffi.BatchPart * self.n_local)()
        keep = []
        for i, (cube, seeds, offs, labels) in enumerate(parts):
            co = (ctypes.c_size_t * len(offs))(*[int(x) for x in offs])
            keep.append(co)
            arr[i] = _ffi.BatchPart(cube.data_ptr() if cube.numel() else None, seeds.data_ptr() if seeds.numel() else None, co, cube.shape[0],
                                    labels.data_ptr() if labels.numel() else None)
        fr, fs = ctypes.c_size_t(), ctypes.c_size_t()
        opt = _ffi.Options(max_level)
        self._check(_ffi.lib().ws_segment_batch_group(self._h, h, w, arr, ctypes.byref(opt), ctypes.byref(fr), ctypes.byref(fs)), "ws_segment_batch_group")

    def segment_batch_host(self, cube, seeds=None, max_level=254, edge=False):
        """A cube (S, h, w) of uint8 slices in HOST memory (numpy) over the ranks of the group (ws_segment_batch_host): rank r takes
        a block of the slices and pipelines them on its device.  seeds: None (every slice's find_local_minima) or one (n, 2) list
        per slice.  Returns the uint64 label cube (and the minima counts with seeds None)."""
        import numpy as np
        c = np.ascontiguousarray(cube, dtype=np.uint8)
        n, h, w = c.shape
        e = 2 if edge else 0
        out = np.zeros((n, h + e, w + e), dtype=np.uint64)
        counts = np.zeros(max(n, 1), dtype=np.uintp)
        flat = offs = None
        if seeds is not None:
            lists = [np.asarray(s, dtype=np.uint64).reshape(-1, 2) for s in seeds]
            offs = np.zeros(n + 1, dtype=np.uintp)
            offs[1:] = np.cumsum([len(l) for l in lists])
            flat = np.ascontiguousarray(np.concatenate(lists + [np.zeros((1, 2), dtype=np.uint64)], axis=0))
        fs = ctypes.c_size_t()
        opt = _ffi.Options(max_level, int(edge))
        self._check(_ffi.lib().ws_segment_batch_host(self._h, c.ctypes.data, n, h, w, w, h * w, flat.ctypes.data if flat is not None else None,
                                                     offs.ctypes.data_as(_ffi.szp) if offs is not None else None, ctypes.byref(opt), out.ctypes.data,
                                                     counts.ctypes.data_as(_ffi.szp), ctypes.byref(fs)), "ws_segment_batch_host")
        return out if seeds is not None else (out, counts[:n].astype(np.int64))
