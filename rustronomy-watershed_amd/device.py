"""Device-resident entry points over PyTorch-ROCm tensors.

PyTorch is plumbing here (HBM allocations, the current HIP stream, torch.distributed);
all compute is the HIP engine behind include/ws_hip.h.  Tensors hold raw bits: label and
seed planes are int32 tensors carrying the uint32 values of the C ABI.
"""
import ctypes

import torch

from . import _ffi
from .api import Context, ENGINE_AUTO


class DeviceEngine:
    """One engine context on torch's CURRENT stream of the device.  Create it under a stream of your own
    (`torch.cuda.set_stream(torch.cuda.Stream(dev))`, as bench.py does) rather than on the legacy default stream: a repeated
    transform is replayed as one hipGraph, and HIP cannot capture on the legacy stream -- there every transform is its eleven
    stream operations (2048^2: 0.155 instead of 0.123 ms; results are the same)."""

    def __init__(self, device_index=0, engine=ENGINE_AUTO):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceEngine needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        # enqueue on torch's current stream so torch events bracket the engine's kernels
        self.stream = torch.cuda.current_stream(self.device)
        self.ctx = Context(device_index, stream=self.stream.cuda_stream)
        self.engine = engine

    def options(self, max_level=254, edge=False, engine=None, seed_shift=False):
        return _ffi.Options(max_level, int(edge), self.engine if engine is None else engine, 0, int(seed_shift))

    def random_field(self, h, w, seed):
        img = torch.empty((h, w), dtype=torch.uint8, device=self.device)
        self.ctx.check(_ffi.lib().ws_random_field_device(self.ctx.handle, img.data_ptr(), h, w, w, seed))
        return img

    def find_local_minima(self, img):
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        h, w = img.shape
        cap = ((max(h, 1) - 1) // 2 + 1) * ((max(w, 1) - 1) // 2 + 1)
        out = torch.empty((max(cap, 1), 2), dtype=torch.int32, device=self.device)
        n = ctypes.c_size_t(0)
        self.ctx.check(_ffi.lib().ws_find_local_minima_device(self.ctx.handle, img.data_ptr(), h, w, w, out.data_ptr(),
                                                              cap, ctypes.byref(n)))
        return out[: n.value]

    def _plane(self, img, edge):
        e = 2 if edge else 0
        return img.shape[0] + e, img.shape[1] + e

    def segment(self, img, seeds, max_level=254, edge=False, engine=None, out=None, seed_shift=False):
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        assert seeds.dtype == torch.int32 and seeds.is_cuda and (seeds.numel() == 0 or seeds.is_contiguous())
        h, w = img.shape
        if out is None:
            out = torch.empty(self._plane(img, edge), dtype=torch.int32, device=self.device)
        opt = self.options(max_level, edge, engine, seed_shift)
        ns = seeds.shape[0] if seeds.dim() == 2 else 0
        self.ctx.check(_ffi.lib().ws_segment_device(self.ctx.handle, img.data_ptr(), h, w, w,
                                                    seeds.data_ptr() if ns else None, ns, ctypes.byref(opt),
                                                    out.data_ptr()))
        return out

    def segment_minima(self, img, max_level=254, edge=False, engine=None, out=None, want_seeds=False):
        """find_local_minima + segment as one call (ws_segment_minima_device): labels, n_seeds[, seeds]."""
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        h, w = img.shape
        if out is None:
            out = torch.empty(self._plane(img, edge), dtype=torch.int32, device=self.device)
        opt = self.options(max_level, edge, engine, False)
        cap = ((max(h, 1) - 1) // 2 + 1) * ((max(w, 1) - 1) // 2 + 1) if want_seeds else 0
        seeds = torch.empty((max(cap, 1), 2), dtype=torch.int32, device=self.device) if want_seeds else None
        n = ctypes.c_size_t(0)
        self.ctx.check(_ffi.lib().ws_segment_minima_device(self.ctx.handle, img.data_ptr(), h, w, w, ctypes.byref(opt), out.data_ptr(),
                                                           seeds.data_ptr() if want_seeds else None, cap, ctypes.byref(n)))
        return (out, n.value, seeds[: n.value]) if want_seeds else (out, n.value)

    def segment_begin(self, img, seeds, out, max_level=254, edge=False, seed_shift=False):
        """First half of segment() (ws_segment_device_begin): queues the transform and returns.  The engine is busy until
        segment_end(); img, seeds and out must stay alive and unchanged until then.  Two engines taking turns keep the
        GPU fed between transforms."""
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        assert seeds.dtype == torch.int32 and seeds.is_cuda and (seeds.numel() == 0 or seeds.is_contiguous())
        assert out.dtype == torch.int32 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == self._plane(img, edge)
        h, w = img.shape
        opt = self.options(max_level, edge, None, seed_shift)
        ns = seeds.shape[0] if seeds.dim() == 2 else 0
        self._pending = (img, seeds, out)
        self.ctx.check(_ffi.lib().ws_segment_device_begin(self.ctx.handle, img.data_ptr(), h, w, w,
                                                          seeds.data_ptr() if ns else None, ns, ctypes.byref(opt),
                                                          out.data_ptr()))
        return out

    def segment_end(self):
        """Second half: waits for the transform begun with segment_begin() and raises if it failed."""
        pending, self._pending = getattr(self, "_pending", None), None
        self.ctx.check(_ffi.lib().ws_segment_device_end(self.ctx.handle))
        return pending[2] if pending else None

    def segment_batch(self, cube, seeds, seed_offsets, max_level=254, edge=False, out=None, seed_shift=False):
        """A stack of independent slices (config C4).  cube: (S, H, W) uint8; seeds: all slices' (row, col) pairs
        concatenated, int32 (n, 2); seed_offsets: S + 1 host integers.  Returns (S, H', W') int32 labels."""
        assert cube.dtype == torch.uint8 and cube.dim() == 3 and cube.is_contiguous() and cube.is_cuda
        assert seeds.dtype == torch.int32 and seeds.is_cuda and (seeds.numel() == 0 or seeds.is_contiguous())
        s, h, w = cube.shape
        assert len(seed_offsets) == s + 1
        e = 2 if edge else 0
        if out is None:
            out = torch.empty((s, h + e, w + e), dtype=torch.int32, device=self.device)
        opt = self.options(max_level, edge, None, seed_shift)
        offs = (ctypes.c_size_t * (s + 1))(*[int(x) for x in seed_offsets])
        failed = ctypes.c_size_t(0)
        self.ctx.check(_ffi.lib().ws_segment_batch_device(self.ctx.handle, cube.data_ptr(), s, h, w, w, h * w,
                                                          seeds.data_ptr() if seeds.numel() else None, offs,
                                                          ctypes.byref(opt), out.data_ptr(), ctypes.byref(failed)))
        return out

    def merge(self, img, seeds, max_level=254, edge=False, out=None, seed_shift=False):
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        h, w = img.shape
        if out is None:
            out = torch.empty(self._plane(img, edge), dtype=torch.int32, device=self.device)
        opt = self.options(max_level, edge, None, seed_shift)
        ns = seeds.shape[0] if seeds.dim() == 2 else 0
        self.ctx.check(_ffi.lib().ws_merge_device(self.ctx.handle, img.data_ptr(), h, w, w,
                                                  seeds.data_ptr() if ns else None, ns, ctypes.byref(opt),
                                                  out.data_ptr()))
        return out

    def merge_begin(self, img, seeds, out, max_level=254, edge=False, seed_shift=False):
        """First half of merge() (ws_merge_device_begin); rules as segment_begin()."""
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        assert out.dtype == torch.int32 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == self._plane(img, edge)
        h, w = img.shape
        opt = self.options(max_level, edge, None, seed_shift)
        ns = seeds.shape[0] if seeds.dim() == 2 else 0
        self._pending = (img, seeds, out)
        self.ctx.check(_ffi.lib().ws_merge_device_begin(self.ctx.handle, img.data_ptr(), h, w, w,
                                                        seeds.data_ptr() if ns else None, ns, ctypes.byref(opt),
                                                        out.data_ptr()))
        return out

    def merge_end(self):
        pending, self._pending = getattr(self, "_pending", None), None
        self.ctx.check(_ffi.lib().ws_merge_device_end(self.ctx.handle))
        return pending[2] if pending else None

    def last_arrival(self):
        """Arrival stamps (level << 24 | ring) of the last fused-engine call, as a tensor copy."""
        p = ctypes.c_void_p()
        h = ctypes.c_size_t()
        w = ctypes.c_size_t()
        self.ctx.check(_ffi.lib().ws_last_arrival_device(self.ctx.handle, ctypes.byref(p), ctypes.byref(h), ctypes.byref(w)))
        out = torch.empty((h.value, w.value), dtype=torch.int32, device=self.device)
        self.ctx.check(_ffi.lib().ws_copy_last_arrival_device(self.ctx.handle, out.data_ptr(), out.numel()))
        return out

    def transform_to_list(self, img, seeds, merging=True, max_level=254, edge=False, lakes=None):
        """transform_to_list with the records left in HBM: returns (lakes (n, 2) int64 tensor of (colour, area) on the device,
        offsets numpy (levels + 1), uncoloured numpy (levels)).  `lakes`: a reusable (cap, 2) int64 device buffer."""
        import numpy as np
        assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous() and img.is_cuda
        h, w = img.shape
        ns = seeds.shape[0] if seeds.dim() == 2 else 0
        levels = max_level + 1
        opt = self.options(max_level, edge)
        offsets = np.zeros(levels + 1, dtype=np.uint64)
        unc = np.zeros(levels, dtype=np.uint64)
        n = ctypes.c_size_t(0)
        cap = int(lakes.shape[0]) if lakes is not None else max(ns, 1) * levels // 2 + 1024
        while True:
            if lakes is None or lakes.shape[0] < cap:
                lakes = torch.empty((cap, 2), dtype=torch.int64, device=self.device)
            rc = _ffi.lib().ws_transform_to_list_device(self.ctx.handle, int(merging), img.data_ptr(), h, w, w,
                                                        seeds.data_ptr() if ns else None, ns, ctypes.byref(opt), lakes.data_ptr(), cap,
                                                        ctypes.byref(n), offsets.ctypes.data, unc.ctypes.data)
            if rc == _ffi.WS_ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            self.ctx.check(rc)
            break
        return lakes[: n.value], offsets, unc

    def level_snapshot(self, labels, water_level, out=None):
        """The segmenting label plane after `water_level` (transform_history's entry for that level), on the device."""
        if out is None:
            out = torch.empty_like(labels)
        self.ctx.check(_ffi.lib().ws_level_snapshot_device(self.ctx.handle, labels.data_ptr(), int(water_level), out.data_ptr()))
        return out

    def stats(self):
        return self.ctx.stats()
