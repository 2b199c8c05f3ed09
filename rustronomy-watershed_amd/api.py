"""Host-side mirror of the reference's public interface for the watershed path.

Same names, argument meaning and error behaviour as rustronomy-watershed v0.4.1
(`TransformBuilder`, `BuildErr`, `HookCtx`, `Watershed` trait methods, `WatershedUtils::
find_local_minima`; "lib.rs:N" = src/lib.rs line N of the reference), implemented over the
C-ABI in include/ws_hip.h.  All compute happens in the HIP engine; this file only moves
numpy arrays across the boundary.

Documented deviations:
  * tie-break where lakes meet: first coloured neighbour in down,right,left,up order (the
    reference picks at random among them, lib.rs:249-253);
  * `SegmentingWatershed.transform` returns the labels after the last level (the reference
    panics at lib.rs:1821);
  * merged-lake ids are canonical (smallest seed colour in the lake) where the reference's are
    arbitrary.
"""
import ctypes

import numpy as np

from . import _ffi

UNCOLOURED = 0          # lib.rs:138
NORMAL_MAX = 254        # lib.rs:139
ALWAYS_FILL = 0         # lib.rs:140
NEVER_FILL = 255        # lib.rs:141

ENGINE_AUTO, ENGINE_FUSED, ENGINE_SWEEP = _ffi.WS_ENGINE_AUTO, _ffi.WS_ENGINE_FUSED, _ffi.WS_ENGINE_SWEEP


class WatershedError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _ffi.lib().ws_strerror(status).decode()
        super().__init__(f"{msg} ({status})" + (f": {detail}" if detail else ""))


class BuildErr(ValueError):
    """lib.rs:1051-1065"""


class MaxToHigh(BuildErr):
    def __init__(self, v):
        self.value = v
        super().__init__(f"Maximum water level set to {v}, which is higher than the maximum allowed value {NORMAL_MAX}")


class MaxToLow(BuildErr):
    def __init__(self, v):
        self.value = v
        # the reference's message names NEVER_FILL here (lib.rs:1062); kept verbatim in meaning
        super().__init__(f"Maximum water level set to {v}, which is lower than the minimum allowed value {NEVER_FILL}")


class SeedOutOfBounds(IndexError):
    """the reference panics with an ndarray index error (lib.rs:1366 / 1676)"""


class HookCtx:
    """lib.rs:844-862"""
    __slots__ = ("water_level", "max_water_level", "image", "colours", "seeds")

    def __init__(self, water_level, max_water_level, image, colours, seeds):
        self.water_level = water_level
        self.max_water_level = max_water_level
        self.image = image
        self.colours = colours
        self.seeds = seeds


class Context:
    """One ws_ctx: a HIP stream plus reusable device workspaces.  Not thread safe."""

    def __init__(self, device=0, stream=None):
        self._h = ctypes.c_void_p()
        L = _ffi.lib()
        rc = (L.ws_ctx_create(device, ctypes.byref(self._h)) if stream is None
              else L.ws_ctx_create_on_stream(device, ctypes.c_void_p(stream), ctypes.byref(self._h)))
        if rc != _ffi.WS_OK:
            self._h = ctypes.c_void_p()
            raise WatershedError(rc, "ws_ctx_create")
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            _ffi.lib().ws_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == _ffi.WS_OK:
            return
        detail = _ffi.lib().ws_last_error(self._h).decode()
        if rc == _ffi.WS_ERR_SEED_OOB:
            raise SeedOutOfBounds(detail)
        raise WatershedError(rc, detail)

    def set_profiling(self, on):
        self.check(_ffi.lib().ws_ctx_set_profiling(self._h, int(on)))

    def stats(self):
        st = _ffi.Stats()
        self.check(_ffi.lib().ws_ctx_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def synchronize(self):
        self.check(_ffi.lib().ws_ctx_synchronize(self._h))

    def set_batch_pixel_limit(self, max_px):
        self.check(_ffi.lib().ws_ctx_set_batch_pixel_limit(self._h, int(max_px)))

    def set_seam_repair_min_pixels(self, min_px):
        self.check(_ffi.lib().ws_ctx_set_seam_repair_min_pixels(self._h, int(min_px)))


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _as_image(img):
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise TypeError("input must be a 2-D uint8 array (ndarray::ArrayView2<u8>)")
    if a.strides[1] != 1 or (a.shape[0] > 1 and a.strides[0] < a.shape[1]):
        a = np.ascontiguousarray(a)          # the Rust shim calls as_standard_layout() likewise
    stride = a.strides[0] if a.shape[0] > 1 else max(a.shape[1], 1)
    return a, int(stride)


def _as_seeds(seeds):
    s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1, 2))
    return s, int(s.shape[0])


class TransformBuilder:
    """lib.rs:908-1047.  `TransformBuilder()` is both `new()` and `default()`."""

    def __init__(self):
        self.max_water_level = NORMAL_MAX     # lib.rs:942
        self.edge_correction = False          # lib.rs:943
        self.wlvl_hook = None                 # lib.rs:944
        self.engine = ENGINE_AUTO
        self.context = None
        self.seed_shift = False               # lib.rs:1675-1677: seeds are NOT moved into the padded plane

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def default(cls):
        return cls()

    def set_max_water_lvl(self, max_water_lvl):          # lib.rs:950
        if not 0 <= int(max_water_lvl) <= 255:
            raise OverflowError("max_water_lvl must fit a u8")
        self.max_water_level = int(max_water_lvl)
        return self

    def enable_edge_correction(self):                    # lib.rs:958
        self.edge_correction = True
        return self

    def set_wlvl_hook(self, hook):                       # lib.rs:967
        self.wlvl_hook = hook
        return self

    # not in the reference: engine / context selection of this implementation
    def set_engine(self, engine):
        self.engine = engine
        return self

    def set_context(self, ctx):
        self.context = ctx
        return self

    def shift_seeds_into_padded_plane(self, on=True):
        """Not in the reference (ws_options.seed_shift): with edge correction, move every seed by (+1, +1) onto the
        pixel it was found at, instead of indexing the padded plane with the caller's coordinates (lib.rs:1675-1677)."""
        self.seed_shift = bool(on)
        return self

    def _validate(self):
        opt = _ffi.Options(self.max_water_level, int(self.edge_correction), self.engine, 0, int(self.seed_shift))
        rc = _ffi.lib().ws_options_validate(ctypes.byref(opt))
        if rc == _ffi.WS_ERR_MAX_TOO_HIGH:
            raise MaxToHigh(self.max_water_level)         # lib.rs:1026-1027
        if rc == _ffi.WS_ERR_MAX_TOO_LOW:
            raise MaxToLow(self.max_water_level)          # lib.rs:1028-1029
        if rc != _ffi.WS_OK:
            raise WatershedError(rc, "ws_options_validate")
        return opt

    def build_segmenting(self):                           # lib.rs:1024-1046
        return SegmentingWatershed(self._validate(), self.wlvl_hook, self.context)

    def build_merging(self):                              # lib.rs:998-1020
        return MergingWatershed(self._validate(), self.wlvl_hook, self.context)


class WatershedUtils:
    """lib.rs:1069-1198"""

    def pre_processor(self, img):
        """lib.rs:1081-1087: any numeric array -> u8 in [0, NORMAL_MAX]; NaN / -inf / subnormals / exact 0 ->
        NEVER_FILL, +inf -> ALWAYS_FILL (as the code does, whatever its comments say)."""
        return self.pre_processor_with_max(img, NORMAL_MAX)

    def pre_processor_with_max(self, img, max_value):
        """lib.rs:1134-1173 (`pre_processor_with_max::<MAX, _, _>`); any dimension."""
        a = np.ascontiguousarray(img)
        if a.dtype.name not in _ffi.WS_DTYPES:
            raise TypeError(f"unsupported dtype {a.dtype} (supported: {sorted(_ffi.WS_DTYPES)})")
        if not 0 <= int(max_value) <= 255:
            raise OverflowError("MAX must fit a u8")
        if int(max_value) >= NEVER_FILL or int(max_value) <= ALWAYS_FILL:
            raise AssertionError("MAX must be in 1..=254 (lib.rs:1143-1144)")      # the reference asserts
        ctx = self._ctx()
        out = np.empty(a.shape, dtype=np.uint8)
        rc = _ffi.lib().ws_pre_processor(ctx.handle, a.ctypes.data, _ffi.WS_DTYPES[a.dtype.name], a.size, int(max_value),
                                         out.ctypes.data)
        ctx.check(rc)
        return out

    def find_local_minima(self, img):
        """Strict 8-neighbour local MAXIMA of the interior, row-major (lib.rs:1178-1197).
        Returns an (n, 2) uint64 array of (row, col)."""
        a, stride = _as_image(img)
        h, w = a.shape
        ctx = self._ctx()
        cap = ((max(h, 1) - 1) // 2 + 1) * ((max(w, 1) - 1) // 2 + 1)
        out = np.empty((max(cap, 1), 2), dtype=np.uint64)
        n = ctypes.c_size_t(0)
        rc = _ffi.lib().ws_find_local_minima(ctx.handle, a.ctypes.data, h, w, stride, out.ctypes.data, cap,
                                             ctypes.byref(n))
        ctx.check(rc)
        return out[: n.value].copy()


class _Transform(WatershedUtils):
    _merging = False

    def __init__(self, opt, hook, ctx):
        self._opt = opt
        self.max_water_level = opt.max_water_level
        self.edge_correction = bool(opt.edge_correction)
        self.wlvl_hook = hook
        self._context = ctx

    def _ctx(self):
        return self._context if self._context is not None else default_context()

    def _shape(self, a):
        e = 2 if self.edge_correction else 0
        return a.shape[0] + e, a.shape[1] + e

    def _run_with_hook(self, img, seeds, hook, want_final):
        a, stride = _as_image(img)
        s, ns = _as_seeds(seeds)
        h, w = a.shape
        ph, pw = self._shape(a)
        ctx = self._ctx()
        results = []
        seed_colours = None
        cb = None
        if hook is not None:
            seed_colours = [(i + 1, (int(r), int(c))) for i, (r, c) in enumerate(s)]   # lib.rs:1671-1672

            def _cb(_user, lvl, mx, pimg, plab, hh, ww):
                image = np.ctypeslib.as_array(pimg, shape=(hh, ww))
                colours = np.ctypeslib.as_array(plab, shape=(hh, ww))
                results.append(hook(HookCtx(lvl, mx, image, colours, seed_colours)))
            cb = _ffi.LEVEL_CB(_cb)
        out = np.empty((ph, pw), dtype=np.uint64) if want_final else None
        fn = _ffi.lib().ws_merge_with_hook if self._merging else _ffi.lib().ws_segment_with_hook
        rc = fn(ctx.handle, a.ctypes.data, h, w, stride, s.ctypes.data, ns, ctypes.byref(self._opt),
                ctypes.cast(cb, ctypes.c_void_p) if cb else None, None, out.ctypes.data if want_final else None)
        ctx.check(rc)
        return results, out

    def transform_with_hook(self, input, seeds):          # lib.rs:1214
        if self.wlvl_hook is None:
            # the reference still runs the whole transform and returns an empty Vec (lib.rs:1796-1807)
            self._run_with_hook(input, seeds, None, False)
            return []
        return self._run_with_hook(input, seeds, self.wlvl_hook, False)[0]

    def transform_history(self, input, seeds):            # lib.rs:1233-1237, 1538-1549, 1824-1835
        return self._run_with_hook(input, seeds, lambda ctx: (ctx.water_level, ctx.colours.copy()), False)[0]

    def transform_to_list(self, input, seeds):            # lib.rs:1220-1224, 1551-1561, 1837-1847
        """[(level, lake_sizes)] with lake_sizes a uint64 vector of length pixels+1 (lib.rs:630)."""
        a, stride = _as_image(input)
        s, ns = _as_seeds(seeds)
        h, w = a.shape
        ph, pw = self._shape(a)
        ctx = self._ctx()
        levels = self.max_water_level + 1
        offsets = np.zeros(levels + 1, dtype=np.uint64)
        unc = np.zeros(levels, dtype=np.uint64)
        n = ctypes.c_size_t(0)
        # one record per live lake and level, at most ns * levels; half of that covers a random field (83 per seed
        # at 1024^2).  A too small guess costs a second transform, not a wrong answer.
        cap = min(max(ns, 1) * (self.max_water_level + 1) // 2 + 1024, 1 << 26)
        while True:
            lakes = np.empty((cap, 2), dtype=np.uint64)
            rc = _ffi.lib().ws_transform_to_list(ctx.handle, int(self._merging), a.ctypes.data, h, w, stride,
                                                 s.ctypes.data, ns, ctypes.byref(self._opt), lakes.ctypes.data, cap,
                                                 ctypes.byref(n), offsets.ctypes.data, unc.ctypes.data)
            if rc == _ffi.WS_ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            ctx.check(rc)
            break
        out = []
        for lvl in range(levels):
            hist = np.zeros(ph * pw + 1, dtype=np.uint64)
            lo, hi = int(offsets[lvl]), int(offsets[lvl + 1])
            hist[lakes[lo:hi, 0].astype(np.int64)] = lakes[lo:hi, 1]
            hist[0] = unc[lvl]
            out.append((lvl, hist))
        return out

    def transform_to_list_sparse(self, input, seeds):
        """Same content as transform_to_list without the dense h*w+1 vectors:
        [(level, uncoloured, colours[], areas[])]."""
        a, stride = _as_image(input)
        s, ns = _as_seeds(seeds)
        h, w = a.shape
        ctx = self._ctx()
        levels = self.max_water_level + 1
        offsets = np.zeros(levels + 1, dtype=np.uint64)
        unc = np.zeros(levels, dtype=np.uint64)
        n = ctypes.c_size_t(0)
        # one record per live lake and level, at most ns * levels; half of that covers a random field (83 per seed
        # at 1024^2).  A too small guess costs a second transform, not a wrong answer.
        cap = min(max(ns, 1) * (self.max_water_level + 1) // 2 + 1024, 1 << 26)
        while True:
            lakes = np.empty((cap, 2), dtype=np.uint64)
            rc = _ffi.lib().ws_transform_to_list(ctx.handle, int(self._merging), a.ctypes.data, h, w, stride,
                                                 s.ctypes.data, ns, ctypes.byref(self._opt), lakes.ctypes.data, cap,
                                                 ctypes.byref(n), offsets.ctypes.data, unc.ctypes.data)
            if rc == _ffi.WS_ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            ctx.check(rc)
            break
        # views into one record array (no per-level copies: 155 MB of them at 1024^2)
        off = offsets.astype(np.int64)
        return [(lvl, int(unc[lvl]), lakes[off[lvl]:off[lvl + 1], 0], lakes[off[lvl]:off[lvl + 1], 1]) for lvl in range(levels)]


class SegmentingWatershed(_Transform):
    """lib.rs:1609-1849"""
    _merging = False

    def transform(self, input, seeds):                    # lib.rs:1810-1822 (intended semantics)
        a, stride = _as_image(input)
        s, ns = _as_seeds(seeds)
        h, w = a.shape
        ctx = self._ctx()
        out = np.empty(self._shape(a), dtype=np.uint64)
        rc = _ffi.lib().ws_segment(ctx.handle, a.ctypes.data, h, w, stride, s.ctypes.data, ns, ctypes.byref(self._opt),
                                   out.ctypes.data)
        ctx.check(rc)
        return out

    def transform_from_minima(self, input, want_seeds=False, labels_u32=False):
        """`self.transform(input, &self.find_local_minima(input))` -- the README's call pair, lib.rs:73-86 -- as ONE call of the
        library (ws_segment_minima): the seed list never crosses PCIe unless want_seeds.  Not a method of the reference."""
        a, stride = _as_image(input)
        h, w = a.shape
        ctx = self._ctx()
        out = np.empty(self._shape(a), dtype=np.uint32 if labels_u32 else np.uint64)
        cap = ((max(h, 1) - 1) // 2 + 1) * ((max(w, 1) - 1) // 2 + 1) if want_seeds else 0
        seeds = np.empty((max(cap, 1), 2), dtype=np.uint64) if want_seeds else None
        n = ctypes.c_size_t(0)
        fn = _ffi.lib().ws_segment_minima_u32 if labels_u32 else _ffi.lib().ws_segment_minima
        ctx.check(fn(ctx.handle, a.ctypes.data, h, w, stride, ctypes.byref(self._opt), out.ctypes.data,
                     seeds.ctypes.data if want_seeds else None, cap, ctypes.byref(n)))
        return (out, seeds[: n.value].copy()) if want_seeds else out

    def transform_cube(self, cube, seeds=None):
        """The slices cube[k] of a 3-D u8 array, each as `transform(cube[k], seeds[k])` -- or, with seeds None, as the README's
        pair `transform(cube[k], &find_local_minima(cube[k]))` -- in ONE call of the library (ws_segment_batch: the slices'
        uploads, transforms and label copies overlap).  What tests/integration.rs:267,356 loops over.  Returns the label cube
        (and, with seeds None, the number of minima of every slice).  Not a method of the reference."""
        c = np.ascontiguousarray(cube, dtype=np.uint8)
        if c.ndim != 3:
            raise ValueError("cube must be 3-D: (slices, rows, columns)")
        n, h, w = c.shape
        ctx = self._ctx()
        ph, pw = self._shape(c[0]) if n else (h, w)
        out = np.empty((n, ph, pw), dtype=np.uint64)
        counts = np.zeros(max(n, 1), dtype=np.uintp)
        failed = ctypes.c_size_t(0)
        if seeds is None:
            flat, offs = None, None
        else:
            if len(seeds) != n:
                raise ValueError("one seed list per slice")
            lists = [np.asarray(s, dtype=np.uint64).reshape(-1, 2) for s in seeds]
            offs = np.zeros(n + 1, dtype=np.uintp)
            offs[1:] = np.cumsum([len(l) for l in lists])
            flat = np.ascontiguousarray(np.concatenate(lists, axis=0) if lists else np.zeros((0, 2), dtype=np.uint64))
            if flat.shape[0] == 0:
                flat = np.zeros((1, 2), dtype=np.uint64)
        ctx.check(_ffi.lib().ws_segment_batch(ctx.handle, c.ctypes.data, n, h, w, w, h * w, flat.ctypes.data if flat is not None else None,
                                              offs.ctypes.data_as(_ffi.szp) if offs is not None else None, ctypes.byref(self._opt),
                                              out.ctypes.data, counts.ctypes.data_as(_ffi.szp), ctypes.byref(failed)))
        return out if seeds is not None else (out, counts[:n].astype(np.int64))


class MergingWatershed(_Transform):
    """lib.rs:1297-1562"""
    _merging = True

    def transform(self, input, _seeds=None):              # lib.rs:1524-1536: a stub in the reference
        a, _ = _as_image(input)
        out = np.empty(a.shape, dtype=np.uint64)
        rc = _ffi.lib().ws_merge_transform_stub(a.shape[0], a.shape[1], out.ctypes.data)
        if rc != _ffi.WS_OK:
            raise WatershedError(rc)
        return out

    def transform_final(self, input, seeds):
        """Not in the reference: the merged label plane after the last level."""
        return self._run_with_hook(input, seeds, None, True)[1]
