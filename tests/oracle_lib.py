"""ctypes binding of the CPU oracle (oracle/ws_oracle.h) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product (rustronomy-watershed_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "_build", "libws_oracle.so")

TIE_FIRST, TIE_RANDOM = 0, 1
MAP_FAITHFUL, MAP_CANONICAL = 0, 1
ERR_SEED_OOB = -1

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_sz = ctypes.c_size_t

LEVEL_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint8, ctypes.c_uint8, _u8p, _u64p, _sz, _sz)


class Stats(ctypes.Structure):
    _fields_ = [("scans", ctypes.c_uint64), ("max_rings", ctypes.c_uint64), ("flooded", ctypes.c_uint64),
                ("conflicts", ctypes.c_uint64), ("merge_pairs", ctypes.c_uint64)]


def build(force=False):
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("ws_oracle.c", "ws_oracle_par.c", "ws_oracle.h", "Makefile")]
    stale = not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _ORACLE_DIR], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.ws_or_mix64.restype = ctypes.c_uint64
        L.ws_or_mix64.argtypes = [ctypes.c_uint64]
        L.ws_or_random_field.argtypes = [_u8p, _sz, _sz, ctypes.c_uint64]
        L.ws_or_find_flooded_px.restype = _sz
        L.ws_or_find_flooded_px.argtypes = [_u8p, _u64p, _sz, _sz, ctypes.c_uint8, ctypes.c_int, _u64p, _u64p, _u64p, _u8p]
        L.ws_or_segment.restype = ctypes.c_int
        L.ws_or_segment.argtypes = [_u8p, _sz, _sz, _u64p, _sz, ctypes.c_uint8, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_uint64, _u64p, _i32p, _u32p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.POINTER(Stats)]
        L.ws_or_find_merge.restype = _sz
        L.ws_or_find_merge.argtypes = [_u64p, _sz, _sz, _u64p, _sz]
        L.ws_or_make_colour_map.restype = ctypes.c_int
        L.ws_or_make_colour_map.argtypes = [_u64p, _sz, _u64p, _sz, ctypes.c_int]
        L.ws_or_recolour.argtypes = [_u64p, _sz, _u64p]
        L.ws_or_find_lake_sizes.argtypes = [_u64p, _sz, _u64p]
        L.ws_or_merge.restype = ctypes.c_int
        L.ws_or_merge.argtypes = [_u8p, _sz, _sz, _u64p, _sz, ctypes.c_uint8, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_uint64, ctypes.c_int, _u64p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.POINTER(Stats)]
        L.ws_or_merge_transform_stub.argtypes = [_sz, _sz, _u64p]
        L.ws_or_find_local_minima.restype = _sz
        L.ws_or_find_local_minima.argtypes = [_u8p, _sz, _sz, _u64p, _sz]
        L.ws_or_check_reachable.restype = ctypes.c_int
        L.ws_or_check_reachable.argtypes = [_u8p, _sz, _sz, _u64p, _sz, ctypes.c_uint8, ctypes.c_int, _u64p,
                                            ctypes.POINTER(_sz)]
        L.ws_or_canonicalise.restype = _sz
        L.ws_or_canonicalise.argtypes = [_u64p, _sz, _sz, _u64p, _sz]
        L.ws_or_segment_arrival.restype = ctypes.c_int
        L.ws_or_segment_arrival.argtypes = [_u8p, _sz, _sz, _u64p, _sz, ctypes.c_uint8, ctypes.c_int, _u64p, _u64p]
        L.ws_or_merge_arrival.restype = ctypes.c_int
        L.ws_or_merge_arrival.argtypes = [_u8p, _sz, _sz, _u64p, _sz, ctypes.c_uint8, ctypes.c_int, _u64p,
                                          ctypes.c_void_p, ctypes.c_void_p]
        L.ws_or_pre_processor.restype = ctypes.c_int
        L.ws_or_pre_processor.argtypes = [ctypes.c_void_p, ctypes.c_int, _sz, ctypes.c_uint8, _u8p]
        L.ws_or_max_threads.restype = ctypes.c_int
        L.ws_or_segment_par.restype = ctypes.c_int
        L.ws_or_segment_par.argtypes = [_u8p, _sz, _sz, _u64p, _sz, ctypes.c_uint8, ctypes.c_int, _u64p,
                                        ctypes.POINTER(Stats)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def _seeds(seeds):
    s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1, 2))
    return s, s.shape[0]


def _pshape(img, edge):
    return (img.shape[0] + 2, img.shape[1] + 2) if edge else img.shape


class SeedOutOfBounds(IndexError):
    """the reference panics on an out-of-bounds seed (lib.rs:1366 / 1676)"""


def random_field(h, w, seed):
    img = np.empty((h, w), dtype=np.uint8)
    lib().ws_or_random_field(_p(img, _u8p), h, w, seed)
    return img


def random_field_numpy(h, w, seed):
    """numpy twin of ws_or_random_field (same stream), used to cross-check the generator."""
    idx = np.arange(h * w, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
    with np.errstate(over="ignore"):
        z = idx + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(254)).astype(np.uint8).reshape(h, w)


def find_flooded_px(img, cols, lvl, tie=TIE_FIRST, rng_seed=0):
    img = _img(img)
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    h, w = img.shape
    n = max(h * w, 1)
    rc = np.empty((n, 2), dtype=np.uint64)
    col = np.empty(n, dtype=np.uint64)
    cf = np.empty(n, dtype=np.uint8)
    rng = ctypes.c_uint64(rng_seed)
    k = lib().ws_or_find_flooded_px(_p(img, _u8p), _p(cols, _u64p), h, w, lvl, tie, ctypes.byref(rng),
                                    _p(rc, _u64p), _p(col, _u64p), _p(cf, _u8p))
    return [((int(rc[i, 0]), int(rc[i, 1])), int(col[i])) for i in range(k)]


def segment(img, seeds, max_level=254, edge=False, tie=TIE_FIRST, rng_seed=0, hook=None, want_arrival=False,
            want_stats=False):
    img = _img(img)
    s, ns = _seeds(seeds)
    h, w = img.shape
    ph, pw = _pshape(img, edge)
    out = np.zeros((ph, pw), dtype=np.uint64)
    al = np.zeros((ph, pw), dtype=np.int32) if want_arrival else None
    ar = np.zeros((ph, pw), dtype=np.uint32) if want_arrival else None
    st = Stats()
    cb = None
    if hook is not None:
        def _cb(_user, lvl, mx, pimg, plab, hh, ww):
            limg = np.ctypeslib.as_array(pimg, shape=(hh, ww))
            llab = np.ctypeslib.as_array(plab, shape=(hh, ww))
            hook(lvl, mx, limg, llab)
        cb = LEVEL_CB(_cb)
    rc = lib().ws_or_segment(_p(img, _u8p), h, w, _p(s, _u64p), ns, max_level, int(edge), tie, rng_seed,
                             _p(out, _u64p), _p(al, _i32p) if want_arrival else None,
                             _p(ar, _u32p) if want_arrival else None,
                             ctypes.cast(cb, ctypes.c_void_p) if cb else None, None, ctypes.byref(st))
    if rc == ERR_SEED_OOB:
        raise SeedOutOfBounds("seed outside the label plane")
    assert rc == 0, rc
    res = [out]
    if want_arrival:
        res += [al, ar]
    if want_stats:
        res.append(st)
    return res[0] if len(res) == 1 else tuple(res)


def segment_par(img, seeds, max_level=254, threads=0):
    img = _img(img)
    s, ns = _seeds(seeds)
    h, w = img.shape
    out = np.zeros((h, w), dtype=np.uint64)
    st = Stats()
    rc = lib().ws_or_segment_par(_p(img, _u8p), h, w, _p(s, _u64p), ns, max_level, threads, _p(out, _u64p),
                                 ctypes.byref(st))
    if rc == ERR_SEED_OOB:
        raise SeedOutOfBounds("seed outside the label plane")
    assert rc == 0, rc
    return out, st


def max_threads():
    return lib().ws_or_max_threads()


def segment_arrival(img, seeds, max_level=254, edge=False, want_keys=False):
    img = _img(img)
    s, ns = _seeds(seeds)
    h, w = img.shape
    ph, pw = _pshape(img, edge)
    out = np.zeros((ph, pw), dtype=np.uint64)
    keys = np.zeros((ph, pw), dtype=np.uint64) if want_keys else None
    rc = lib().ws_or_segment_arrival(_p(img, _u8p), h, w, _p(s, _u64p), ns, max_level, int(edge), _p(out, _u64p),
                                     _p(keys, _u64p) if want_keys else None)
    if rc == ERR_SEED_OOB:
        raise SeedOutOfBounds("seed outside the label plane")
    assert rc == 0, rc
    return (out, keys) if want_keys else out


def find_merge(labels):
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    h, w = labels.shape
    cap = 4 * h * w + 4
    out = np.empty((cap, 2), dtype=np.uint64)
    n = lib().ws_or_find_merge(_p(labels, _u64p), h, w, _p(out, _u64p), cap)
    return [(int(out[i, 0]), int(out[i, 1])) for i in range(n)]


def make_colour_map(base_map, pairs, mode=MAP_FAITHFUL):
    m = np.ascontiguousarray(base_map, dtype=np.uint64).copy()
    p = np.ascontiguousarray(np.asarray(pairs, dtype=np.uint64).reshape(-1, 2))
    rc = lib().ws_or_make_colour_map(_p(m, _u64p), m.size, _p(p, _u64p), p.shape[0], mode)
    assert rc == 0
    return m


def recolour(labels, cmap):
    lab = np.ascontiguousarray(labels, dtype=np.uint64).copy()
    cm = np.ascontiguousarray(cmap, dtype=np.uint64)
    lib().ws_or_recolour(_p(lab, _u64p), lab.size, _p(cm, _u64p))
    return lab


def find_lake_sizes(labels):
    lab = np.ascontiguousarray(labels, dtype=np.uint64)
    hist = np.empty(lab.size + 1, dtype=np.uint64)
    lib().ws_or_find_lake_sizes(_p(lab, _u64p), lab.size, _p(hist, _u64p))
    return hist


def merge(img, seeds, max_level=254, edge=False, tie=TIE_FIRST, rng_seed=0, mode=MAP_CANONICAL, hook=None,
          want_stats=False):
    img = _img(img)
    s, ns = _seeds(seeds)
    h, w = img.shape
    ph, pw = _pshape(img, edge)
    out = np.zeros((ph, pw), dtype=np.uint64)
    st = Stats()
    cb = None
    if hook is not None:
        def _cb(_user, lvl, mx, pimg, plab, hh, ww):
            hook(lvl, mx, np.ctypeslib.as_array(pimg, shape=(hh, ww)), np.ctypeslib.as_array(plab, shape=(hh, ww)))
        cb = LEVEL_CB(_cb)
    rc = lib().ws_or_merge(_p(img, _u8p), h, w, _p(s, _u64p), ns, max_level, int(edge), tie, rng_seed, mode,
                           _p(out, _u64p), ctypes.cast(cb, ctypes.c_void_p) if cb else None, None, ctypes.byref(st))
    if rc == ERR_SEED_OOB:
        raise SeedOutOfBounds("seed outside the label plane")
    assert rc == 0, rc
    return (out, st) if want_stats else out


def merge_arrival(img, seeds, max_level=254, edge=False, hook=None):
    img = _img(img)
    s, ns = _seeds(seeds)
    h, w = img.shape
    ph, pw = _pshape(img, edge)
    out = np.zeros((ph, pw), dtype=np.uint64)
    cb = None
    if hook is not None:
        def _cb(_user, lvl, mx, pimg, plab, hh, ww):
            hook(lvl, mx, np.ctypeslib.as_array(pimg, shape=(hh, ww)), np.ctypeslib.as_array(plab, shape=(hh, ww)))
        cb = LEVEL_CB(_cb)
    rc = lib().ws_or_merge_arrival(_p(img, _u8p), h, w, _p(s, _u64p), ns, max_level, int(edge), _p(out, _u64p),
                                   ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
    if rc == ERR_SEED_OOB:
        raise SeedOutOfBounds("seed outside the label plane")
    assert rc == 0, rc
    return out


def merge_transform_stub(h, w):
    out = np.empty((h, w), dtype=np.uint64)
    lib().ws_or_merge_transform_stub(h, w, _p(out, _u64p))
    return out


def find_local_minima(img):
    img = _img(img)
    h, w = img.shape
    cap = max(h * w, 1)
    out = np.empty((cap, 2), dtype=np.uint64)
    n = lib().ws_or_find_local_minima(_p(img, _u8p), h, w, _p(out, _u64p), cap)
    return out[:n].copy()


def check_reachable(img, seeds, cand, max_level=254, edge=False):
    img = _img(img)
    s, ns = _seeds(seeds)
    h, w = img.shape
    cand = np.ascontiguousarray(cand, dtype=np.uint64)
    assert cand.shape == _pshape(img, edge)
    bad = _sz(0)
    v = lib().ws_or_check_reachable(_p(img, _u8p), h, w, _p(s, _u64p), ns, max_level, int(edge), _p(cand, _u64p),
                                    ctypes.byref(bad))
    return v, int(bad.value)


def canonicalise(labels, seeds):
    lab = np.ascontiguousarray(labels, dtype=np.uint64).copy()
    s, ns = _seeds(seeds)
    h, w = lab.shape
    n = lib().ws_or_canonicalise(_p(lab, _u64p), h, w, _p(s, _u64p), ns)
    return lab, int(n)


DTYPES = {"float32": 0, "float64": 1, "int32": 2, "uint16": 3, "int16": 4, "uint8": 5}


def pre_processor(arr, max_value=254):
    a = np.ascontiguousarray(arr)
    out = np.empty(a.shape, dtype=np.uint8)
    rc = lib().ws_or_pre_processor(a.ctypes.data, DTYPES[a.dtype.name], a.size, max_value, _p(out, _u8p))
    if rc != 0:
        raise AssertionError("MAX must be in 1..=254 (lib.rs:1143-1144)")
    return out


def pre_processor_numpy(arr, max_value=254):
    """Independent numpy restatement of lib.rs:1134-1173 (IEEE f64 arithmetic, same operation order)."""
    x = np.asarray(arr).astype(np.float64)
    fin = np.isfinite(x)
    mn = min(0.0, float(x[fin].min())) if fin.any() else 0.0
    mx = max(0.0, float(x[fin].max())) if fin.any() else 0.0
    out = np.full(x.shape, 255, dtype=np.uint8)
    normal = fin & (np.abs(x) >= np.finfo(np.float64).tiny)
    with np.errstate(all="ignore"):
        q = ((x - mn) / (mx - mn)) * float(max_value)
    out[normal] = np.trunc(q[normal]).astype(np.uint8)
    out[np.isposinf(x)] = 0
    return out
