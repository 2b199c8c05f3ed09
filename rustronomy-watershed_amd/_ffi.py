"""ctypes binding of include/ws_hip.h (the C-ABI of the HIP engine).

There is no CPU fallback: if the shared library has not been built this module raises,
and every compute call needs a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WS_HIP_LIB: another build of the same ABI (tools/ point it at the -DWS_TUNING build for their A/B experiments)
LIB_PATH = os.environ.get("WS_HIP_LIB") or os.path.join(_HERE, "libws_hip.so")

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)
szp = ctypes.POINTER(ctypes.c_size_t)
sz = ctypes.c_size_t
vp = ctypes.c_void_p

WS_OK = 0
WS_ERR_BAD_ARG = -1
WS_ERR_MAX_TOO_HIGH = -2
WS_ERR_MAX_TOO_LOW = -3
WS_ERR_SEED_OOB = -4
WS_ERR_HIP = -5
WS_ERR_OOM = -6
WS_ERR_NO_DEVICE = -7
WS_ERR_CAPACITY = -8
WS_ERR_RING_OVERFLOW = -9
WS_ERR_TOO_LARGE = -10
WS_ERR_UNSUPPORTED = -11
WS_ERR_RCCL = -12
WS_RCCL_ID_BYTES = 128

WS_ENGINE_AUTO, WS_ENGINE_FUSED, WS_ENGINE_SWEEP = 0, 1, 2
WS_DTYPES = {"float32": 0, "float64": 1, "int32": 2, "uint16": 3, "int16": 4, "uint8": 5}


WS_ABI_VERSION = 3


class Options(ctypes.Structure):
    """ws_options of include/ws_hip.h (ABI version 2: 8 bytes)."""
    _fields_ = [("max_water_level", ctypes.c_uint8), ("edge_correction", ctypes.c_uint8),
                ("engine", ctypes.c_uint8), ("tie_rule", ctypes.c_uint8),
                ("seed_shift", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 3)]

    def __init__(self, max_water_level=254, edge_correction=0, engine=0, tie_rule=0, seed_shift=0):
        super().__init__(max_water_level, edge_correction, engine, tie_rule, seed_shift)


class Stats(ctypes.Structure):
    _fields_ = [("relax_passes", ctypes.c_uint32), ("resolve_passes", ctypes.c_uint32),
                ("sweep_steps", ctypes.c_uint32), ("merge_levels", ctypes.c_uint32),
                ("tiles_run_relax", ctypes.c_uint64), ("tiles_run_resolve", ctypes.c_uint64),
                ("ms_relax", ctypes.c_float), ("ms_resolve", ctypes.c_float), ("ms_sweep", ctypes.c_float),
                ("ms_other", ctypes.c_float), ("ms_total", ctypes.c_float),
                ("launches_relax", ctypes.c_uint32), ("launches_resolve", ctypes.c_uint32),
                ("launches_sweep", ctypes.c_uint32), ("relax_tile_iterations", ctypes.c_uint32),
                ("graph_launches", ctypes.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Lake(ctypes.Structure):
    _fields_ = [("colour", ctypes.c_uint64), ("area", ctypes.c_uint64)]


class TileBlock(ctypes.Structure):
    """ws_tile_block: one rank's row block of a tiled field, device resident."""
    _fields_ = [("d_img", vp), ("d_seeds_rc", vp), ("d_colours", vp), ("n_seeds", sz),
                ("first_colour", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("d_labels", vp)]


class TileBlock2D(ctypes.Structure):
    """ws_tile_block2d: one rank's tile of a field cut in both directions, device resident."""
    _fields_ = [("d_img", vp), ("img_stride", sz), ("d_seeds_rc", vp), ("d_colours", vp), ("n_seeds", sz), ("d_labels", vp)]


class BatchPart(ctypes.Structure):
    """ws_batch_part: one rank's slices of a batch, device resident (seed_offsets on the host)."""
    _fields_ = [("d_cube", vp), ("d_seeds_rc", vp), ("seed_offsets", szp), ("n_slices", sz), ("d_labels", vp)]


LEVEL_CB = ctypes.CFUNCTYPE(None, vp, ctypes.c_uint8, ctypes.c_uint8, u8p, u64p, sz, sz)

# every symbol include/ws_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ws_abi_version": (ctypes.c_int, []),
    "ws_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "ws_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
    "ws_ctx_create_on_stream": (ctypes.c_int, [ctypes.c_int, vp, ctypes.POINTER(vp)]),
    "ws_ctx_destroy": (None, [vp]),
    "ws_last_error": (ctypes.c_char_p, [vp]),
    "ws_ctx_set_profiling": (ctypes.c_int, [vp, ctypes.c_int]),
    "ws_ctx_get_stats": (ctypes.c_int, [vp, ctypes.POINTER(Stats)]),
    "ws_ctx_synchronize": (ctypes.c_int, [vp]),
    "ws_ctx_set_batch_pixel_limit": (ctypes.c_int, [vp, sz]),
    "ws_ctx_set_seam_repair_min_pixels": (ctypes.c_int, [vp, sz]),
    "ws_ctx_set_live_list_min_colours": (ctypes.c_int, [vp, sz]),
    "ws_ctx_set_persistent_pass": (ctypes.c_int, [vp, ctypes.c_int]),
    "ws_ctx_set_host_threads": (ctypes.c_int, [vp, ctypes.c_int]),
    "ws_options_default": (ctypes.c_int, [ctypes.POINTER(Options)]),
    "ws_options_validate": (ctypes.c_int, [ctypes.POINTER(Options)]),
    "ws_find_local_minima": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, szp]),
    "ws_segment": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp]),
    "ws_segment_u32": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp]),
    "ws_segment_minima": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.POINTER(Options), vp, vp, sz, szp]),
    "ws_segment_minima_u32": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.POINTER(Options), vp, vp, sz, szp]),
    "ws_segment_minima_device": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.POINTER(Options), vp, vp, sz, szp]),
    "ws_segment_with_hook": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp, vp, vp]),
    "ws_merge_with_hook": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp, vp, vp]),
    "ws_transform_to_list": (ctypes.c_int, [vp, ctypes.c_int, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp, sz,
                                            szp, vp, vp]),
    "ws_transform_to_list_device": (ctypes.c_int, [vp, ctypes.c_int, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp, sz,
                                                   szp, vp, vp]),
    "ws_lists_from_arrival_device": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, sz, sz, sz, ctypes.POINTER(Options), vp, sz, szp, vp, vp]),
    "ws_merge_transform_stub": (ctypes.c_int, [sz, sz, vp]),
    "ws_segment_batch": (ctypes.c_int, [vp, vp, sz, sz, sz, sz, sz, vp, szp, ctypes.POINTER(Options), vp, szp, szp]),
    "ws_segment_batch_host": (ctypes.c_int, [vp, vp, sz, sz, sz, sz, sz, vp, szp, ctypes.POINTER(Options), vp, szp, szp]),
    "ws_find_local_minima_device": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, szp]),
    "ws_segment_device": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp]),
    "ws_segment_device_begin": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp]),
    "ws_segment_device_end": (ctypes.c_int, [vp]),
    "ws_segment_batch_device": (ctypes.c_int, [vp, vp, sz, sz, sz, sz, sz, vp, ctypes.POINTER(ctypes.c_size_t),
                                               ctypes.POINTER(Options), vp, ctypes.POINTER(ctypes.c_size_t)]),
    "ws_merge_device": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp]),
    "ws_merge_device_begin": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp]),
    "ws_merge_device_end": (ctypes.c_int, [vp]),
    "ws_last_arrival_device": (ctypes.c_int, [vp, ctypes.POINTER(vp), szp, szp]),
    "ws_copy_last_arrival_device": (ctypes.c_int, [vp, vp, sz]),
    "ws_level_snapshot_device": (ctypes.c_int, [vp, vp, ctypes.c_uint8, vp]),
    "ws_pre_processor": (ctypes.c_int, [vp, vp, ctypes.c_int, sz, ctypes.c_uint8, vp]),
    "ws_pre_processor_device": (ctypes.c_int, [vp, vp, ctypes.c_int, sz, ctypes.c_uint8, vp]),
    "ws_block_init": (ctypes.c_int, [vp, sz, sz, vp, vp, sz, vp, vp]),
    "ws_block_relax": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.c_uint8, vp, ctypes.POINTER(ctypes.c_int)]),
    "ws_block_resolve": (ctypes.c_int, [vp, vp, vp, sz, sz, ctypes.POINTER(ctypes.c_int)]),
    "ws_block_resolve_ring": (ctypes.c_int, [vp, vp, vp, sz, sz]),
    "ws_block_begin": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.c_uint8, vp, sz, ctypes.c_uint32, vp]),
    "ws_block_relax_halo": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.c_uint8, ctypes.c_int, ctypes.c_int, vp]),
    "ws_block_resolve_local": (ctypes.c_int, [vp, vp, vp, sz, sz, ctypes.c_int, ctypes.c_int]),
    "ws_block_export_boundary": (ctypes.c_int, [vp, vp, sz, sz, ctypes.c_int, ctypes.c_int, sz, vp]),
    "ws_block_import_boundary": (ctypes.c_int, [vp, vp, sz, sz, vp, sz, sz, ctypes.c_int, ctypes.c_int]),
    "ws_block_merge_local": (ctypes.c_int, [vp, vp, sz, sz, sz, sz, sz, vp]),
    "ws_block_merge_export": (ctypes.c_int, [vp, vp, sz, sz, vp, vp]),
    "ws_block_merge_import": (ctypes.c_int, [vp, vp, sz, vp]),
    "ws_block_merge_relabel": (ctypes.c_int, [vp, vp, sz, vp, sz, vp]),
    "ws_random_field_device": (ctypes.c_int, [vp, vp, sz, sz, sz, ctypes.c_uint64]),
    "ws_group_create_local": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp)]),
    "ws_group_rccl_unique_id": (ctypes.c_int, [vp]),
    "ws_group_create_rccl": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(vp)]),
    "ws_group_destroy": (None, [vp]),
    "ws_group_info": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ws_group_last_error": (ctypes.c_char_p, [vp]),
    "ws_group_selftest": (ctypes.c_int, [vp]),
    "ws_tile_rows": (ctypes.c_int, [sz, ctypes.c_int, ctypes.c_int, szp, szp, szp, szp]),
    "ws_segment_tiled": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), ctypes.c_int, vp, u32p]),
    "ws_segment_tiled_device": (ctypes.c_int, [vp, sz, sz, sz, ctypes.POINTER(TileBlock), ctypes.POINTER(Options), ctypes.c_int, u32p]),
    "ws_transform_to_list_tiled": (ctypes.c_int, [vp, ctypes.c_int, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), vp, sz, szp, vp, vp, u32p]),
    "ws_transform_to_list_tiled_device": (ctypes.c_int, [vp, sz, sz, sz, vp, ctypes.POINTER(Options), ctypes.c_int, vp, sz, szp, vp, vp, u32p]),
    "ws_tile_grid": (ctypes.c_int, [sz, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, szp, szp]),
    "ws_segment_tiled2d_device": (ctypes.c_int, [vp, sz, sz, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(TileBlock2D), ctypes.POINTER(Options), ctypes.c_int, u32p]),
    "ws_transform_to_list_tiled2d_device": (ctypes.c_int, [vp, sz, sz, ctypes.c_int, ctypes.c_int, sz, vp, ctypes.POINTER(Options), ctypes.c_int, vp, sz, szp, vp, vp, u32p]),
    "ws_segment_tiled2d": (ctypes.c_int, [vp, vp, sz, sz, sz, vp, sz, ctypes.POINTER(Options), ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, u32p]),
    "ws_segment_batch_group": (ctypes.c_int, [vp, sz, sz, ctypes.POINTER(BatchPart), ctypes.POINTER(Options), szp, szp]),
}

_lib = None


def lib():
    """Loads libws_hip.so (RTLD_LOCAL).  Raises if it has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = the library does not export the ABI
            fn.restype = res
            fn.argtypes = args
        if L.ws_abi_version() != WS_ABI_VERSION:
            raise ImportError(f"{LIB_PATH} speaks ABI version {L.ws_abi_version()}, this binding {WS_ABI_VERSION}: rebuild it")
        _lib = L
    return _lib
