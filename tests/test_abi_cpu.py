"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/ws_hip.h declares, the host-side mirror reproduces the reference's builder behaviour,
and the product fails loudly (no fallback) when there is no HIP device."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_unit_vectors.json")))


@pytest.fixture(scope="module")
def pkg():
    ge.build_hip()
    return ge.load_package()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ws_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", text))
    names -= {"ws_level_cb"}
    return names


def test_library_exports_every_declared_symbol(pkg):
    declared = _declared_functions()
    assert len(declared) >= 20
    assert declared == set(pkg._ffi.SIGNATURES), declared ^ set(pkg._ffi.SIGNATURES)
    raw = ctypes.CDLL(pkg._ffi.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert pkg._ffi.lib().ws_abi_version() == pkg._ffi.WS_ABI_VERSION == 3


def test_options_default_and_validation(pkg):
    L = pkg._ffi.lib()
    opt = pkg._ffi.Options()
    assert L.ws_options_default(ctypes.byref(opt)) == 0
    b = GOLD["builder"]
    assert opt.max_water_level == b["default_max_water_level"] and opt.edge_correction == 0
    for v in b["valid"]:
        opt.max_water_level = v
        assert L.ws_options_validate(ctypes.byref(opt)) == 0
    for v in b["max_to_high"]:
        opt.max_water_level = v
        assert L.ws_options_validate(ctypes.byref(opt)) == pkg._ffi.WS_ERR_MAX_TOO_HIGH
    for v in b["max_to_low"]:
        opt.max_water_level = v
        assert L.ws_options_validate(ctypes.byref(opt)) == pkg._ffi.WS_ERR_MAX_TOO_LOW
    assert b"254" in L.ws_strerror(pkg._ffi.WS_ERR_MAX_TOO_HIGH)
    # ABI version 2: 8 bytes, seed_shift (default 0 = lib.rs:1675-1677, seeds not moved into the padded plane) + reserved
    assert ctypes.sizeof(pkg._ffi.Options) == 8 and pkg._ffi.Options.seed_shift.offset == 4
    L.ws_options_default(ctypes.byref(opt))
    assert opt.seed_shift == 0 and list(opt.reserved) == [0, 0, 0]
    opt.seed_shift = 2
    assert L.ws_options_validate(ctypes.byref(opt)) == pkg._ffi.WS_ERR_BAD_ARG
    opt.seed_shift = 1
    assert L.ws_options_validate(ctypes.byref(opt)) == 0
    opt.reserved[1] = 7
    assert L.ws_options_validate(ctypes.byref(opt)) == pkg._ffi.WS_ERR_BAD_ARG


def test_shipped_library_ignores_its_environment(pkg):
    # debug / A-B knobs (WS_DEBUG_MAXIT, WS_NO_GRAPH ...) exist only in a -DWS_TUNING build: the product .so must not
    # even contain their names, let alone read them
    blob = open(pkg._ffi.LIB_PATH, "rb").read()
    for knob in (b"WS_DEBUG_MAXIT", b"WS_NO_GRAPH", b"WS_NO_SPECULATION", b"WS_NO_SEED_TABLES", b"WS_NO_BATCH_STACK",
                 b"WS_BATCH_MAX_PX", b"WS_RELAX_P0_ROUNDS", b"WS_RELAX_CHUNK_FROM", b"WS_PAINT_STEPS", b"WS_RELAX_LATE_CAP",
                 b"WS_RELAX_EARLY_CAP", b"WS_RELAX_NO_SPLIT", b"WS_RELAX_NO_SEAM", b"WS_RELAX_LIST_FROM", b"WS_RELAX_SAME_GRID_FROM", b"WS_RELAX_LITE_FROM",
                 b"WS_DEBUG_LIST", b"WS_RELAX_NO_APPEND", b"WS_RELAX_SCAN_FROM", b"getenv"):
        assert knob not in blob, knob
    csrc = os.path.join(ROOT, "rustronomy-watershed_amd", "csrc")
    sources = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")) and f != "ws_common.hpp")
    assert "ws_segment.hip" in sources and "ws_relax.hip" in sources
    for src in sources:
        text = open(os.path.join(csrc, src)).read()
        assert "getenv" not in text.replace("tuning_env", ""), src


def test_builder_mirrors_reference(pkg):
    # lib.rs:936-946 defaults; lib.rs:1026-1030 validation, for both build_* methods
    b = pkg.TransformBuilder.default()
    assert b.max_water_level == 254 and b.edge_correction is False and b.wlvl_hook is None
    for build in ("build_segmenting", "build_merging"):
        with pytest.raises(pkg.MaxToHigh):
            getattr(pkg.TransformBuilder.new().set_max_water_lvl(255), build)()
        with pytest.raises(pkg.MaxToLow):
            getattr(pkg.TransformBuilder.new().set_max_water_lvl(0), build)()
        ws = getattr(pkg.TransformBuilder.new().set_max_water_lvl(127).enable_edge_correction(), build)()
        assert ws.max_water_level == 127 and ws.edge_correction is True
    assert issubclass(pkg.MaxToHigh, pkg.BuildErr) and issubclass(pkg.MaxToLow, pkg.BuildErr)
    assert (pkg.UNCOLOURED, pkg.NORMAL_MAX, pkg.ALWAYS_FILL, pkg.NEVER_FILL) == (0, 254, 0, 255)


def test_merge_transform_stub_matches_reference_stub(pkg):
    # lib.rs:1524-1536: zeros with the interior set to 123, seeds ignored (pure host code)
    ws = pkg.TransformBuilder.default().build_merging()
    out = ws.transform(np.zeros((5, 7), np.uint8), [(1, 1)])
    assert out.dtype == np.uint64 and (out[1:-1, 1:-1] == 123).all()
    assert out[0].sum() == 0 and out[-1].sum() == 0 and out[:, 0].sum() == 0 and out[:, -1].sum() == 0


def test_no_device_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert pkg._ffi.lib().ws_ctx_create(0, ctypes.byref(h)) == pkg._ffi.WS_ERR_NO_DEVICE
    with pytest.raises(pkg.WatershedError):
        pkg.TransformBuilder.default().build_segmenting().transform(np.zeros((8, 8), np.uint8), [(3, 3)])


def test_missing_library_raises(pkg, monkeypatch):
    ffi = pkg._ffi
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", ffi.LIB_PATH + ".absent")
    with pytest.raises(ImportError):
        ffi.lib()


def test_product_does_not_reference_the_oracle():
    # the shipped package must never import, load or call anything under oracle/
    pkg_dir = os.path.join(ROOT, "rustronomy-watershed_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_stats_struct_layout_is_stable():
    # ws_stats is filled through a caller-provided pointer: its size and the offsets of the old fields are ABI.
    # graph_launches was added in what used to be tail padding (68 -> 72 bytes were always reserved by alignment).
    import ctypes
    import importlib
    ge.load_package()
    ffi = importlib.import_module("rustronomy_watershed_amd._ffi")
    assert ctypes.sizeof(ffi.Stats) == 72
    assert ffi.Stats.relax_tile_iterations.offset == 64 and ffi.Stats.graph_launches.offset == 68
    assert ffi.Stats.tiles_run_relax.offset == 16 and ffi.Stats.ms_relax.offset == 32


def test_integration_extern_block_lists_every_header_function():
    # INTEGRATION.md shows the binding a maintainer of the reference would add: it must not fall behind the header
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "ws_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"^(?:int|void|const char \*)\s*(ws_[a-z0-9_]+)\(", header, flags=re.M)))
    assert len(names) >= 25
    missing = [n for n in names if f"fn {n}(" not in doc]
    assert not missing, missing
    # the Rust shim's extern block (rust/src/hip_ffi.rs) declares every one of them, with the ABI version of the header
    ffi = open(os.path.join(root, "rust", "src", "hip_ffi.rs")).read()
    missing = [n for n in names if f"pub fn {n}(" not in ffi]
    assert not missing, missing
    version = re.search(r"#define WS_ABI_VERSION (\d+)", header).group(1)
    assert f"WS_ABI_VERSION: c_int = {version};" in ffi
    shim = open(os.path.join(root, "rust", "src", "watershed_hip.rs")).read()
    for method in ("fn transform(", "fn transform_with_hook(", "fn transform_to_list(", "fn transform_history(",
                   "fn find_local_minima(", "fn pre_processor_with_max<", "fn build_segmenting(", "fn build_merging("):
        assert method in shim, method


def test_bench_and_entry_point_compile_and_parse_their_arguments():
    # bench.py only runs on a GPU box: here it must at least be importable Python with the contract's flags
    import importlib.util
    import py_compile
    for name in ("bench.py", "__graft_entry__.py"):
        py_compile.compile(os.path.join(ROOT, name), doraise=True)
    spec = importlib.util.spec_from_file_location("ws_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, sys.argv = sys.argv, ["bench.py", "--gpus", "2", "--steps", "7", "--warmup", "3", "--config", "c5", "--tiles", "2x1"]
    try:
        a = mod.parse_args()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup, a.config, a.tiles) == (2, 7, 3, "c5", "2x1")
    sys.argv = ["bench.py"]
    try:
        d = mod.parse_args()
    finally:
        sys.argv = argv
    assert d.gpus == 1 and d.config == "headline" and d.steps >= 1 and d.warmup >= 0
