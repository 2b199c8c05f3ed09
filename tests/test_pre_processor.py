"""pre_processor / pre_processor_with_max (lib.rs:1081-1173): the first "next" row of SURVEY 8f.
The reference has no test for it; the restatement is pinned by its own code reading (quirks listed in
include/ws_hip.h) with two independent implementations (C oracle, numpy) and the GPU against both."""
import numpy as np
import pytest

import __graft_entry__ as ge
import oracle_lib as ol


def _cases():
    rng = np.random.default_rng(3)
    out = []
    a = rng.random((40, 50))                                     # README-style Uniform(0,1) f64 field
    out.append(("uniform01_f64", a))
    b = rng.poisson(0.85, (30, 70)).astype(np.float64)           # tests/integration.rs:189 Poisson(0.85): many exact zeros
    out.append(("poisson_f64", b))
    c = rng.normal(0, 5, (33, 17)).astype(np.float32)
    c[3, 4] = np.nan; c[5, 6] = np.inf; c[7, 8] = -np.inf; c[9, 9] = 0.0; c[1, 1] = -0.0; c[2, 2] = 1e-42   # f32 subnormal
    out.append(("specials_f32", c))
    d = rng.normal(100, 30, (20, 20))
    d[0, 0] = 5e-324; d[1, 1] = np.nan; d[2, 2] = np.inf
    out.append(("all_positive_f64", d))                          # min fold stays at its seed 0
    out.append(("all_negative_f64", -np.abs(rng.normal(3, 1, (9, 11)))))   # max fold stays at 0
    out.append(("int32", rng.integers(-1000, 1000, (25, 25), dtype=np.int32)))
    out.append(("uint16", rng.integers(0, 65535, (16, 64), dtype=np.uint16)))
    out.append(("int16", rng.integers(-32768, 32767, (8, 8), dtype=np.int16)))
    out.append(("uint8", rng.integers(0, 255, (31, 3), dtype=np.uint8)))
    out.append(("cube_f32", rng.random((4, 10, 12)).astype(np.float32)))   # any dimension (lib.rs:1081: ArrayView<T, D>)
    out.append(("all_nan", np.full((5, 5), np.nan)))
    out.append(("all_zero", np.zeros((4, 4))))
    out.append(("empty", np.zeros((0, 7), dtype=np.float32)))
    return out


@pytest.mark.parametrize("name,arr", _cases(), ids=[c[0] for c in _cases()])
def test_oracle_pre_processor_two_restatements_agree(name, arr):
    for mx in (254, 127, 1):
        a = ol.pre_processor(arr, mx)
        b = ol.pre_processor_numpy(arr, mx)
        assert a.shape == arr.shape and (a == b).all(), (name, mx)


def test_oracle_pre_processor_quirks():
    x = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 5e-324, 1.0, 2.0, -2.0], dtype=np.float64)
    q = ol.pre_processor(x)
    assert q[0] == 255 and q[2] == 255 and q[3] == 255 and q[4] == 255 and q[5] == 255   # NaN, -inf, zeros, subnormal
    assert q[1] == 0                                                                  # +inf -> ALWAYS_FILL
    assert q[7] == 254 and q[8] == 0 and q[6] == int((1.0 + 2.0) / 4.0 * 254)          # min -2, max 2
    # folds are seeded with zero: an all-positive array is scaled from 0, not from its minimum
    y = np.array([10.0, 20.0], dtype=np.float64)
    assert ol.pre_processor(y).tolist() == [127, 254]
    with pytest.raises(AssertionError):
        ol.pre_processor(y, 255)
    with pytest.raises(AssertionError):
        ol.pre_processor(y, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,arr", _cases(), ids=[c[0] for c in _cases()])
def test_gpu_pre_processor_bit_exact(name, arr):
    ge.build_hip()
    pkg = ge.load_package()
    ws = pkg.TransformBuilder.default().build_segmenting()
    for mx in (254, 127, 1):
        got = ws.pre_processor_with_max(arr, mx)
        assert got.dtype == np.uint8 and got.shape == arr.shape
        assert (got == ol.pre_processor(arr, mx)).all(), (name, mx)
    assert (ws.pre_processor(arr) == ol.pre_processor(arr, 254)).all()


@pytest.mark.gpu
def test_gpu_pre_processor_large_and_pipeline():
    # tests/integration.rs:189-204 shape: Poisson f64 field -> pre_processor -> find_local_minima -> transform
    ge.build_hip()
    pkg = ge.load_package()
    rng = np.random.default_rng(11)
    field = rng.poisson(0.85, (1000, 1000)).astype(np.float64) + rng.random((1000, 1000)) * 1e-3
    ws = pkg.TransformBuilder.default().build_merging()
    img = ws.pre_processor(field)
    assert (img == ol.pre_processor(field)).all()
    seeds = ws.find_local_minima(img)
    assert (seeds == ol.find_local_minima(img)).all()
    seg = pkg.TransformBuilder.default().build_segmenting().transform(img, seeds)
    assert (seg == ol.segment_arrival(img, seeds)).all()
    with pytest.raises(AssertionError):
        ws.pre_processor_with_max(field, 255)
