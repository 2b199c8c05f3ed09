"""Checks the oracle (CPU) and the HIP path (-m gpu) against the committed fixtures
tests/golden/oracle_fixtures.npz (made by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge
import cases
import oracle_lib as ol

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_fixtures.npz"), allow_pickle=False)
SIZES = (16, 64, 256)
ADV = sorted({k[4:-4] for k in FX.files if k.startswith("adv_") and k.endswith("_img")})


def _field(n):
    return cases.field(n, n, int(FX[f"f{n}_seed"][0]))


@pytest.mark.parametrize("n", SIZES)
def test_oracle_matches_fixtures(n):
    img = _field(n)
    seeds = ol.find_local_minima(img)
    assert (seeds == FX[f"f{n}_seeds"]).all()
    assert (ol.segment(img, seeds) == FX[f"f{n}_seg"]).all()
    assert (ol.segment_arrival(img, seeds, edge=True) == FX[f"f{n}_seg_edge"]).all()
    assert (ol.segment_par(img, seeds, max_level=127)[0] == FX[f"f{n}_seg_max127"]).all()
    assert (ol.merge_arrival(img, seeds) == FX[f"f{n}_merge_final"]).all()


def test_oracle_matches_adversarial_fixtures():
    assert len(ADV) >= 15
    for name in ADV:
        img, seeds = FX[f"adv_{name}_img"], FX[f"adv_{name}_seeds"]
        assert (ol.segment_arrival(img, seeds) == FX[f"adv_{name}_seg"]).all(), name
        assert (ol.segment(img, seeds, edge=True) == FX[f"adv_{name}_seg_edge"]).all(), name


@pytest.fixture(scope="module")
def pkg():
    ge.build_hip()
    return ge.load_package()


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES)
def test_gpu_matches_fixtures(pkg, n):
    img = _field(n)
    seg = pkg.TransformBuilder.default().build_segmenting()
    seeds = seg.find_local_minima(img)
    assert (seeds == FX[f"f{n}_seeds"]).all()
    assert (seg.transform(img, seeds) == FX[f"f{n}_seg"]).all()
    assert (pkg.TransformBuilder.new().enable_edge_correction().build_segmenting().transform(img, seeds) == FX[f"f{n}_seg_edge"]).all()
    assert (pkg.TransformBuilder.new().set_max_water_lvl(127).build_segmenting().transform(img, seeds) == FX[f"f{n}_seg_max127"]).all()
    mer = pkg.TransformBuilder.default().build_merging()
    assert (mer.transform_final(img, seeds) == FX[f"f{n}_merge_final"]).all()
    sparse = mer.transform_to_list_sparse(img, seeds)
    for lvl in (0, 50, 127, 200, 254):
        assert (np.sort(sparse[lvl][3]) == FX[f"f{n}_merge_sizes_l{lvl}"]).all(), lvl


@pytest.mark.gpu
def test_gpu_matches_adversarial_fixtures(pkg):
    for name in ADV:
        img, seeds = FX[f"adv_{name}_img"], FX[f"adv_{name}_seeds"]
        for engine in (pkg.ENGINE_FUSED, pkg.ENGINE_SWEEP):
            got = pkg.TransformBuilder.new().set_engine(engine).build_segmenting().transform(img, seeds)
            assert (got == FX[f"adv_{name}_seg"]).all(), (name, engine)
        got = pkg.TransformBuilder.new().enable_edge_correction().build_segmenting().transform(img, seeds)
        assert (got == FX[f"adv_{name}_seg_edge"]).all(), name
