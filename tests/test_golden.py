"""The oracle against the committed golden fixtures (tests/golden/*.npz, made
by tools/make_golden.py from the oracle itself: the reference holds no fixtures
of its own). Guards the oracle against silent drift."""
import os

import numpy as np
import pytest

from scenes import GOLDEN_CASES, Inputs, mixed_oracle_render, mixed_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_oracle_reproduces_golden(name, rt, oracle):
    w, h, n, y0, y1 = GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    rgba, packed, cnt = Inputs(rt, n).oracle_render(oracle, w, h, y0=y0, y1=y1)
    assert np.array_equal(rgba[..., :3].view(np.uint32), g["rgb"].view(np.uint32))   # bit-exact, NaN-safe
    assert np.array_equal(packed, g["packed"])
    assert [cnt["primary_tests"], cnt["shadow_tests"], cnt["hit_pixels"], cnt["unshadowed"]] == g["counters"].tolist()


def test_workload_statistics_match_survey(rt, oracle):
    """C2 ~65 % hits / ~2.5k tests per pixel, C3 ~99.4 % hits /
    ~12.5k tests per pixel (shadow rays ~92 % of the work)."""
    g2 = np.load(os.path.join(GOLD, "c2_160x90_n256.npz"))["counters"].astype(np.float64)
    g3 = np.load(os.path.join(GOLD, "c3_160x90_n1024.npz"))["counters"].astype(np.float64)
    px = 160 * 90
    assert 0.60 < g2[2] / px < 0.70 and 2300 < (g2[0] + g2[1]) / px < 2700
    assert 0.99 < g3[2] / px <= 1.0 and 12000 < (g3[0] + g3[1]) / px < 13200
    assert g3[1] / (g3[0] + g3[1]) > 0.9


def test_spp4_golden(rt, oracle):
    g = np.load(os.path.join(GOLD, "spp4_96x54_n256.npz"))
    acc, packed = Inputs(rt, 256).oracle_render_spp(oracle, rt, 96, 54, 4)
    assert np.array_equal(acc.view(np.uint32), g["acc"].view(np.uint32))
    assert np.array_equal(packed, g["packed"])
    assert (acc[..., 3] == 4).all()


def test_mixed_primitives_golden(rt, oracle):
    g = np.load(os.path.join(GOLD, "mixed_160x96.npz"))
    rgba, packed, cnt = mixed_oracle_render(mixed_scene(rt), oracle, 160, 96)
    assert np.array_equal(rgba[..., :3].view(np.uint32), g["rgb"].view(np.uint32))
    assert np.array_equal(packed, g["packed"])
    assert cnt["hit_pixels"] == int(g["counters"][2]) > 160 * 96 * 0.8      # the plane fills most of the lower image
