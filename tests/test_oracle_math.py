"""The oracle's bit-reproducible transcendentals vs the host libm: the two
definitions must agree to within 1 ulp everywhere (they agree exactly almost
everywhere), which bounds how much the choice of libm can matter."""
import numpy as np


def _ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def _map1(fn, xs):
    return np.array([fn(float(x)) for x in xs], dtype=np.float32)


def test_uses_expected_math(oracle):
    assert oracle.load(False).oracle_uses_libm() == 0
    assert oracle.load(True).oracle_uses_libm() == 1


def test_sin_cos(oracle):
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-7, 7, 40000), rng.uniform(-1e-3, 1e-3, 2000),
                         np.linspace(-100, 100, 5001), [0.0, 3.1415, -0.34905556, 1.5707964, 3.1415927]]).astype(np.float32)
    lib = oracle.load()
    for name, ref in (("oracle_cosf", np.cos), ("oracle_sinf", np.sin)):
        got = _map1(getattr(lib, name), xs)
        want = ref(xs.astype(np.float64)).astype(np.float32)   # correctly rounded reference
        d = _ulp_diff(got, want)
        assert d.max() <= 1, (name, d.max())
        assert (d == 0).mean() > 0.9999
    assert np.isnan(lib.oracle_cosf(float("inf"))) and np.isnan(lib.oracle_sinf(float("nan")))


def test_acos(oracle):
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.uniform(-1, 1, 40000), 1 - rng.uniform(0, 1e-5, 2000), -1 + rng.uniform(0, 1e-5, 2000),
                         [-1.0, 1.0, 0.0, -0.0, 0.5, -0.5]]).astype(np.float32)
    lib = oracle.load()
    got = _map1(lib.oracle_acosf, xs)
    want = np.arccos(xs.astype(np.float64)).astype(np.float32)
    d = _ulp_diff(got, want)
    assert d.max() <= 1 and (d == 0).mean() > 0.9999
    assert lib.oracle_acosf(1.0) == 0.0
    assert lib.oracle_acosf(-1.0) == np.float32(np.pi)
    # one ulp outside [-1,1] -> NaN (the NaN-direction case of kernel.cu:1466)
    assert np.isnan(lib.oracle_acosf(float(np.nextafter(np.float32(1), np.float32(2)))))
    assert np.isnan(lib.oracle_acosf(float("nan")))


def test_atan2(oracle):
    rng = np.random.default_rng(9)
    ys = rng.uniform(-1, 1, 40000).astype(np.float32)
    xs = rng.uniform(-1, 1, 40000).astype(np.float32)
    lib = oracle.load()
    got = np.array([lib.oracle_atan2f(float(y), float(x)) for y, x in zip(ys, xs)], dtype=np.float32)
    want = np.arctan2(ys.astype(np.float64), xs.astype(np.float64)).astype(np.float32)
    d = _ulp_diff(got, want)
    assert d.max() <= 1 and (d == 0).mean() > 0.9999
    # special cases that decide texels at the seams
    assert lib.oracle_atan2f(0.0, 1.0) == 0.0
    assert lib.oracle_atan2f(0.0, -1.0) == np.float32(np.pi)
    assert lib.oracle_atan2f(-0.0, -1.0) == -np.float32(np.pi)
    assert lib.oracle_atan2f(1.0, 0.0) == np.float32(np.pi / 2)
    assert lib.oracle_atan2f(-1.0, 0.0) == -np.float32(np.pi / 2)
    assert lib.oracle_atan2f(0.0, 0.0) == 0.0
    assert np.isnan(lib.oracle_atan2f(float("nan"), 1.0))


def test_portable_matches_libm_flavour(oracle):
    """Same frame through both flavours of the oracle: continuous outputs within
    1e-5 relative on pixels whose discrete decisions agree; flips are counted."""
    import rt_amd
    from scenes import Inputs
    rt = rt_amd.load()
    inp = Inputs(rt, 256)
    a, pa, _ = inp.oracle_render(oracle, 160, 90)
    b, pb, _ = inp.oracle_render(oracle, 160, 90, libm=True)
    diff = np.abs(a[..., :3] - b[..., :3])
    scale = np.maximum(np.abs(a[..., :3]), np.abs(b[..., :3]))
    rel = np.where(scale > 0, diff / np.maximum(scale, 1e-30), 0.0)
    flipped = (rel > 1e-5).any(axis=2)
    assert flipped.mean() <= 1e-3          # budget for silhouette/shadow/texel flips
    assert (rel[~flipped] <= 1e-5).all()
