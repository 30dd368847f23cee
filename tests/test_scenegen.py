"""Host-side scene construction of the product (no GPU): MSVC rand() replay,
the sphere scene, synthetic textures and the PPM loader."""
import ctypes as C

import numpy as np


def test_msvc_rand_sequence(rt):
    lib = rt.load_library()
    out = (C.c_int * 8)()
    assert lib.rt_msvc_rand_sequence(1, out, 8) == 0
    assert list(out) == [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358]


def test_spheres_match_oracle_generator(rt, oracle):
    n = 4096
    mine = rt.generate_spheres(n, 1)
    ref = (oracle.OSphere * n)()
    oracle.load().oracle_generate_spheres(ref, n, 1)
    a = np.frombuffer(bytes(mine), dtype=np.uint8).reshape(n, 32)
    b = np.frombuffer(bytes(ref), dtype=np.uint8).reshape(n, 32)
    assert (a == b).all()
    xyz = np.array([[s.orgin.x, s.orgin.y, s.orgin.z, s.radius] for s in mine[:n]], dtype=np.float32)
    assert xyz[:, :3].min() >= 0 and xyz[:, :3].max() <= np.float32(9.9)
    assert xyz[:, 3].max() <= np.float32(0.99) * np.float32(0.99)
    # other seeds give other scenes
    other = rt.generate_spheres(4, 2)
    assert (other[0].orgin.x, other[0].orgin.y) != (mine[0].orgin.x, mine[0].orgin.y)


def test_sphere_init_squares_radius(rt):
    s = rt.Sphere()
    rt.load_library().rt_sphere_init(C.byref(s), 1, 2, 3, 2)
    assert (s.orgin.x, s.orgin.y, s.orgin.z, s.radius) == (1, 2, 3, 4)     # kernel.cu:287


def test_synthetic_textures(rt):
    r, g, b = rt.synth_texture(0)
    assert r.shape == (512, 512)
    y, x = np.mgrid[0:512, 0:512]
    assert (r == ((64 + ((x + 2 * y) & 127)).astype(np.float32) / np.float32(255))).all()
    assert (g == ((48 + (((3 * x + y) >> 1) & 127)).astype(np.float32) / np.float32(255))).all()
    assert (b == (np.where(((x >> 5) + (y >> 5)) & 1, 200, 90).astype(np.float32) / np.float32(255))).all()
    sr, sg, sb = rt.synth_texture(1)
    assert sr.shape == (1024, 2048)
    y, x = np.mgrid[0:1024, 0:2048]
    assert (sb == ((255 - (y >> 4) - 16 * ((x >> 7) & 1)).astype(np.float32) / np.float32(255))).all()
    for p in (r, g, b, sr, sg, sb):
        assert p.min() >= 0 and p.max() <= 1


def test_ppm_loader(rt, tmp_path):
    lib = rt.load_library()
    w, h = 5, 3
    data = (np.arange(w * h * 3, dtype=np.uint8) * 5).reshape(h, w, 3)
    p = tmp_path / "t.ppm"
    p.write_bytes(b"P6\n# comment\n%d %d\n255\n" % (w, h) + data.tobytes())
    fp = C.POINTER(C.c_float)
    r, g, b = fp(), fp(), fp()
    cw, ch = C.c_int(), C.c_int()
    assert lib.rt_load_ppm(str(p).encode(), C.byref(r), C.byref(g), C.byref(b), C.byref(cw), C.byref(ch)) == 0
    assert (cw.value, ch.value) == (w, h)
    got = np.ctypeslib.as_array(r, shape=(h, w)).copy()
    assert (got == data[..., 0].astype(np.float32) / np.float32(255)).all()       # Sprite.cpp:43
    gotb = np.ctypeslib.as_array(b, shape=(h, w)).copy()
    assert (gotb == data[..., 2].astype(np.float32) / np.float32(255)).all()
    lib.rt_free_planes(r, g, b)
    bad = tmp_path / "bad.ppm"
    bad.write_bytes(b"P5\n1 1\n255\n\0")
    assert lib.rt_load_ppm(str(bad).encode(), C.byref(r), C.byref(g), C.byref(b), C.byref(cw), C.byref(ch)) != 0
    assert lib.rt_load_ppm(b"/nonexistent.ppm", C.byref(r), C.byref(g), C.byref(b), C.byref(cw), C.byref(ch)) != 0


def test_offscreen_window(rt, tmp_path):
    lib = rt.load_library()
    assert lib.rt_offscreen_resize(4, 2) == 0
    assert lib.rt_offscreen_width() == 4 and lib.rt_offscreen_height() == 2
    px = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(2, 4))
    assert (px == 0).all()
    out = tmp_path / "o.ppm"
    assert lib.rt_offscreen_write_ppm(str(out).encode()) == 0
    assert out.read_bytes().startswith(b"P6\n4 2\n255\n") and len(out.read_bytes()) == 11 + 24


def test_band_rows_partition(rt):
    for h in (2160, 4320, 1080, 7):
        for world in (1, 2, 3, 4, 8):
            rows = [rt.band_rows(h, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            for (a0, a1), (b0, b1) in zip(rows, rows[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in rows]
            assert max(sizes) - min(sizes) <= 1
