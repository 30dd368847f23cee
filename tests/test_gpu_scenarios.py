"""Differential GPU-vs-oracle tests on scenes chosen to stress the exactness
arguments of the HIP kernel (conservative beams, grouping, shortcuts, fallbacks)
and the edge cases of the reference's semantics. All comparisons are bit-exact,
for the culled kernel AND the brute-force loops."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


class Scn:
    """Free-form scene: explicit spheres (x,y,z,ctor_r), lights, camera, textures."""

    def __init__(self, rt, spheres, lights=None, cam=None, tex=None, sky=None, sky_size=10000.0, aspect=None,
                 planes=(), cubes=()):
        self.rt = rt
        lib = rt.load_library()
        self.n = len(spheres)
        self.spheres = (rt.Sphere * max(self.n, 1))()
        for i, (x, y, z, r) in enumerate(spheres):
            lib.rt_sphere_init(C.byref(self.spheres[i]), float(x), float(y), float(z), float(r))
        self.n_planes, self.n_cubes = len(planes), len(cubes)
        self.planes = (rt.Plane * max(self.n_planes, 1))()
        for i, v in enumerate(planes):
            lib.rt_plane_init(C.byref(self.planes[i]), *[float(x) for x in v])
        self.cubes = (rt.Cube * max(self.n_cubes, 1))()
        for i, v in enumerate(cubes):
            lib.rt_cube_init(C.byref(self.cubes[i]), *[float(x) for x in v])
        if lights is None:
            self.lights, self.n_lights = rt.default_lights(), 3
        else:
            self.n_lights = len(lights)
            self.lights = (rt.Light * max(self.n_lights, 1))()
            for i, (p, size, r, g, b) in enumerate(lights):
                self.lights[i] = rt.Light(rt.Vec3(*[float(v) for v in p]), size, r, g, b)
        self.cam = cam if cam is not None else rt.default_camera()
        self.tex = tex if tex is not None else rt.synth_texture(0)
        self.sky = sky if sky is not None else rt.synth_texture(1)
        self.sky_box = rt.sky_sphere(sky_size)
        self.aspect = rt.default_aspect() if aspect is None else aspect

    def scene(self):
        s = self.rt.Scene()
        s.set_spheres(self.spheres, self.n)
        s.set_texture(self.tex)
        s.set_sky(self.sky_box, self.sky)
        s.set_lights(self.lights, self.n_lights)
        if getattr(self, "n_planes", 0):
            s.set_planes(self.planes, self.n_planes)
        if getattr(self, "n_cubes", 0):
            s.set_cubes(self.cubes, self.n_cubes)
        return s

    def check(self, w, h, tiles=(8,), spp=1, nthreads=16):
        import oracle_py
        import torch
        acc = None
        lib = self.rt.load_library()
        for k in range(spp):
            ox, oy = C.c_double(), C.c_double()
            assert lib.rt_sample_offset(k, spp, C.byref(ox), C.byref(oy)) == 0
            rgba, packed, cnt = oracle_py.render(self.spheres, self.n, self.tex, self.sky, self.sky_box, self.lights,
                                                 self.n_lights, self.cam, w, h, self.aspect,
                                                 off=(ox.value, oy.value), nthreads=nthreads,
                                                 cubes=getattr(self, "cubes", None), n_cubes=getattr(self, "n_cubes", 0),
                                                 planes=getattr(self, "planes", None), n_planes=getattr(self, "n_planes", 0))
            acc = rgba if acc is None else (acc + rgba).astype(np.float32)
        sc = self.scene()
        for tile in tiles:
            for cull in (True, False):
                out = sc.render(w, h, cull=cull, tile=tile, spp=spp, cam=self.cam, aspect=self.aspect)
                torch.cuda.synchronize()
                got = out["rgba"].cpu().numpy()
                assert np.array_equal(_bits(got), _bits(acc)), (tile, cull, int((_bits(got) != _bits(acc)).any(axis=2).sum()))
                if spp == 1:
                    assert np.array_equal(out["packed"].cpu().numpy().view(np.uint32), packed), (tile, cull)
        return cnt


def _cam(rt, org=(4, 3, 10), yaw=180.0, pitch=-20.0):
    return rt.Camera(rt.Vec3(*org), rt.Vec3(0, 0, 1), 0.0, yaw, pitch)


def test_empty_scene_is_all_sky(rt, gpu):
    cnt = Scn(rt, []).check(96, 64, tiles=(8, 64))
    assert cnt["hit_pixels"] == 0


def test_camera_inside_a_sphere_negative_root(rt, gpu):
    # F4: every pixel hits with a NEGATIVE t; new_org lies behind the camera
    cnt = Scn(rt, [(4, 3, 9, 2.0)]).check(64, 48)
    assert cnt["hit_pixels"] == 64 * 48


def test_duplicate_and_zero_radius_spheres(rt, gpu):
    # identical spheres: the first index wins the tie (strict <, kernel.cu:1335);
    # radius 0 spheres can only "hit" through rounding noise
    sph = [(4, 2, 5, 0.9), (4, 2, 5, 0.9), (5, 3, 4, 0.0), (3, 3, 6, 0.0), (4.5, 2.5, 5.5, 0.7), (4, 2, 5, 0.9)]
    Scn(rt, sph).check(96, 64, tiles=(8, 16))


@pytest.mark.parametrize("n_lights", [0, 1, 8])
def test_light_counts(rt, gpu, n_lights):
    rng = np.random.default_rng(n_lights)
    lights = [((rng.uniform(-30, 30), rng.uniform(5, 30), rng.uniform(-30, 30)), 20.0, *rng.uniform(0, 1, 3))
              for _ in range(n_lights)]
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.2, 1.0),) for _ in range(48)]
    Scn(rt, sph, lights=lights).check(96, 64)


def test_too_many_lights_is_refused(rt, gpu):
    s = rt.Scene()
    lights = (rt.Light * 9)()
    with pytest.raises(rt.RtError):
        s.set_lights(lights, 9)


def test_degenerate_lights(rt, gpu):
    """A light at the world origin has no beam axis and one inside the scene
    makes the sample cone wide: culling must step aside (whole-table fallback),
    results stay exact. Also a light straight above along +z (toL = (0,0,1):
    the rotation axis is the zero vector)."""
    rng = np.random.default_rng(5)
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.2, 0.9),) for _ in range(64)] + [(5, 5, 2, 0.8)]
    lights = [((0, 0, 0), 20.0, 1, 0.5, 0.2), ((5, 5, 5), 3.0, 0.3, 1, 0.3), ((5, 5, 60), 10.0, 0.2, 0.2, 1),
              ((1, 2, 0.5), 1.0, 1, 1, 1)]
    Scn(rt, sph, lights=lights).check(96, 64, tiles=(8, 32))


@pytest.mark.parametrize("yaw,pitch,org", [(0.0, 0.0, (5, 5, -8)), (90.0, -45.0, (-6, 12, 5)), (180.0, 89.0, (5, -15, 5)),
                                           (37.5, 12.25, (20, 9, 22)), (180.0, -20.0, (4, 3, 60))])
def test_camera_poses(rt, gpu, yaw, pitch, org):
    # the far camera makes spheres a few pixels wide: many distinct spheres per
    # tile, which exercises the grouping by closest sphere
    rng = np.random.default_rng(11)
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.1, 1.0),) for _ in range(200)]
    Scn(rt, sph, cam=_cam(rt, org, yaw, pitch)).check(112, 80, tiles=(8,))


def test_one_scene_many_cameras_and_moving_lights(rt, gpu):
    """The culling tables derived from the eye (cones) and from each light (columns) are
    cached in the scene: moving the camera or a light between frames of the SAME scene must
    rebuild them. Every frame is compared with the brute-force loops, the last ones with the
    oracle as well."""
    import torch
    rng = np.random.default_rng(21)
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.1, 1.0),) for _ in range(300)]
    base = Scn(rt, sph)
    sc = base.scene()
    cams = [_cam(rt), _cam(rt, (5, 5, -8), 0.0, 0.0), _cam(rt, (-6, 12, 5), 90.0, -45.0), _cam(rt),
            _cam(rt, (4.5, 4.5, 4.5), 10.0, 5.0)]
    for k, cam in enumerate(cams):
        if k == 3:    # a light moves, the others stay
            lights = (rt.Light * 3)(*rt.default_lights())
            lights[1] = rt.Light(rt.Vec3(-15.0, 25.0, 3.0), 20.0, 0.0, 0.0, 1.0)
            sc.set_lights(lights, 3)
        a = sc.render(96, 64, cam=cam, cull=True)
        b = sc.render(96, 64, cam=cam, cull=False)
        torch.cuda.synchronize()
        assert torch.equal(a["rgba"], b["rgba"]) and torch.equal(a["packed"], b["packed"]), k
    Scn(rt, sph, cam=cams[-1]).check(96, 64)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (67, 45), (130, 3), (9, 70)])
def test_odd_frame_sizes(rt, gpu, w, h):
    from scenes import Inputs
    inp = Inputs(rt, 256)
    Scn.check(_as_scn(rt, inp), w, h, tiles=(8, 16, 32, 64))


def _as_scn(rt, inp):
    s = Scn.__new__(Scn)
    s.rt, s.n, s.spheres, s.lights, s.n_lights = rt, inp.n, inp.spheres, inp.lights, inp.n_lights
    s.cam, s.tex, s.sky, s.sky_box, s.aspect = inp.cam, inp.tex, inp.sky, inp.sky_box, inp.aspect
    return s


@pytest.mark.parametrize("n,seed", [(1, 3), (63, 4), (65, 5), (100, 6), (1000, 7), (1025, 8), (3000, 9)])
def test_random_seeds_and_counts(rt, gpu, n, seed):
    from scenes import Inputs
    Scn.check(_as_scn(rt, Inputs(rt, n, seed)), 80, 56)


def test_table_in_global_memory_matches_lds_staging(rt, gpu):
    """By default the kernel reads the table from global memory and keeps only
    the tiles' survivor lists in LDS; rt_launch_opts.table_lds stages the whole table per
    workgroup instead (north_star's first design). Neither mode may change a bit."""
    import torch
    from scenes import Inputs
    for n in (100, 1024):
        sc = Inputs(rt, n, 3).scene()
        a = sc.render(160, 90, table_lds=True)
        b = sc.render(160, 90)
        c = sc.render(160, 90, cull=False)
        d = sc.render(160, 90, cull=False, table_lds=True)
        e = sc.render(160, 90, spp=4, table_lds=True)
        f = sc.render(160, 90, spp=4)
        torch.cuda.synchronize()
        assert torch.equal(a["rgba"], b["rgba"]) and torch.equal(a["packed"], b["packed"])
        assert torch.equal(a["rgba"], c["rgba"]) and torch.equal(a["packed"], c["packed"])
        assert torch.equal(a["rgba"], d["rgba"]) and torch.equal(a["packed"], d["packed"])
        assert torch.equal(e["rgba"], f["rgba"]) and torch.equal(e["packed"], f["packed"])


def test_large_sphere_counts(rt, gpu):
    """Maximum sizes: 9000 and 20000 spheres (global-memory table; 9000 also with
    the table staged in LDS, which it still fits) equal the brute-force loops. The
    oracle agrees on a tiny frame. 9000 spheres are beyond the device-side eye-cone
    builder (RT_EYE_DEVICE_MAX), so this is also the host-built table."""
    import torch
    from scenes import Inputs
    for n in (9000, 20000):
        inp = Inputs(rt, n, 12)
        sc = inp.scene()
        a = sc.render(24, 16, cull=True)
        b = sc.render(24, 16, cull=False)
        torch.cuda.synchronize()
        assert torch.equal(a["rgba"], b["rgba"]) and torch.equal(a["packed"], b["packed"])
        if n == 9000:
            c = sc.render(24, 16, cull=True, table_lds=True)
            torch.cuda.synchronize()
            assert torch.equal(a["rgba"], c["rgba"]) and torch.equal(a["packed"], c["packed"])
    Scn.check(_as_scn(rt, Inputs(rt, 20000, 12)), 16, 8)
    s = rt.Scene()
    n = (1 << 22) + 1
    big = rt.generate_spheres(n, 1)
    with pytest.raises(rt.RtError) as e:
        s.set_spheres(big, n)
    assert "exceed" in str(e.value)


@pytest.mark.parametrize("spp", [2, 3, 16])
def test_sample_counts(rt, gpu, spp):
    from scenes import Inputs
    Scn.check(_as_scn(rt, Inputs(rt, 128, 2)), 48, 32, spp=spp)


def test_small_and_odd_textures(rt, gpu):
    rng = np.random.default_rng(3)
    tex = [rng.integers(0, 256, (7, 13)).astype(np.float32) / np.float32(255) for _ in range(3)]
    sky = [rng.integers(0, 256, (1, 1)).astype(np.float32) / np.float32(255) for _ in range(3)]
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.3, 1.0),) for _ in range(80)]
    Scn(rt, sph, tex=tex, sky=sky).check(96, 64)


def test_nan_and_huge_spheres(rt, gpu):
    """NaN centre: every test against it is false (NaN compares). A sphere far
    outside the box and one that swallows half the scene stress the beam padding."""
    rng = np.random.default_rng(8)
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.2, 0.9),) for _ in range(40)]
    sph += [(float("nan"), 1, 1, 0.5), (200, 150, -300, 9.0), (5, -40, 5, 6.2), (5, 5, 5, 1.6)]
    Scn(rt, sph).check(96, 64)


def test_non_finite_and_enclosing_spheres_in_the_culling_tables(rt, gpu):
    """Above 64 spheres the host builds the eye-cone and per-light column tables: entries with
    NaN/inf centres or radii, a far giant and spheres around the camera (default Org (4,3,10))
    and around a light must land in blocks that are always examined."""
    rng = np.random.default_rng(18)
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.2, 0.9),) for _ in range(150)]
    sph += [(float("nan"), 1, 1, 0.5), (2, float("inf"), 3, 0.4), (3, 3, 3, float("nan")), (6, 2, 7, float("inf")),
            (200, 150, -300, 9.0), (5, -40, 5, 6.2), (4, 3, 10.9, 1.3), (4.2, 3.1, 10.2, 0.5), (20, 20, 20, 1.5)]
    Scn(rt, sph).check(96, 64)
    # the same entries first in the list (ties and order of the primary list)
    Scn(rt, sph[150:] + sph[:150]).check(64, 40)


def test_texture_values_outside_unit_range(rt, gpu):
    """Brightness > 1 clamps at 255 in rgbToInt (kernel.cu:548-553); negative and
    non-finite texels must not be skipped by the facing-away shortcut."""
    rng = np.random.default_rng(4)
    tex = [rng.uniform(-2, 6, (16, 16)).astype(np.float32) for _ in range(3)]
    tex[1][3, 5] = np.float32("inf")
    tex[2][8, 2] = np.float32("nan")
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.4, 1.0),) for _ in range(60)]
    Scn(rt, sph, tex=tex).check(96, 64)


# ---------------------------------------------------------------- cubes and planes
def test_reference_plane_under_the_spheres(rt, gpu):
    """The reference's own plane ({0,-4,0}, normal +y, kernel.cu:1187) below the
    default sphere scene: plane pixels take texel (0.5,0.5) and are shadowed by spheres."""
    from scenes import Inputs
    inp = Inputs(rt, 256)
    s = _as_scn(rt, inp)
    lib = rt.load_library()
    s.n_planes, s.planes = 1, (rt.Plane * 1)()
    lib.rt_plane_init(C.byref(s.planes[0]), 0.0, -4.0, 0.0, 0.0, 1.0, 0.0)
    s.n_cubes, s.cubes = 0, (rt.Cube * 1)()
    cnt = s.check(160, 90, tiles=(8, 32))
    assert cnt["hit_pixels"] > 9401          # more than the spheres alone (golden C2 count)


def test_cubes_planes_and_spheres_mixed(rt, gpu):
    rng = np.random.default_rng(21)
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.3, 0.9),) for _ in range(120)]
    cubes = []
    for _ in range(12):
        a = rng.uniform(0, 9, 3)
        cubes.append(tuple(a) + tuple(a + rng.uniform(0.3, 1.5, 3)))
    cubes.append((8, 0, 8, 6, 2, 6))                       # corners given in the "wrong" order
    planes = [(0, -1, 0, 0, 1, 0), (0, 0, -2, 0.2, 0.1, 1.0), (12, 0, 0, -1, 0, 0)]   # one with a non-unit normal
    Scn(rt, sph, planes=planes, cubes=cubes).check(128, 80, tiles=(8, 16))


def test_only_cubes_and_planes_camera_inside_a_cube(rt, gpu):
    # no spheres at all; the camera (4,3,~9) sits inside the first cube: negative tmin hit
    cubes = [(3, 2, 8, 5, 4, 10), (0, 0, 0, 2, 2, 2), (6, 1, 1, 7, 5, 2)]
    planes = [(0, -3, 0, 0, 1, 0)]
    cnt = Scn(rt, [], planes=planes, cubes=cubes).check(96, 64)
    assert cnt["hit_pixels"] == 96 * 64


def test_axis_aligned_rays_hit_cube_slabs_exactly(rt, gpu):
    # yaw 0 / pitch 0 and a centred pixel give direction components that are exactly 0
    # for some lanes: 1/0 = inf and 0*inf = NaN flow through the min/max macros
    cubes = [(3, 2, 12, 5, 4, 14), (4, 3, 15, 6, 5, 16)]
    cam = _cam(rt, (4, 3, 2), 0.0, 0.0)
    Scn(rt, [(4, 3, 20, 1.0)], cubes=cubes, cam=cam).check(64, 64, tiles=(8, 64))


def test_too_many_planes_or_cubes_are_refused(rt, gpu):
    s = rt.Scene()
    with pytest.raises(rt.RtError):
        s.set_planes((rt.Plane * 65)(), 65)
    with pytest.raises(rt.RtError):
        s.set_cubes((rt.Cube * 257)(), 257)


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_random_scenes(rt, gpu, seed):
    """Random scenes (sphere count/size/placement, lights anywhere including inside
    the scene, camera pose, texture sizes, optional cubes/planes): culled kernel ==
    brute-force kernel == oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(0, 300))
    ext = float(rng.choice([3.0, 10.0, 40.0]))
    sph = [tuple(rng.uniform(-ext * 0.2, ext, 3)) + (float(rng.choice([rng.uniform(0, 1), rng.uniform(1, 1.8), 0.05])),)
           for _ in range(n)]
    lights = []
    for _ in range(int(rng.integers(0, 5))):
        pos = rng.uniform(-30, 30, 3) if rng.random() < 0.7 else rng.uniform(0, ext, 3)
        lights.append((tuple(pos), float(rng.uniform(0.5, 25)), *[float(v) for v in rng.uniform(0, 1.5, 3)]))
    cam = _cam(rt, tuple(rng.uniform(-5, ext + 5, 3)), float(rng.uniform(0, 360)), float(rng.uniform(-60, 60)))
    th, tw = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    tex = [rng.integers(0, 256, (th, tw)).astype(np.float32) / np.float32(255) for _ in range(3)]
    planes, cubes = [], []
    if rng.random() < 0.4:
        for _ in range(int(rng.integers(1, 3))):
            nrm = rng.normal(size=3)
            planes.append(tuple(rng.uniform(-5, ext, 3)) + tuple(nrm))
        for _ in range(int(rng.integers(0, 6))):
            a = rng.uniform(0, ext, 3)
            cubes.append(tuple(a) + tuple(a + rng.uniform(0.1, 2.0, 3)))
    Scn(rt, sph, lights=lights, cam=cam, tex=tex, planes=planes, cubes=cubes).check(64, 48, tiles=(8,))


@pytest.mark.parametrize("seed", list(range(16)))
def test_fuzz_occluder_shortcuts(rt, gpu, seed):
    """Scenes built to sit on the decision boundaries of the beam-level shortcuts:
    big spheres that just cover / just fail to cover a tile's shadow beam, lights at
    many distances, dense and sparse layers. Culled == brute force == oracle."""
    rng = np.random.default_rng(7000 + seed)
    n_small = int(rng.integers(20, 150))
    sph = [tuple(rng.uniform(0, 10, 3)) + (float(rng.uniform(0.1, 0.7)),) for _ in range(n_small)]
    for _ in range(int(rng.integers(1, 6))):          # large occluders (effective radius r^2 up to ~3)
        sph.append(tuple(rng.uniform(-2, 12, 3)) + (float(rng.uniform(0.9, 1.7)),))
    rng.shuffle(sph)
    lights = []
    for _ in range(3):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        lights.append((tuple(d * float(rng.choice([6.0, 15.0, 40.0, 200.0]))), float(rng.uniform(1, 30)),
                       *[float(v) for v in rng.uniform(0.2, 1.0, 3)]))
    cam = _cam(rt, tuple(rng.uniform(0, 10, 3) + np.array([0, 2, 6])), float(rng.uniform(150, 210)), float(rng.uniform(-35, 5)))
    Scn(rt, sph, lights=lights, cam=cam).check(96, 64, tiles=(8, 16))


@pytest.mark.parametrize("cam_up", [0.9, -0.9])
def test_camera_between_concentric_spheres(rt, gpu, cam_up):
    """The camera sits between two CONCENTRIC spheres, two lights and seventy small occluders in the shell with it.
    intersect() returns the near root even when it is negative (kernel.cu:1335 keeps the smallest t of any sign), so
    every pixel's closest hit is the OUTER sphere, behind the camera -- the inner one is never a closest hit, which is
    what lets the kernel form its shading groups (and pick their occluder lists) by the centre of the closest sphere."""
    rng = np.random.default_rng(91)
    c = np.array([20.0, 15.0, 25.0])
    sph = [tuple(c) + (1.0,), tuple(c) + (3.0,)]
    while len(sph) < 72:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        sph.append(tuple(c + d * float(rng.uniform(1.3, 2.7))) + (float(rng.uniform(0.05, 0.2)),))
    lights = [(tuple(c + np.array([1.8, 1.5, 0.5])), 1.0, 1.0, 0.9, 0.8), (tuple(c + np.array([-1.2, -1.9, 0.8])), 3.0, 0.3, 0.6, 1.0)]
    cam = _cam(rt, tuple(c + np.array([0.0, cam_up, 2.2])), 180.0, -22.0 * np.sign(cam_up))
    cnt = Scn(rt, sph, lights=lights, cam=cam).check(128, 96, tiles=(8, 16))
    assert cnt["hit_pixels"] == 128 * 96
