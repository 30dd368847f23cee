"""Known-answer tests that pin the CPU oracle.

The reference has no tests or fixtures, so these hand-derived values -- each
computed from the cited formula, independently of the restatement -- are what
anchors the oracle (and catch misreadings of quirks F3/F4)."""
import ctypes as C
import math

import numpy as np
import pytest


def _sphere(o, x, y, z, r):
    s = o.OSphere()
    o.load().oracle_make_sphere(C.byref(s), x, y, z, r)
    return s


def _isect(o, s, org, d):
    ray = o.ORay(o.OVec3(*org), o.OVec3(*d))
    t = C.c_float()
    hit = o.load().oracle_sphere_intersect(C.byref(s), C.byref(ray), C.byref(t))
    return bool(hit), t.value


def test_struct_abi(oracle):
    # kernel.cu:1214,1219 hard-code 40 and 32 bytes; camera 36, light 28, ray 24
    assert C.sizeof(oracle.OSphere) == 32
    assert oracle.OSphere.orgin.offset == 8 and oracle.OSphere.radius.offset == 24
    assert C.sizeof(oracle.OCamera) == 36
    assert C.sizeof(oracle.OLight) == 28
    assert C.sizeof(oracle.ORay) == 24


def test_front_hit(oracle):
    # kernel.cu:332-351: O=0, D=+z, c=(0,0,5), ctor r=1 -> t = 4
    hit, t = _isect(oracle, _sphere(oracle, 0, 0, 5, 1), (0, 0, 0), (0, 0, 1))
    assert hit and t == 4.0


def test_radius_squared_twice(oracle):
    # F3: ctor stores r*r (kernel.cu:287), intersect uses radius*radius (:334):
    # ctor r=2 behaves as radius 4 -> t = 5 - 4 = 1 (not 3)
    s = _sphere(oracle, 0, 0, 5, 2)
    assert s.radius == 4.0
    hit, t = _isect(oracle, s, (0, 0, 0), (0, 0, 1))
    assert hit and t == 1.0


def test_miss_is_nan(oracle):
    # no discriminant test: sqrt of a negative -> NaN -> both compares false
    hit, t = _isect(oracle, _sphere(oracle, 0, 5, 5, 1), (0, 0, 0), (0, 0, 1))
    assert not hit and math.isnan(t)


def test_inside_returns_negative_root(oracle):
    # F4 (kernel.cu:342-351): origin inside -> far root 1 >= 1e-4 -> t = min -> -1
    hit, t = _isect(oracle, _sphere(oracle, 0, 0, 0, 1), (0, 0, 0), (0, 0, 1))
    assert hit and t == -1.0


def test_behind_is_false(oracle):
    hit, t = _isect(oracle, _sphere(oracle, 0, 0, -5, 1), (0, 0, 0), (0, 0, 1))
    assert not hit and t == -4.0


def test_skybox_root(oracle):
    # kernel.cu:1122,1151: ctor r=10000 -> field 1e8 -> effective radius 1e8;
    # the camera is inside, so the NEGATIVE root (~ -1e8) comes back.
    s = _sphere(oracle, 0, 0, 0, 10000)
    assert s.radius == 1.0e8
    hit, t = _isect(oracle, s, (4, 3, 9), (0, 0, -1))
    assert hit and abs(t - (-99999990.0)) <= 16.0


def test_t_threshold(oracle):
    # t >= 0.0001 is a double compare (kernel.cu:342): a far root of exactly
    # float(1e-4) (< 1e-4 as a double) is rejected, the next float is accepted.
    lo = np.float32(1e-4)
    assert float(lo) < 1e-4
    hi = np.nextafter(lo, np.float32(1))
    for tfar, expect in ((lo, False), (hi, True)):
        # origin at the sphere's far pole minus tfar: choose c so far root = tfar
        # with D=+z, O=0: far root = c + R. R = 1 -> c = tfar - 1
        s = _sphere(oracle, 0, 0, float(np.float32(tfar) - np.float32(1)), 1)
        hit, t = _isect(oracle, s, (0, 0, 0), (0, 0, 1))
        # rounding of c may move the root by an ulp; only check the clear cases
        far = np.float32(s.orgin.z) + np.float32(1)
        if float(far) >= 1e-4:
            assert hit
        elif far != 0:
            assert not hit


def test_pack(oracle):
    lib = oracle.load()
    assert lib.oracle_rgb_to_int(300, 128, 0) == 0xFF8000      # kernel.cu:547-555
    assert lib.oracle_rgb_to_int(254, 254, 254) == 0xFEFEFE
    assert lib.oracle_pack_color(1.0, 0.5, 0.0) == (254 << 16) + (127 << 8)
    assert lib.oracle_pack_color(float("nan"), 2.0, 0.0) == (255 << 8)  # NaN -> 0, clamp > 255
    assert lib.oracle_f2i(float("nan")) == 0
    assert lib.oracle_f2i(1e20) == 2**31 - 1
    assert lib.oracle_f2i(-3.9) == -3


def test_constants(oracle):
    lib = oracle.load()
    aspect = lib.oracle_default_aspect()
    assert aspect == np.float32(0.9999537)                      # kernel.cu:1701
    assert np.float32(-1) / np.float32(aspect) == np.float32(-1.0000464)   # eye.z, :1629
    assert np.float32(180 * (3.1415 / 180)) == np.float32(3.1415)          # :249
    assert np.float32(np.float32(-20) * (3.1415 / 180)) == np.float32(-0.34905556)   # :250


def test_msvc_rand_prefix_and_scene(oracle):
    lib = oracle.load()
    lib.oracle_msvc_srand(1)
    seq = [lib.oracle_msvc_rand() for _ in range(8)]
    assert seq == [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358]
    sph = (oracle.OSphere * 2)()
    lib.oracle_generate_spheres(sph, 2, 1)
    assert (sph[0].orgin.x, sph[0].orgin.y, sph[0].orgin.z) == (np.float32(4.1), np.float32(6.7), np.float32(3.4))
    assert sph[0].radius == 0.0
    assert (sph[1].orgin.x, sph[1].orgin.y, sph[1].orgin.z) == (np.float32(6.9), np.float32(2.4), np.float32(7.8))
    assert sph[1].radius == np.float32(np.float32(0.58) * np.float32(0.58)) == np.float32(0.33639997)


def test_primary_ray_is_off_centre(oracle):
    # kernel.cu:1624-1625: the "- 1" sits outside the aspect scaling, so
    # dy spans (-1, -1 + 2*aspect*H/W): the image is vertically off-centre.
    lib = oracle.load()
    cam = oracle.OCamera(oracle.OVec3(4, 3, 10), oracle.OVec3(0, 0, 1), 0, 0.0, 0.0)   # no rotation
    aspect = lib.oracle_default_aspect()
    r = oracle.ORay()
    lib.oracle_primary_ray(0, 0, 8, 4, aspect, C.byref(cam), 0.5, 0.5, C.byref(r))
    # dx = aspect*(2*0.5/8) - 1 ; dy = aspect*(2*0.5/4)*(4/8) - 1
    dx = np.float32(float(aspect) * (2 * 0.5 / 8.0) - 1)
    dy = np.float32(float(aspect) * (2 * 0.5 / 4.0) * float(np.float32(4) / np.float32(8)) - 1)
    nz = -(np.float32(-1) / np.float32(aspect))
    l = np.sqrt(np.float32(np.float32(dx * dx + dy * dy) + nz * nz))
    assert r.Dir.x == np.float32(dx / l) and r.Dir.y == np.float32(dy / l)
    assert r.Org.z == np.float32(np.float32(-1) / np.float32(aspect) + np.float32(10))
    # yaw 180 (3.1415 rad, not pi) flips x and z up to sin(3.1415) ~ 9.27e-5
    cam.Camyaw = 180.0
    r2 = oracle.ORay()
    lib.oracle_primary_ray(0, 0, 8, 4, aspect, C.byref(cam), 0.5, 0.5, C.byref(r2))
    assert abs(r2.Dir.x + r.Dir.x) < 2e-4 and abs(r2.Dir.z + r.Dir.z) < 2e-4 and r2.Dir.y == r.Dir.y


def test_brightness_steps(oracle):
    # b += 0.1 is float += double (kernel.cu:1538); with no spheres all 10 samples
    # are unshadowed and b ends at the float reached by ten such steps.
    lib = oracle.load()
    b = np.float32(0)
    for _ in range(10):
        b = np.float32(float(b) + 0.1)
    start = oracle.OVec3(5, 5, 5)
    light = oracle.OLight(oracle.OVec3(0, 20, 0), 20, 0, 1, 0)
    normal = oracle.OVec3(0, 1, 0)
    got = lib.oracle_cast_light_ray(None, 0, C.byref(start), C.byref(light), C.byref(normal))
    # toL = normalise((−5,15,−5)); a = normal.toL = toL.y
    v = np.array([-5, 15, -5], dtype=np.float32)
    ln = np.sqrt(np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    ty = np.float32(v[1] / ln)
    assert abs(got - float(b * ty)) <= 2e-7


def test_shadow_rays_ignore_start_for_direction(oracle):
    # kernel.cu:1468: new_dir = normalise(l.pos - R*(x,y,z)) is relative to the
    # WORLD ORIGIN, so every sample direction stays within a few degrees of
    # l.pos/|l.pos| whatever the start point.
    lib = oracle.load()
    light = oracle.OLight(oracle.OVec3(20, 20, 20), 20, 1, 0, 0)
    u = np.array([1, 1, 1], dtype=np.float64) / math.sqrt(3)
    for start in ((1, 2, 3), (9, 0.5, 4), (5, 5, 5)):
        dirs = (C.c_float * 30)()
        st = oracle.OVec3(*start)
        lib.oracle_light_dirs(C.byref(st), C.byref(light), dirs)
        d = np.array(dirs[:], dtype=np.float64).reshape(10, 3)
        cosang = d @ u
        assert cosang.min() > math.cos(math.radians(8.0))


@pytest.mark.parametrize("libm", [False, True])
def test_c1_plumbing_frame(oracle, rt, libm):
    """BASELINE config C1: 256x256, 8 spheres, CPU scalar loop into an offscreen
    RGBA buffer (plumbing; no GPU). 99.7 % of the pixels are sky."""
    from scenes import Inputs
    inp = Inputs(rt, 8)
    rgba, packed, cnt = inp.oracle_render(oracle, 256, 256, libm=libm)
    assert cnt["primary_tests"] == 256 * 256 * 8
    assert cnt["hit_pixels"] == 168
    assert rgba.shape == (256, 256, 4) and np.isfinite(rgba).all()
    assert (rgba[..., 3] == 1).all()
    # packed word is exactly the reference's pack of the float channels
    lib = oracle.load()
    for (y, x) in ((0, 0), (128, 128), (255, 255), (200, 17)):
        assert packed[y, x] == lib.oracle_pack_color(*[float(v) for v in rgba[y, x, :3]])


# ---------------------------------------------------------------- cube / plane
def _ray(o, org, d):
    return o.ORay(o.OVec3(*org), o.OVec3(*d))


def test_plane_and_cube_layouts(oracle):
    assert C.sizeof(oracle.OPlane) == 40 and oracle.OPlane.normal.offset == 24      # kernel.cu:1214: sizeof(float)*10
    assert C.sizeof(oracle.OCube) == 80 and oracle.OCube.bounds.offset == 56


def test_plane_intersect_kats(oracle):
    lib = oracle.load()
    p = oracle.OPlane()
    lib.oracle_make_plane(C.byref(p), 0, -4, 0, 0, 1, 0)            # the reference's own plane, kernel.cu:1187
    t = C.c_float(123.0)
    # looking down: denom = -1 < 0, t = (-4)/(-1) = 4
    assert lib.oracle_plane_intersect(C.byref(p), C.byref(_ray(oracle, (0, 0, 0), (0, -1, 0))), C.byref(t)) == 1 and t.value == 4.0
    # looking up or parallel: denom >= 0 -> false and t is left untouched (kernel.cu:374-379)
    t = C.c_float(123.0)
    assert lib.oracle_plane_intersect(C.byref(p), C.byref(_ray(oracle, (0, 0, 0), (0, 1, 0))), C.byref(t)) == 0 and t.value == 123.0
    assert lib.oracle_plane_intersect(C.byref(p), C.byref(_ray(oracle, (0, 0, 0), (1, 0, 0))), C.byref(t)) == 0
    # below the plane looking down: t = 2/(-1) = -2 -> false, but t is written
    assert lib.oracle_plane_intersect(C.byref(p), C.byref(_ray(oracle, (0, -6, 0), (0, -1, 0))), C.byref(t)) == 0 and t.value == -2.0
    # the normal is used as given: a normal of length 2 halves nothing (t = dot(pl0,n)/dot(n,D) is scale free)
    lib.oracle_make_plane(C.byref(p), 0, -4, 0, 0, 2, 0)
    assert lib.oracle_plane_intersect(C.byref(p), C.byref(_ray(oracle, (0, 0, 0), (0, -1, 0))), C.byref(t)) == 1 and t.value == 4.0


def test_cube_intersect_kats(oracle):
    lib = oracle.load()
    c = oracle.OCube()
    lib.oracle_make_cube(C.byref(c), 1, 1, 1, 3, 3, 3)
    assert (c.orgin.x, c.orgin.y, c.orgin.z) == (2, 2, 2)           # divide(add(c1,c2),2), kernel.cu:395
    t = C.c_float()
    # axis-aligned ray: 1/0 = inf on two axes, slabs give (-inf, +inf) there
    assert lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (0, 2, 2), (1, 0, 0))), C.byref(t)) == 1 and t.value == 1.0
    # origin inside: tmin = -1 < 0 < tmax -> TRUE with a negative t (like the sphere's negative root)
    assert lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (2, 2, 2), (1, 0, 0))), C.byref(t)) == 1 and t.value == -1.0
    # behind: tmax = -1 < 0 -> false, t = tmax
    assert lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (4, 2, 2), (1, 0, 0))), C.byref(t)) == 0 and t.value == -1.0
    # miss beside it: tmax < tmin -> false
    assert lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (0, 5, 2), (1, 0, 0))), C.byref(t)) == 0
    # diagonal through the corner region
    d = 1 / math.sqrt(3)
    assert lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (0, 0, 0), (d, d, d))), C.byref(t)) == 1
    assert abs(t.value - math.sqrt(3)) < 1e-6
    # swapped corners behave the same (min/max per slab)
    lib.oracle_make_cube(C.byref(c), 3, 3, 3, 1, 1, 1)
    assert lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (0, 2, 2), (1, 0, 0))), C.byref(t)) == 1 and t.value == 1.0
    # on a slab boundary with a zero direction component: (bound - org) * inf = 0 * inf = NaN; the
    # reference's max/min MACROS then pick their second operand (kernel.cu:16-26)
    lib.oracle_make_cube(C.byref(c), 1, 1, 1, 3, 3, 3)
    r = lib.oracle_cube_intersect(C.byref(c), C.byref(_ray(oracle, (0, 1, 2), (1, 0, 0))), C.byref(t))
    t3, t4 = float("nan"), float("inf")        # (1-1)*inf, (3-1)*inf
    mn = t3 if t3 < t4 else t4                 # MIN(t3,t4) -> t4 = inf
    assert mn == float("inf") and r == 0       # tmin = inf > tmax -> false
