"""Independent pins for the quirk-laden rows of SURVEY.md 8(a): a7 / a8 (castLightRay's sample
construction with the non-standard rotate(), kernel.cu:1262-1280, 1437-1468) and one whole pixel
of every kind (lit, fully shadowed, penumbra, sky) through rgbToInt (kernel.cu:1614-1690).

The reference holds no fixtures and cannot be built here, so the only pin it allows is a SECOND,
independent restatement of its source text: the functions below are written in numpy binary32 /
binary64 scalars INSIDE this test file, from /root/reference/kernel.cu, and share no code with
oracle/ (C) or with the device (HIP). Each numpy operation is the IEEE operation the C++
expression performs (no FMA; binary64 where a bare literal or a `double` forces it, narrowed where
the reference assigns to a float); the transcendentals are numpy's binary64 functions rounded once
to binary32, i.e. the correctly rounded float, which is what both the oracle and the device define
theirs to be (up to ~1e-9 of inputs). Bit-equality is required against the oracle (CPU suite) and
against the device (gpu-marked test)."""
import ctypes as C
import math

import numpy as np
import pytest

f32 = np.float32
f64 = np.float64


# ----------------------------------------------------------------------------- kernel.cu:38-128
def _dot(a, b):                      # dotproduct(vec1, vec2), kernel.cu:95-98: left to right
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _sub(a, b):
    return [f32(a[0] - b[0]), f32(a[1] - b[1]), f32(a[2] - b[2])]


def _add(a, b):
    return [f32(a[0] + b[0]), f32(a[1] + b[1]), f32(a[2] + b[2])]


def _mulf(a, k):                     # multiply(vec3d, float), kernel.cu:79-82
    k = f32(k)
    return [f32(a[0] * k), f32(a[1] * k), f32(a[2] * k)]


def _cross(a, b):                    # kernel.cu:84-89
    return [f32(f32(a[1] * b[2]) - f32(a[2] * b[1])),
            f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
            f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))]


def _normalise(v):
    """normalise(vec3d&), kernel.cu:103-108: `double l = length(v)`; `v.x /= l` divides in binary64 and
    narrows, IN PLACE; returns the updated components, or (0,0,0) leaving v alone when l == 0."""
    with np.errstate(all="ignore"):
        l = f64(np.sqrt(_dot(v, v)))              # sqrtf of a float, widened
        if l != 0:
            for i in range(3):
                v[i] = f32(f64(v[i]) / l)
            return list(v)
    return [f32(0), f32(0), f32(0)]


def _cosf(x):
    with np.errstate(all="ignore"):
        return f32(np.cos(f64(x)))


def _sinf(x):
    with np.errstate(all="ignore"):
        return f32(np.sin(f64(x)))


def _acosf(x):
    with np.errstate(all="ignore"):
        return f32(np.arccos(f64(x)))


def _atan2f(y, x):
    with np.errstate(all="ignore"):
        return f32(np.arctan2(f64(y), f64(x)))


def _sqrtf(x):
    with np.errstate(all="ignore"):
        return f32(np.sqrt(f32(x)))


# ----------------------------------------------------------------------------- kernel.cu:1262-1280
def _rotate_apply(angle, ax, v):
    """multiply(rotate(angle, axis), v): the matrix of kernel.cu:1267-1277 (m00 = cos + x*x WITHOUT a
    (1 - cos) factor; the sign pattern is not Rodrigues') applied as kernel.cu:123-125 does."""
    c, s = _cosf(angle), _sinf(angle)
    omc = f32(f32(1) - c)
    x, y, z = ax
    m00 = f32(c + f32(x * x))
    m01 = f32(f32(f32(x * y) * omc) - f32(z * s))
    m02 = f32(f32(f32(x * z) * omc) - f32(y * s))
    m10 = f32(f32(f32(y * x) * omc) + f32(z * s))
    m11 = f32(c + f32(f32(y * y) * omc))
    m12 = f32(f32(f32(y * z) * omc) - f32(x * s))
    m20 = f32(f32(f32(z * x) * omc) - f32(y * s))
    m21 = f32(f32(f32(z * y) * omc) + f32(x * s))
    m22 = f32(c + f32(f32(z * z) * omc))
    return [f32(f32(f32(v[0] * m00) + f32(v[1] * m10)) + f32(v[2] * m20)),
            f32(f32(f32(v[0] * m01) + f32(v[1] * m11)) + f32(v[2] * m21)),
            f32(f32(f32(v[0] * m02) + f32(v[1] * m12)) + f32(v[2] * m22))]


# ----------------------------------------------------------------------------- kernel.cu:293-354
def _intersect(c, radius_field, org, d):
    """sphere::intersect. Returns (hit, t). No discriminant test: a NaN t fails every comparison."""
    with np.errstate(all="ignore"):
        ocx, ocy, ocz = f32(org[0] - c[0]), f32(org[1] - c[1]), f32(org[2] - c[2])
        A = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
        B = f32(f32(2) * f32(f32(f32(d[0] * ocx) + f32(d[1] * ocy)) + f32(d[2] * ocz)))
        Cq = f32(f32(f32(f32(ocx * ocx) + f32(ocy * ocy)) + f32(ocz * ocz)) - f32(radius_field * radius_field))
        disc = f32(f32(B * B) - f32(f32(f32(4) * A) * Cq))
        sq = _sqrtf(disc)
        twoA = f32(f32(2) * A)
        t = f32(f32(f32(-B) + sq) / twoA)
        if t == f32(0):
            return True, t
        if f64(t) >= 0.0001:
            t2 = f32(f32(f32(-B) - sq) / twoA)
            if t > t2:
                t = t2
            return True, t
        return False, t


# ----------------------------------------------------------------------------- kernel.cu:1433-1544
def light_dirs(lpos, lsize, start):
    """The ten sample directions of castLightRay and toL as it is left (kernel.cu:1438-1468)."""
    lpos = [f32(v) for v in lpos]
    start = [f32(v) for v in start]
    toL = _sub(lpos, start)
    _normalise(toL)
    dirs = []
    with np.errstate(all="ignore"):
        for j in range(10):
            P = _cross(toL, [f32(0), f32(1), f32(0)])
            e = _sub(_add(lpos, _mulf(P, lsize)), start)
            toEdge = _normalise(e)
            angle = _cosf(f32(_dot(toL, toEdge) * f32(2)))
            jf = f32(f32(j) / f32(10))
            z = f32(f32(jf * f32(f32(1) - angle)) + angle)
            phi = f32(f32(jf * f32(2)) * f32(3.1415))
            sq = _sqrtf(f32(f32(1) - f32(z * z)))
            x = f32(sq * _cosf(phi))
            sq2 = _sqrtf(f32(f32(1) - f32(z * z)))
            y = f32(sq2 * _sinf(phi))
            n1 = _normalise(toL)                                   # in place
            axis = _normalise(_cross([f32(0), f32(0), f32(1)], n1))
            n2 = _normalise(toL)                                   # in place, again
            nAngle = _acosf(_dot(n2, [f32(0), f32(0), f32(1)]))
            nd = _sub(lpos, _rotate_apply(nAngle, axis, [x, y, z]))
            dirs.append(_normalise(nd))
    return dirs, toL


def cast_light_ray(spheres, lpos, lsize, start, normal):
    dirs, toL = light_dirs(lpos, lsize, start)
    b = f32(0)
    unshadowed = 0
    for d in dirs:
        shadow = False
        for (c, rf) in spheres:
            hit, _ = _intersect(c, rf, start, d)
            if hit:
                shadow = True
                break
        if not shadow:
            b = f32(f64(b) + 0.1)                                  # kernel.cu:1538: float += double literal
            unshadowed += 1
    a = _dot(normal, toL)
    b = f32(b * (a if a > 0 else f32(0)))
    return b, unshadowed


# ----------------------------------------------------------------------------- kernel.cu:546-556, 1614-1690
def _f2i(v):                         # float -> int as CUDA's cvt.rzi: truncation, NaN -> 0, saturating
    v = float(v)
    if math.isnan(v):
        return 0
    return int(max(min(math.trunc(v), 2**31 - 1), -2**31))


def _rgb_to_int(r, g, b):
    r, g, b = min(r, 255), min(g, 255), min(b, 255)
    return ((r & 0xff) << 16) + ((g & 0xff) << 8) + (b & 0xff)


def trace_pixel(px, py, W, H, aspect, cam_org, yaw, pitch, spheres, lights, tex, sky, sky_field):
    """rayTrace for one pixel. Returns (packed word, (fr, fg, fb), kind, unshadowed counts)."""
    aspect = f32(aspect)
    with np.errstate(all="ignore"):
        dx = f32(f64(aspect) * (f64(2) * (f64(px) + 0.5) / f64(f32(W))) - f64(1))
        dy = f32(f64(aspect) * (f64(2) * (f64(py) + 0.5) / f64(f32(H))) * f64(f32(f32(H) / f32(W))) - f64(1))
        eye = [f32(0), f32(0), f32(f32(-1) / aspect)]
        org = _add(eye, [f32(v) for v in cam_org])
        v = _sub([dx, dy, f32(0)], eye)
        _normalise(v)
        yr = f32(f64(f32(yaw)) * (3.1415 / 180))
        pr = f32(f64(f32(pitch)) * (3.1415 / 180))
        y = f32(f32(v[1] * _cosf(pr)) - f32(v[2] * _sinf(pr)))
        z = f32(f32(v[1] * _sinf(pr)) + f32(v[2] * _cosf(pr)))
        x = f32(f32(v[0] * _cosf(yr)) + f32(z * _sinf(yr)))
        z = f32(f32(f32(-v[0]) * _sinf(yr)) + f32(z * _cosf(yr)))
        D = [x, y, z]
        nt, hi = f32(np.inf), -1
        for i, (c, rf) in enumerate(spheres):
            hit, t = _intersect(c, rf, org, D)
            if hit and t < nt:
                nt, hi = t, i
        if hi >= 0:
            new_org = _add(org, _mulf(D, nt))
            normal = _sub(new_org, spheres[hi][0])
            _normalise(normal)
            tx = f32((f64(1) + f64(_atan2f(normal[2], normal[0])) / 3.1415) * 0.5)
            ty = f32(f64(_acosf(normal[1])) / 3.1415)
            th, tw = tex[0].shape
            ci = _f2i(f32(ty * f32(th))) * tw + _f2i(f32(tx * f32(tw)))
            ci = max(0, min(ci, tw * th - 1))                       # the build's documented clamp (reference: UB)
            r, g, b = (f32(p.reshape(-1)[ci]) for p in tex)
            start = _add(_mulf(normal, f32(0.00001)), new_org)
            fr = fg = fb = f32(0)
            counts = []
            for (lpos, lsize, lr, lg, lb) in lights:
                br, un = cast_light_ray(spheres, lpos, lsize, start, normal)
                counts.append(un)
                fr = f32(fr + f32(f32(br * f32(lr)) * r))
                fg = f32(fg + f32(f32(br * f32(lg)) * g))
                fb = f32(fb + f32(f32(br * f32(lb)) * b))
            word = _rgb_to_int(_f2i(f32(fr * f32(254))), _f2i(f32(fg * f32(254))), _f2i(f32(fb * f32(254))))
            return word, (fr, fg, fb), "hit", counts
        # skybox::getFColor, kernel.cu:1147-1166: the box is hit at its NEGATIVE near root
        _, t = _intersect([f32(0), f32(0), f32(0)], f32(sky_field), org, D)
        hp = _add(org, _mulf(D, t))
        n = _sub(hp, [f32(0), f32(0), f32(0)])
        _normalise(n)
        sh, sw = sky[0].shape
        ix = _f2i(f32(f32(f32(f32(1) + f32(_atan2f(n[2], n[0]) / f32(3.1415))) * f32(0.5)) * f32(sw)))
        iy = _f2i(f32(f32(_acosf(n[1]) / f32(3.1415)) * f32(sh)))
        idx = max(0, min(iy * sw + ix, sw * sh - 1))
        r, g, b = (f32(p.reshape(-1)[idx]) for p in sky)
        word = _rgb_to_int(_f2i(f32(r * f32(254))), _f2i(f32(g * f32(254))), _f2i(f32(b * f32(254))))
        return word, (r, g, b), "sky", []


# ----------------------------------------------------------------------------- the pins
REFERENCE_LIGHTS = [((20, 20, 20), 20.0), ((0, 20, -20), 20.0), ((0, 20, 0), 20.0)]      # kernel.cu:1708-1710
STARTS = [(4.1, 6.7, 3.4), (1.0, 2.0, 3.0), (9.0, 0.5, 4.0), (5.0, 5.0, 5.0), (0.25, 9.75, 7.5), (6.9, 2.4, 7.8), (-3.0, 1.0, 12.0)]


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_sample_directions_match_the_oracle_bit_for_bit(oracle):
    """kernel.cu:1437-1468 restated here vs oracle_light_dirs: 3 reference lights x 7 starts x 10 directions."""
    lib = oracle.load()
    for (lpos, lsize) in REFERENCE_LIGHTS:
        for st in STARTS:
            want, _ = light_dirs(lpos, lsize, st)
            dirs = (C.c_float * 30)()
            s = oracle.OVec3(*st)
            light = oracle.OLight(oracle.OVec3(*lpos), lsize, 1, 1, 1)
            lib.oracle_light_dirs(C.byref(s), C.byref(light), dirs)
            got = np.array(dirs[:], dtype=np.float32).reshape(10, 3)
            assert np.array_equal(_bits(got), _bits(np.array(want, dtype=np.float32))), (lpos, st)


def test_rotate_is_the_references_matrix_not_rodrigues():
    """Hand-checkable: angle = pi/2 about (0, 1, 0). kernel.cu:1267-1277 gives m00 = cos + x*x = cos,
    m02 = -y*sin = -1, m20 = -y*sin = -1 (Rodrigues: m02 = +1 / m20 = -1 up to convention, never equal), m11 = cos + y*y*(1-cos)
    = 1: (1,0,0) -> (cos, 0, -1) and (0,0,1) -> (-1, 0, cos) -- the matrix is symmetric here, which no rotation by pi/2 is."""
    a = f32(np.pi / 2)
    c = _cosf(a)
    v1 = _rotate_apply(a, [f32(0), f32(1), f32(0)], [f32(1), f32(0), f32(0)])
    v2 = _rotate_apply(a, [f32(0), f32(1), f32(0)], [f32(0), f32(0), f32(1)])
    assert (v1[0], v1[1], v1[2]) == (c, f32(0), f32(-1))
    assert (v2[0], v2[1], v2[2]) == (f32(-1), f32(0), c)


def _kat_scene(rt):
    """Receiver sphere under the reference's green light (0,20,0), a smaller occluder straight above it
    (shadow rays leave along ~l.pos/|l.pos| = +y whatever the start, kernel.cu:1468), sky around."""
    lib = rt.load_library()
    sph = (rt.Sphere * 2)()
    lib.rt_sphere_init(C.byref(sph[0]), 4.0, 0.5, 5.0, 1.2)       # field 1.44: effective radius 1.44
    lib.rt_sphere_init(C.byref(sph[1]), 4.0, 4.0, 5.0, 0.8)       # field 0.64
    lights = (rt.Light * 1)()
    lights[0] = rt.Light(rt.Vec3(0, 20, 0), 20, 0.25, 1.0, 0.5)
    return sph, lights


def _restated_frame(rt, W, H):
    from scenes import Inputs
    inp = Inputs(rt, 0)
    sph, lights = _kat_scene(rt)
    spheres = [([f32(s.orgin.x), f32(s.orgin.y), f32(s.orgin.z)], f32(s.radius)) for s in sph]
    lts = [((l.pos.x, l.pos.y, l.pos.z), l.size, l.r, l.g, l.b) for l in lights]
    cam = inp.cam
    words = np.zeros((H, W), dtype=np.uint32)
    rgb = np.zeros((H, W, 3), dtype=np.float32)
    kinds = {}
    for y in range(H):
        for x in range(W):
            w, c, kind, counts = trace_pixel(x, y, W, H, inp.aspect, (cam.Org.x, cam.Org.y, cam.Org.z), cam.Camyaw, cam.Campitch,
                                             spheres, lts, inp.tex, inp.sky, inp.sky_box.radius)
            words[y, x] = w
            rgb[y, x] = c
            key = kind if kind == "sky" else ("lit" if counts[0] == 10 else "dark" if counts[0] == 0 else "penumbra")
            if key == "dark" and not (c[1] == 0):
                key = "penumbra"
            kinds.setdefault(key, []).append((y, x, float(c[1])))
    return inp, sph, lights, words, rgb, kinds


def test_whole_pixels_match_the_oracle_bit_for_bit(oracle, rt):
    """A 40x30 frame of the two-sphere / one-light scene, EVERY pixel derived by the restatement above through
    rgbToInt, against the oracle's frame: float channels and packed words, bit for bit. The frame must contain
    sky pixels, fully lit pixels (10 unshadowed samples, brightness > 0), fully shadowed pixels lying under the
    occluder on the lit side (0 unshadowed, facing the light) and penumbra pixels."""
    W, H = 40, 30
    inp, sph, lights, words, rgb, kinds = _restated_frame(rt, W, H)
    rgba, packed, cnt = oracle.render(sph, 2, inp.tex, inp.sky, inp.sky_box, lights, 1, inp.cam, W, H, inp.aspect, nthreads=4)
    assert {"sky", "lit", "dark", "penumbra"} <= set(kinds), {k: len(v) for k, v in kinds.items()}
    assert any(g > 0.05 for (_, _, g) in kinds["lit"])
    assert np.array_equal(_bits(rgba[..., :3]), _bits(rgb))
    assert np.array_equal(packed, words)
    assert cnt["hit_pixels"] == sum(len(v) for k, v in kinds.items() if k != "sky")


@pytest.mark.gpu
def test_device_matches_the_independent_restatement(rt, gpu):
    """The same pins against the DEVICE: the sample directions through rt_debug_light, the whole frame through
    the frame kernel (culling and brute-force instantiations)."""
    import torch
    lib = rt.load_library()
    n = len(STARTS)
    starts = (rt.Vec3 * n)(*[rt.Vec3(*s) for s in STARTS])
    normals = (rt.Vec3 * n)(*[rt.Vec3(0, 1, 0) for _ in STARTS])
    none = (rt.Sphere * 1)()
    for (lpos, lsize) in REFERENCE_LIGHTS:
        light = rt.Light(rt.Vec3(*lpos), lsize, 1, 1, 1)
        dirs = np.zeros((n, 10, 3), dtype=np.float32)
        bright = np.zeros(n, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        assert lib.rt_debug_light(none, 0, starts, normals, C.byref(light), n, dirs.ctypes.data_as(fp), bright.ctypes.data_as(fp)) == 0, \
            lib.rt_last_error()
        for i, st in enumerate(STARTS):
            want, toL = light_dirs(lpos, lsize, st)
            assert np.array_equal(_bits(dirs[i]), _bits(np.array(want, dtype=np.float32))), (lpos, st)
            b = f32(0)
            for _ in range(10):
                b = f32(f64(b) + 0.1)
            a = _dot([f32(0), f32(1), f32(0)], toL)
            assert _bits(bright[i]) == _bits(f32(b * (a if a > 0 else f32(0)))), (lpos, st)
    W, H = 40, 30
    inp, sph, lights, words, rgb, kinds = _restated_frame(rt, W, H)
    sc = rt.Scene()
    sc.set_spheres(sph, 2)
    sc.set_texture(inp.tex)
    sc.set_sky(inp.sky_box, inp.sky)
    sc.set_lights(lights, 1)
    for cull in (True, False):
        out = sc.render(W, H, cull=cull)
        torch.cuda.synchronize()
        assert np.array_equal(out["packed"].cpu().numpy().view(np.uint32), words), cull
        assert np.array_equal(_bits(out["rgba"].cpu().numpy()[..., :3]), _bits(rgb)), cull
