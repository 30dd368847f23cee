"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle
and the committed golden fixtures. Everything here is bit-exact: the oracle's
transcendentals and the device's are the same binary64 algorithm rounded once,
and every deciding operation is an IEEE binary32/64 operation on both sides.

Tolerance note (north_star: 1e-5 relative per channel): the tests below demand
0 ulp. tests/test_oracle_math.py::test_portable_matches_libm_flavour bounds what
swapping in another libm does (<=1e-5 relative on non-flipped pixels)."""
import ctypes as C
import os

import numpy as np
import pytest

from scenes import GOLDEN_CASES, GOLDEN_SPP_CASES, Inputs, mixed_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _render(scene, w, h, **kw):
    out = scene.render(w, h, **kw)
    import torch
    torch.cuda.synchronize()
    rgba = out["rgba"].cpu().numpy() if out["rgba"] is not None else None
    packed = out["packed"].cpu().numpy().view(np.uint32)
    return rgba, packed, out.get("stats")


# ---------------------------------------------------------------- building blocks
def test_device_math_is_bit_identical_to_oracle(rt, oracle, gpu):
    lib, ol = rt.load_library(), oracle.load()
    rng = np.random.default_rng(1)
    fp = C.POINTER(C.c_float)
    cases = {
        0: np.concatenate([rng.uniform(-7, 7, 60000), np.linspace(-50, 50, 4001), [0, 3.1415, 1e-8, 6.2830]]),
        1: np.concatenate([rng.uniform(-7, 7, 60000), np.linspace(-50, 50, 4001), [0, 3.1415, 1e-8, 6.2830]]),
        2: np.concatenate([rng.uniform(-1, 1, 60000), 1 - rng.uniform(0, 1e-5, 2000), [-1, 1, 0, 1.0000001, -1.0000001]]),
    }
    names = {0: "oracle_cosf", 1: "oracle_sinf", 2: "oracle_acosf"}
    for op, xs in cases.items():
        xs = xs.astype(np.float32)
        out = np.empty_like(xs)
        assert lib.rt_debug_math(op, xs.ctypes.data_as(fp), None, out.ctypes.data_as(fp), len(xs)) == 0
        want = np.array([getattr(ol, names[op])(float(x)) for x in xs], dtype=np.float32)
        assert np.array_equal(_bits(out), _bits(want)), names[op]
    inf = float("inf")
    ys = np.concatenate([rng.uniform(-1, 1, 60000), [0, 0, -0.0, 1, -1, 0, inf, -inf, inf, -inf, 1, inf, -inf, 2]]).astype(np.float32)
    xs = np.concatenate([rng.uniform(-1, 1, 60000), [1, -1, -1, 0, 0, 0, inf, inf, -inf, -inf, inf, 1, -3, -inf]]).astype(np.float32)
    out = np.empty_like(xs)
    assert lib.rt_debug_math(3, ys.ctypes.data_as(fp), xs.ctypes.data_as(fp), out.ctypes.data_as(fp), len(xs)) == 0
    want = np.array([ol.oracle_atan2f(float(y), float(x)) for y, x in zip(ys, xs)], dtype=np.float32)
    assert np.array_equal(_bits(out), _bits(want))
    # "/ 3.1415" of kernel.cu:1402-1403 is a multiply + residual correction on the device
    # (rtm::div_by_3p1415); the binary64 quotient is the reference (tests/test_const_div.py
    # checks the sequence over all 2^32 floats on the CPU)
    xs = np.concatenate([rng.uniform(-3.2, 3.2, 200000), rng.uniform(-1e-30, 1e-30, 1000),
                         [0.0, 3.1415, -3.1415, 3.1415927, 1e-45, -1e-45, 1.5707964]]).astype(np.float32)
    out = np.empty_like(xs)
    assert lib.rt_debug_math(4, xs.ctypes.data_as(fp), None, out.ctypes.data_as(fp), len(xs)) == 0
    assert np.array_equal(_bits(out), _bits(((1.0 + xs.astype(np.float64) / 3.1415) * 0.5).astype(np.float32)))
    assert lib.rt_debug_math(5, xs.ctypes.data_as(fp), None, out.ctypes.data_as(fp), len(xs)) == 0
    assert np.array_equal(_bits(out), _bits((xs.astype(np.float64) / 3.1415).astype(np.float32)))


def test_device_intersect_matches_oracle(rt, oracle, gpu):
    lib, ol = rt.load_library(), oracle.load()
    rng = np.random.default_rng(2)
    n = 20000
    sph = (rt.Sphere * n)()
    rays = (rt.Ray * n)()
    kat = [((0, 0, 5, 1), (0, 0, 0), (0, 0, 1)), ((0, 0, 5, 2), (0, 0, 0), (0, 0, 1)),
           ((0, 5, 5, 1), (0, 0, 0), (0, 0, 1)), ((0, 0, 0, 1), (0, 0, 0), (0, 0, 1)),
           ((0, 0, -5, 1), (0, 0, 0), (0, 0, 1)), ((0, 0, 0, 10000), (4, 3, 9), (0, 0, -1))]
    for i in range(n):
        if i < len(kat):
            (cx, cy, cz, r), o, d = kat[i]
        else:
            cx, cy, cz = rng.uniform(0, 10, 3)
            r = rng.uniform(0, 1)
            o = rng.uniform(-2, 12, 3)
            d = rng.normal(size=3)
            if i % 3 == 0:                      # aim near the sphere: grazing cases
                d = np.array([cx, cy, cz]) - o + rng.normal(size=3) * r * r
            if i % 7 == 0:                      # origin inside / on the surface
                o = np.array([cx, cy, cz]) + rng.normal(size=3) * r * r * 0.6
            d = d / np.linalg.norm(d)
        lib.rt_sphere_init(C.byref(sph[i]), cx, cy, cz, r)
        rays[i] = rt.Ray(rt.Vec3(*[float(v) for v in o]), rt.Vec3(*[float(v) for v in d]))
    hit = (C.c_int * n)()
    t = (C.c_float * n)()
    assert lib.rt_debug_intersect(sph, rays, n, hit, t) == 0
    for i in range(n):
        os_, or_ = oracle.OSphere(), oracle.ORay()
        C.memmove(C.byref(os_), C.byref(sph[i]), 32)
        C.memmove(C.byref(or_), C.byref(rays[i]), 24)
        tt = C.c_float()
        h = ol.oracle_sphere_intersect(C.byref(os_), C.byref(or_), C.byref(tt))
        assert h == hit[i], i
        assert _bits(np.float32(tt.value)) == _bits(np.float32(t[i])), i
    assert [hit[i] for i in range(6)] == [1, 1, 0, 1, 0, 1]
    assert (t[0], t[1], t[3], t[4]) == (4.0, 1.0, -1.0, -4.0)


@pytest.mark.parametrize("light_index", [0, 1, 2])
def test_device_light_sampling_matches_oracle(rt, oracle, gpu, light_index):
    lib, ol = rt.load_library(), oracle.load()
    rng = np.random.default_rng(3 + light_index)
    n, ns = 3000, 64
    inp = Inputs(rt, ns)
    starts = (rt.Vec3 * n)()
    normals = (rt.Vec3 * n)()
    for i in range(n):
        p = rng.uniform(0, 10, 3)
        nv = rng.normal(size=3)
        nv /= np.linalg.norm(nv)
        starts[i] = rt.Vec3(*[float(v) for v in p])
        normals[i] = rt.Vec3(*[float(v) for v in nv])
    light = inp.lights[light_index]
    dirs = np.empty((n, 30), dtype=np.float32)
    bright = np.empty(n, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    assert lib.rt_debug_light(inp.spheres, ns, starts, normals, C.byref(light), n, dirs.ctypes.data_as(fp),
                              bright.ctypes.data_as(fp)) == 0
    osph = (oracle.OSphere * ns)()
    C.memmove(osph, inp.spheres, 32 * ns)
    ol_light = oracle.OLight()
    C.memmove(C.byref(ol_light), C.byref(light), 28)
    for i in range(n):
        st, nm = oracle.OVec3(), oracle.OVec3()
        C.memmove(C.byref(st), C.byref(starts[i]), 12)
        C.memmove(C.byref(nm), C.byref(normals[i]), 12)
        want = (C.c_float * 30)()
        ol.oracle_light_dirs(C.byref(st), C.byref(ol_light), want)
        assert np.array_equal(_bits(dirs[i]), _bits(np.array(want[:], dtype=np.float32))), i
        b = ol.oracle_cast_light_ray(osph, ns, C.byref(st), C.byref(ol_light), C.byref(nm))
        assert _bits(np.float32(b)) == _bits(bright[i]), i


def test_shortcuts_equal_the_long_forms(rt, gpu):
    """The culling kernels' cheaper forms (rt_kernels.hip: lean normalise / sqrt, fast texel index)
    against what they stand for, on the device: the lean square root on EVERY float of its range,
    the lean normalise on 2^28 vectors of every scale (zero, denormal, huge components included:
    those must take the IEEE path and still agree), the approximate (tx, ty) on 2^28 unit normals
    (poles and seams over-sampled): error below half of RT_UV_DELTA and not one accepted lane with
    a texel index different from the exact binary64 expressions'."""
    lib = rt.load_library()
    out = (C.c_ulonglong * 4)()
    assert lib.rt_debug_shortcuts(1, 0, 0, out) == 0, lib.rt_last_error()
    assert out[0] == 0, f"lean sqrt differs from IEEE sqrtf on {out[0]} inputs"
    for seed in (1, 2):
        assert lib.rt_debug_shortcuts(0, seed, 1 << 27, out) == 0
        assert out[0] == 0, f"lean normalise differs from the IEEE one on {out[0]} vectors"
        assert lib.rt_debug_shortcuts(2, seed, 1 << 27, out) == 0
        ex = np.array([out[0]], dtype=np.uint32).view(np.float32)[0]
        ey = np.array([out[1]], dtype=np.uint32).view(np.float32)[0]
        assert ex < 2.5e-7 and ey < 2.5e-7, (ex, ey)
        assert out[3] == 0, f"{out[3]} accepted lanes select another texel than the exact expressions"
        assert out[2] > 0.995 * (1 << 27) * 0.8, out[2]      # the over-sampled seams and poles are rejected more often


def test_prepass_directions_stay_within_their_error_bound(rt, gpu):
    """The sample pre-pass of the frame kernel (rt_trace.inc: setup_approx / direction_approx / presure_test)
    classifies shadow rays with APPROXIMATE directions and relies on |approximate - exact| < RT_PRE_DELTA = 1e-5
    wherever its guards let a direction through. Measured here on the device for 200 000 starts x 10 samples per
    light: starts in and around the scene for the three reference lights, and for random lights -- close ones, ones
    along the axes (toL near +-z is where the frame's rotation axis is ill-conditioned: the guard must cut in) and
    far ones. The largest deviation seen must stay below a third of the bound."""
    lib = rt.load_library()
    rng = np.random.default_rng(5)
    n = 200000
    fp = C.POINTER(C.c_float)
    cases = [((20, 20, 20), 20.0), ((0, 20, -20), 20.0), ((0, 20, 0), 20.0), ((3, 4, 40), 5.0), ((0.5, -0.25, -30), 30.0),
             ((12, 9, 11), 2.0), ((-40, 3, 5), 0.5), ((5, 5, 30), 20.0)]
    worst = 0.0
    used = 0
    for (lpos, size) in cases:
        pts = rng.uniform(-2.0, 12.0, size=(n, 3)).astype(np.float32)
        starts = (rt.Vec3 * n).from_buffer_copy(pts.tobytes())
        light = rt.Light(rt.Vec3(*lpos), size, 1, 1, 1)
        d = np.zeros((n, 10, 3), dtype=np.float32)
        a = np.zeros((n, 10, 3), dtype=np.float32)
        ok = np.zeros((n, 10), dtype=np.int32)
        assert lib.rt_debug_light_prepass(starts, C.byref(light), n, d.ctypes.data_as(fp), a.ctypes.data_as(fp),
                                          ok.ctypes.data_as(C.POINTER(C.c_int))) == 0, lib.rt_last_error()
        dev = np.linalg.norm(a.astype(np.float64) - d.astype(np.float64), axis=2)
        dev = np.where(np.isfinite(dev), dev, np.inf)          # an exact NaN direction with the guard open would be a miss
        m = ok.astype(bool)
        used += int(m.sum())
        if m.any():
            worst = max(worst, float(dev[m].max()))
    assert used > 0.5 * len(cases) * n * 10                    # the guards let most samples through
    assert worst < 1.0e-5 / 3, worst


# ---------------------------------------------------------------- frames vs golden
@pytest.mark.parametrize("cull", [True, False])
@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_frame_matches_golden(name, cull, rt, gpu):
    w, h, n, y0, y1 = GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    scene = Inputs(rt, n).scene()
    rgba, packed, stats = _render(scene, w, h, y0=y0, y1=y1, cull=cull, want_stats=True)
    assert np.array_equal(_bits(rgba[..., :3]), _bits(g["rgb"])), "float channels differ from the oracle"
    assert (rgba[..., 3] == 1).all()
    assert np.array_equal(packed, g["packed"])
    cnt = g["counters"].tolist()
    assert stats["hit_pixels"] == cnt[2]
    if not cull:
        assert stats["primary_tests"] == cnt[0]          # brute force issues exactly the reference's primary tests
        assert stats["unshadowed"] == cnt[3]
        assert stats["cull_tests"] == 0
    else:
        assert stats["primary_tests"] <= cnt[0]
        assert stats["unshadowed"] <= cnt[3]             # lights a surface faces away from are skipped


@pytest.mark.parametrize("cull", [True, False])
def test_mixed_primitives_match_golden(cull, rt, gpu):
    g = np.load(os.path.join(GOLD, "mixed_160x96.npz"))
    inp = mixed_scene(rt)
    scene = inp.scene()
    scene.set_planes(inp.planes, inp.n_planes)
    scene.set_cubes(inp.cubes, inp.n_cubes)
    rgba, packed, stats = _render(scene, 160, 96, cull=cull, want_stats=True)
    assert np.array_equal(_bits(rgba[..., :3]), _bits(g["rgb"])) and np.array_equal(packed, g["packed"])
    assert stats["hit_pixels"] == int(g["counters"][2])


@pytest.mark.parametrize("tile", [8, 16, 32, 64])
def test_tile_shape_and_slow_path_invariance(tile, rt, gpu):
    w, h, n = 160, 90, 1024
    g = np.load(os.path.join(GOLD, "c3_160x90_n1024.npz"))
    scene = Inputs(rt, n).scene()
    for kw in (dict(cull=True), dict(cull=True, force_slow=True), dict(cull=False, force_slow=True)):
        rgba, packed, _ = _render(scene, w, h, tile=tile, **kw)
        assert np.array_equal(_bits(rgba[..., :3]), _bits(g["rgb"])), (tile, kw)
        assert np.array_equal(packed, g["packed"])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_bands_reassemble_the_frame(world, rt, gpu):
    w, h, n = 160, 90, 256
    g = np.load(os.path.join(GOLD, "c2_160x90_n256.npz"))
    scene = Inputs(rt, n).scene()
    parts_rgb, parts_pk = [], []
    for r in range(world):
        y0, y1 = rt.band_rows(h, r, world)
        rgba, packed, _ = _render(scene, w, h, y0=y0, y1=y1)
        assert rgba.shape[0] == y1 - y0
        parts_rgb.append(rgba)
        parts_pk.append(packed)
    assert np.array_equal(_bits(np.concatenate(parts_rgb)[..., :3]), _bits(g["rgb"]))
    assert np.array_equal(np.concatenate(parts_pk), g["packed"])


@pytest.mark.parametrize("world,tile", [(2, 8), (3, 16), (8, 64)])
def test_interleaved_row_blocks_reassemble_the_frame(world, tile, rt, gpu):
    """Balanced multi-GPU split: 16-row blocks dealt round-robin, compact local
    buffers; scattering the rows back gives the single-GPU frame bit for bit."""
    w, h, n = 160, 90, 1024
    g = np.load(os.path.join(GOLD, "c3_160x90_n1024.npz"))
    scene = Inputs(rt, n).scene()
    rgb = np.zeros((h, w, 3), dtype=np.float32)
    pk = np.zeros((h, w), dtype=np.uint32)
    seen = np.zeros(h, dtype=np.int32)
    for r in range(world):
        rows = rt.interleaved_rows(h, r, world, 16)
        rgba, packed, _ = _render(scene, w, h, interleave=(world, r, 16), tile=tile)
        assert rgba.shape[0] == len(rows)
        rgb[rows] = rgba[..., :3]
        pk[rows] = packed
        seen[rows] += 1
    assert (seen == 1).all()
    assert np.array_equal(_bits(rgb), _bits(g["rgb"])) and np.array_equal(pk, g["packed"])


def test_spp4_in_kernel_and_progressive(rt, gpu):
    import torch
    w, h, n = 96, 54, 256
    g = np.load(os.path.join(GOLD, "spp4_96x54_n256.npz"))
    scene = Inputs(rt, n).scene()
    rgba, packed, _ = _render(scene, w, h, spp=4)                  # four samples inside one launch
    assert np.array_equal(_bits(rgba), _bits(g["acc"]))
    assert np.array_equal(packed, g["packed"])
    # progressive: one sample per launch, accumulated in the float4 buffer
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    pk = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    for k in range(4):
        fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), rgba=acc.data_ptr(), spp=1, sample_base=k,
                              sample_total=4, accumulate=k > 0, resolve=0 if k == 3 else -1)
        scene.render_raw(fd, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(acc.cpu().numpy()), _bits(g["acc"]))
    assert np.array_equal(pk.cpu().numpy().view(np.uint32), g["packed"])
    # 1 spp through the sample machinery is the reference's pixel centre
    rgba1, packed1, _ = _render(scene, w, h, spp=1)
    ref, refp, _ = Inputs(rt, n).oracle_render(__import__("oracle_py"), w, h)
    assert np.array_equal(_bits(rgba1), _bits(ref)) and np.array_equal(packed1, refp)


@pytest.mark.parametrize("passes", [4, -4])   # four samples in one kernel node / four progressive one-sample nodes
def test_graph_replay_equals_direct_launch(rt, gpu, passes):
    import torch
    lib = rt.load_library()
    w, h, n = 96, 54, 256
    g = np.load(os.path.join(GOLD, "spp4_96x54_n256.npz"))
    scene = Inputs(rt, n).scene()
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    pk = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    host = torch.zeros((h, w), dtype=torch.int32).pin_memory()
    stream = torch.cuda.Stream()
    fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), rgba=acc.data_ptr())
    gr = lib.rt_graph_capture(scene.handle, C.byref(fd), passes, host.data_ptr(), stream.cuda_stream)
    assert gr, lib.rt_last_error()
    for _ in range(3):                                   # replays are idempotent
        assert lib.rt_graph_launch(gr, stream.cuda_stream) == 0
    stream.synchronize()
    assert np.array_equal(_bits(acc.cpu().numpy()), _bits(g["acc"]))
    assert np.array_equal(host.numpy().view(np.uint32), g["packed"])
    # camera moves: the kernel nodes' parameters are replaced in the instantiated graph (no
    # re-capture), the graph's own eye-cone table is rebuilt by its build node; every frame is
    # compared with a direct render
    cam = rt.default_camera()
    for k in range(5):
        cam.Org.x, cam.Org.z, cam.Camyaw = 5.0 - 0.3 * k, 10.0 + 0.25 * k, 170.0 + 3.0 * k
        assert lib.rt_graph_set_camera(gr, C.byref(cam)) == 0, lib.rt_last_error()
        assert lib.rt_graph_launch(gr, stream.cuda_stream) == 0, lib.rt_last_error()
        stream.synchronize()
        rgba, packed, _ = _render(scene, w, h, spp=4, cam=cam)
        assert np.array_equal(host.numpy().view(np.uint32), packed), k
        assert not np.array_equal(packed, g["packed"])
    lib.rt_graph_destroy(gr)


def test_graph_never_replays_against_tables_of_another_frame(rt, gpu):
    """A graph built for camera A keeps rendering camera A after direct renders of the same
    scene with camera B (which builds other eye cones), at another resolution (other raygen
    tables), with a light moved (other column tables: the graph then follows the scene) and
    with another sphere list."""
    import torch
    lib = rt.load_library()
    w, h, n = 160, 90, 1024
    inp = Inputs(rt, n)
    scene = inp.scene()
    pk = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    cam_a = rt.default_camera()
    fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), cam=cam_a)
    gr = lib.rt_graph_capture(scene.handle, C.byref(fd), 1, None, stream.cuda_stream)
    assert gr, lib.rt_last_error()
    _, want_a, _ = _render(scene, w, h, cam=cam_a)

    def replay():
        pk.zero_()
        torch.cuda.synchronize()
        assert lib.rt_graph_launch(gr, stream.cuda_stream) == 0, lib.rt_last_error()
        stream.synchronize()
        return pk.cpu().numpy().view(np.uint32)

    assert np.array_equal(replay(), want_a)
    cam_b = rt.default_camera()
    cam_b.Org.x, cam_b.Org.y, cam_b.Camyaw, cam_b.Campitch = 1.0, 6.0, 140.0, -35.0
    for _ in range(4):                                   # more direct frames than there are cone slots
        cam_b.Org.x += 0.5
        _render(scene, w, h, cam=cam_b)
    assert np.array_equal(replay(), want_a)
    _render(scene, 96, 54, cam=cam_b)                    # another resolution rewrites the raygen tables
    assert np.array_equal(replay(), want_a)
    lights = rt.default_lights()
    lights[0].pos.x = 25.0
    scene.set_lights(lights, 3)
    _, want_l, _ = _render(scene, w, h, cam=cam_a)       # rebuilds the light tables in place
    got = replay()
    assert np.array_equal(got, want_l) and not np.array_equal(got, want_a)
    sph = rt.generate_spheres(n, 7)
    scene.set_spheres(sph, n)
    got = replay()
    _, want_s, _ = _render(scene, w, h, cam=cam_a)
    assert np.array_equal(got, want_s) and not np.array_equal(got, want_l)
    lib.rt_graph_destroy(gr)


def test_moving_camera_on_two_streams_matches_the_oracle(rt, gpu):
    """The reference moves `cam` every frame (checkKey, kernel.cu:1716-1764). Frames with a
    camera nudged each time alternate between two streams -- so that one frame's eye-cone table
    is rebuilt (on the device, in the frame's stream) while the previous frame is still reading
    another -- and every frame equals the oracle's."""
    import torch
    import oracle_py
    w, h, n = 96, 54, 256
    inp = Inputs(rt, n)
    scene = inp.scene()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [(torch.zeros((h, w), dtype=torch.int32, device="cuda"), torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"))
            for _ in range(8)]
    cams = []
    for k in range(8):
        cam = rt.default_camera()
        cam.Org.z = 10.0 + 0.1 * k                       # what checkKey's 'S' does, kernel.cu:1727
        cam.Org.x = 4.0 - 0.05 * k
        cam.Camyaw = 180.0 + 1.5 * k
        cams.append(cam)
        fd = scene.frame_desc(w, h, pixels=bufs[k][0].data_ptr(), rgba=bufs[k][1].data_ptr(), cam=cam)
        scene.render_raw(fd, streams[k & 1].cuda_stream)
    torch.cuda.synchronize()
    for k in range(8):
        inp.cam = cams[k]
        want_rgba, want, _ = inp.oracle_render(oracle_py, w, h)
        assert np.array_equal(bufs[k][0].cpu().numpy().view(np.uint32), want), k
        assert np.array_equal(_bits(bufs[k][1].cpu().numpy()), _bits(want_rgba)), k


def test_sphere_list_changes_between_frames_in_flight(rt, gpu):
    """Two frames in flight on two streams over one scene, the sphere list replaced in between:
    the upload waits for the frame that still reads the old table (it is not torn), and the next
    frame sees the new one."""
    import torch
    w, h, n = 960, 540, 1024
    scene = Inputs(rt, n).scene()
    other = rt.generate_spheres(n, 5)
    first = rt.generate_spheres(n, 1)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    b = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    for _ in range(3):
        scene.set_spheres(first, n)
        scene.render_raw(scene.frame_desc(w, h, pixels=a.data_ptr()), s0.cuda_stream)
        scene.set_spheres(other, n)                      # while the frame on s0 may still be running
        scene.render_raw(scene.frame_desc(w, h, pixels=b.data_ptr()), s1.cuda_stream)
        torch.cuda.synchronize()
        scene.set_spheres(first, n)
        _, want_a, _ = _render(scene, w, h, want_rgba=False)
        scene.set_spheres(other, n)
        _, want_b, _ = _render(scene, w, h, want_rgba=False)
        assert np.array_equal(a.cpu().numpy().view(np.uint32), want_a)
        assert np.array_equal(b.cpu().numpy().view(np.uint32), want_b)
        assert not np.array_equal(want_a, want_b)


def test_product_library_reads_no_environment(rt, gpu, monkeypatch):
    """The diagnostics of the tuning builds (RT_ABLATE & co.) do not exist in the product library:
    setting them changes nothing."""
    g = np.load(os.path.join(GOLD, "c2_160x90_n256.npz"))
    for k, v in (("RT_ABLATE", "16"), ("RT_NO_EYE_CONES", "1"), ("RT_NO_LIGHT_COLUMNS", "1"), ("RT_TABLE_LDS", "1")):
        monkeypatch.setenv(k, v)
    scene = Inputs(rt, 256).scene()
    rgba, packed, _ = _render(scene, 160, 90)
    assert np.array_equal(packed, g["packed"]) and np.array_equal(_bits(rgba[..., :3]), _bits(g["rgb"]))
    # update() reads no environment either: RT_GPUS belongs to the application shell and is looked at ONCE, by
    # onStart(). Five devices do not exist here -- a frame that obeyed the variable would end in rt_check().
    lib = rt.load_library()
    assert lib.rt_config_set_sphere_count(256) == 0 and lib.rt_config_set_seed(1) == 0 and lib.rt_config_set_gpus(0) == 0
    monkeypatch.delenv("RT_GPUS", raising=False)
    lib.rt_on_start()
    os.environ["RT_GPUS"] = "5"
    try:
        assert lib.rt_offscreen_resize(160, 90) == 0
        lib.rt_update()
        got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(90, 160)).copy()
        assert np.array_equal(got, g["packed"])
    finally:
        del os.environ["RT_GPUS"]


# ---------------------------------------------------------------- the reference's own surfaces
def _managed_sprite(rt, planes):
    """A sprite graph laid out as the reference does it (Sprite.cpp:13-52):
    three managed `buffer`s behind a managed `sprite`."""
    lib = rt.load_library()
    h, w = planes[0].shape
    sp = C.cast(lib.rt_managed_alloc(C.sizeof(rt.Sprite)), C.POINTER(rt.Sprite))
    bufs = []
    for p in planes:
        b = C.cast(lib.rt_managed_alloc(C.sizeof(rt.Buffer)), C.POINTER(rt.Buffer))
        data = lib.rt_managed_alloc(p.nbytes)
        C.memmove(data, p.ctypes.data, p.nbytes)
        b.contents.data = C.cast(data, C.POINTER(C.c_float))
        b.contents.size = p.nbytes
        bufs.append(b)
    sp.contents.rBuff, sp.contents.gBuff, sp.contents.bBuff = bufs
    sp.contents.width, sp.contents.height = w, h
    return sp


def test_raytrace_launch_signature_on_managed_scene(rt, gpu):
    """rt_launch_raytrace(pixels,width,height,aspect,objs,lights,light_size,cam,sky)
    with every argument in managed memory, as kernel.cu:1775-1783 passes them."""
    import torch
    lib = rt.load_library()
    w, h, n = 160, 90, 256
    g = np.load(os.path.join(GOLD, "c2_160x90_n256.npz"))
    inp = Inputs(rt, n)
    obj = C.cast(lib.rt_managed_alloc(C.sizeof(rt.Object)), C.POINTER(rt.Object))
    C.memset(obj, 0, C.sizeof(rt.Object))
    obj.contents.sphere_count = n
    dsp = lib.rt_managed_alloc(32 * n)
    C.memmove(dsp, inp.spheres, 32 * n)
    obj.contents.d_spheres = C.cast(dsp, C.POINTER(rt.Sphere))
    obj.contents.texture = _managed_sprite(rt, inp.tex)
    sky = C.cast(lib.rt_managed_alloc(C.sizeof(rt.Skybox)), C.POINTER(rt.Skybox))
    box = C.cast(lib.rt_managed_alloc(32), C.POINTER(rt.Sphere))
    C.memmove(box, C.byref(inp.sky_box), 32)
    sky.contents.box = box
    sky.contents.skyboxTex = _managed_sprite(rt, inp.sky)
    pixels = lib.rt_managed_alloc(4 * w * h)                       # cudaMallocManaged(pixels), kernel.cu:1775
    assert lib.rt_launch_raytrace(pixels, w, h, inp.aspect, obj, inp.lights, 3, inp.cam, sky, None) == 0, \
        lib.rt_last_error()
    torch.cuda.synchronize()
    got = np.ctypeslib.as_array(C.cast(pixels, C.POINTER(C.c_uint32)), shape=(h, w)).copy()
    assert np.array_equal(got, g["packed"])
    # the sphere list may change between frames (it is re-mirrored every launch)
    lib.rt_sphere_init(C.byref(obj.contents.d_spheres[0]), 4.0, 3.0, 7.0, 0.9)
    assert lib.rt_launch_raytrace(pixels, w, h, inp.aspect, obj, inp.lights, 3, inp.cam, sky, None) == 0
    torch.cuda.synchronize()
    got2 = np.ctypeslib.as_array(C.cast(pixels, C.POINTER(C.c_uint32)), shape=(h, w)).copy()
    sph2 = (rt.Sphere * n)()
    C.memmove(sph2, dsp, 32 * n)
    import oracle_py
    _, want2, _ = oracle_py.render(sph2, n, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam, w, h, inp.aspect,
                                   nthreads=8)
    assert np.array_equal(got2, want2) and not np.array_equal(got2, got)
    # planes and cubes ride along in the same object graph (kernel.cu:1213-1228)
    pl = C.cast(lib.rt_managed_alloc(40), C.POINTER(rt.Plane))
    lib.rt_plane_init(pl, 0.0, -4.0, 0.0, 0.0, 1.0, 0.0)          # the reference's plane, kernel.cu:1187
    cu = C.cast(lib.rt_managed_alloc(80 * 2), C.POINTER(rt.Cube))
    lib.rt_cube_init(C.byref(cu[0]), 1.0, 0.0, 1.0, 3.0, 2.0, 3.0)
    lib.rt_cube_init(C.byref(cu[1]), 6.0, 1.0, 2.0, 7.5, 2.5, 3.5)
    obj.contents.d_planes, obj.contents.plane_count = pl, 1
    obj.contents.d_cubes, obj.contents.cube_count = cu, 2
    assert lib.rt_launch_raytrace(pixels, w, h, inp.aspect, obj, inp.lights, 3, inp.cam, sky, None) == 0
    torch.cuda.synchronize()
    got3 = np.ctypeslib.as_array(C.cast(pixels, C.POINTER(C.c_uint32)), shape=(h, w)).copy()
    _, want3, _ = oracle_py.render(sph2, n, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam, w, h, inp.aspect,
                                   nthreads=8, cubes=cu, n_cubes=2, planes=pl, n_planes=1)
    assert np.array_equal(got3, want3) and not np.array_equal(got3, got2)



def test_onstart_update_present_path(rt, gpu):
    """onStart() + update() through the offscreen window (kernel.cuh:3-4,
    window.h:7-16): the presented buffer equals the oracle's packed frame, and a
    resize between frames keeps working (window.cpp:29-46)."""
    import oracle_py
    lib = rt.load_library()
    assert lib.rt_config_set_sphere_count(256) == 0 and lib.rt_config_set_seed(1) == 0
    lib.rt_on_start()
    inp = Inputs(rt, 256)
    for (w, h) in ((160, 90), (96, 54)):
        assert lib.rt_offscreen_resize(w, h) == 0
        lib.rt_update()
        got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
        _, want, _ = inp.oracle_render(oracle_py, w, h)
        assert np.array_equal(got, want), (w, h)
        assert lib.rt_last_frame_ms() > 0
    cam = lib.rt_config_camera()
    cam.contents.Org.z = 11.0                                     # what checkKey's 'S' does, kernel.cu:1727
    lib.rt_update()
    inp.cam.Org.z = 11.0
    _, want, _ = inp.oracle_render(oracle_py, 96, 54)
    got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(54, 96)).copy()
    assert np.array_equal(got, want)
    cam.contents.Org.z = 10.0


def test_onstart_ingests_ppm_textures_and_an_obj_mesh(rt, gpu, tmp_path):
    """The step Sprite.cpp:28-52 / kernel.cu:1181-1207 perform, end to end on the GPU: texture
    FILES (binary PPM instead of the JPEGs OpenCV decodes) and an OBJ file go through onStart()
    -> sprite(file) / mesh(file) -> planar float buffers / flat BVH -> update() -> the frame
    handed to setPixelBuff(); it equals the oracle fed the same arrays and the same OBJ text."""
    import oracle_py
    import meshes
    lib = rt.load_library()
    rng = np.random.default_rng(11)

    def write_ppm(path, h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 7 + yy * 3) % 256, (xx ^ yy) % 256, (xx * yy + 31) % 256], axis=-1).astype(np.uint8)
        img[rng.integers(0, h, 50), rng.integers(0, w, 50)] = rng.integers(0, 256, (50, 3), dtype=np.uint8)
        path.write_bytes(b"P6\n%d %d\n255\n" % (w, h) + img.tobytes())
        return [np.ascontiguousarray(img[..., c].astype(np.float32) / np.float32(255)) for c in range(3)]   # Sprite.cpp:43-45

    tex = write_ppm(tmp_path / "object.ppm", 96, 160)        # not square, not a power of two
    sky = write_ppm(tmp_path / "sky.ppm", 128, 300)
    obj_text = meshes.uv_sphere_obj()
    (tmp_path / "mesh.obj").write_text(obj_text)
    n = 64
    assert lib.rt_config_set_sphere_count(n) == 0 and lib.rt_config_set_seed(1) == 0
    assert lib.rt_config_set_assets(str(tmp_path / "object.ppm").encode(), str(tmp_path / "sky.ppm").encode(),
                                    str(tmp_path / "mesh.obj").encode()) == 0
    try:
        lib.rt_on_start()
        w, h = 160, 90
        assert lib.rt_offscreen_resize(w, h) == 0
        lib.rt_update()
        got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
    finally:
        lib.rt_config_set_assets(None, None, None)
    inp = Inputs(rt, n)
    om = oracle_py.Mesh(obj_text)
    _, want, cnt = oracle_py.render(inp.spheres, n, tex, sky, inp.sky_box, inp.lights, 3, inp.cam, w, h, inp.aspect,
                                    nthreads=8, mesh=om.handle)
    assert cnt["hit_pixels"] > 1000
    assert np.array_equal(got, want)
    # and the synthetic stand-ins are back once the assets are unset
    assert lib.rt_config_set_sphere_count(256) == 0
    lib.rt_on_start()
    lib.rt_update()
    _, want, _ = Inputs(rt, 256).oracle_render(oracle_py, w, h)
    assert np.array_equal(np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)), want)


# ---------------------------------------------------------------- full-size properties (BASELINE sizes)
def test_full_size_properties_c3(rt, gpu):
    """3840x2160 / 1024 spheres: too big for the oracle, so check what does not
    depend on size -- the golden row bands inside the full frame, culling vs the
    brute-force loops on a sub-band, packed == pack(float), determinism."""
    import oracle_py
    w, h, n = 3840, 2160, 1024
    scene = Inputs(rt, n).scene()
    rgba, packed, stats = _render(scene, w, h, want_stats=True)
    for name in ("c3_3840x2160_rows1080", "c3_3840x2160_rows300"):
        _, _, _, y0, y1 = GOLDEN_CASES[name]
        g = np.load(os.path.join(GOLD, name + ".npz"))
        assert np.array_equal(_bits(rgba[y0:y1, :, :3]), _bits(g["rgb"]))
        assert np.array_equal(packed[y0:y1], g["packed"])
    assert (rgba[..., 3] == 1).all() and np.isfinite(rgba).all()
    # pack(float channels) == packed words, everywhere
    v = (rgba[..., :3] * np.float32(254)).astype(np.int64)
    v = np.minimum(v, 255)
    assert np.array_equal(((v[..., 0] & 255) << 16) + ((v[..., 1] & 255) << 8) + (v[..., 2] & 255), packed)
    # determinism
    rgba2, packed2, _ = _render(scene, w, h)
    assert np.array_equal(_bits(rgba2), _bits(rgba)) and np.array_equal(packed2, packed)
    # brute force (the reference's loops as written, IEEE forms everywhere) over the WHOLE frame ==
    # the culled frame, all 2160 rows, for the default tile and one other
    rb, pb, sb = _render(scene, w, h, cull=False, want_stats=True)
    assert np.array_equal(_bits(rb), _bits(rgba)) and np.array_equal(pb, packed)
    assert sb["primary_tests"] == h * w * n
    r16, p16, _ = _render(scene, w, h, tile=16)
    assert np.array_equal(_bits(r16), _bits(rgba)) and np.array_equal(p16, packed)
    # workload statistics of SURVEY.md 8(d): 99.4 % primary hits
    assert 0.99 < stats["hit_pixels"] / (w * h) < 0.999
    assert stats["list_overflows"] >= 0


def test_full_size_c5_8k_band(rt, gpu):
    """7680x4320 / 4096 spheres (C5) is an 8-GPU config; one GPU renders rank 3's
    540-row band and a slice of it is compared with the brute-force loops."""
    w, h, n = 7680, 4320, 4096
    scene = Inputs(rt, n).scene()
    y0, y1 = rt.band_rows(h, 3, 8)
    assert (y0, y1) == (1620, 2160)
    rgba, packed, _ = _render(scene, w, h, y0=y0, y1=y1)
    rb, pb, _ = _render(scene, w, h, y0=y0, y1=y1, cull=False)          # the whole band, brute force
    assert np.array_equal(_bits(rb), _bits(rgba)) and np.array_equal(pb, packed)
    del rb, pb
    # the same rows as rank 3's share of the balanced split (16-row blocks dealt round-robin)
    rows = rt.interleaved_rows(h, 3, 8, 16)
    ri, pi, _ = _render(scene, w, h, interleave=(8, 3, 16), want_rgba=False)
    full_rows = [r for r in rows if y0 <= r < y1]
    idx = [rows.index(r) for r in full_rows]
    assert np.array_equal(pi[idx], packed[[r - y0 for r in full_rows]])


def test_full_size_c2_and_c4(rt, gpu):
    """C2 (1920x1080, 256 spheres) whole frame culled == brute force, with the golden bands inside;
    C4 (3840x2160, 1024 spheres, 4 spp): the golden bands inside the full frame rendered with the
    four samples in one launch, as four accumulate passes replayed from the hipGraph (the BASELINE
    config), and through the graph's copy to the present buffer."""
    import torch
    lib = rt.load_library()
    w, h, n = 1920, 1080, 256
    scene = Inputs(rt, n).scene()
    rgba, packed, _ = _render(scene, w, h)
    rb, pb, _ = _render(scene, w, h, cull=False)
    assert np.array_equal(_bits(rb), _bits(rgba)) and np.array_equal(pb, packed)
    for name in ("c2_1920x1080_rows100", "c2_1920x1080_rows700"):
        _, _, _, y0, y1 = GOLDEN_CASES[name]
        g = np.load(os.path.join(GOLD, name + ".npz"))
        assert np.array_equal(_bits(rgba[y0:y1, :, :3]), _bits(g["rgb"])) and np.array_equal(packed[y0:y1], g["packed"])
    w, h, n = 3840, 2160, 1024
    scene = Inputs(rt, n).scene()
    rgba4, packed4, _ = _render(scene, w, h, spp=4)                       # four samples inside one launch
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    pk = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    host = torch.zeros((h, w), dtype=torch.int32).pin_memory()
    stream = torch.cuda.Stream()
    fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), rgba=acc.data_ptr())
    for passes in (4, -4):   # one node with the sample loop inside / four progressive nodes
        acc.zero_(); pk.zero_(); host.zero_()
        gr = lib.rt_graph_capture(scene.handle, C.byref(fd), passes, host.data_ptr(), stream.cuda_stream)
        assert gr, lib.rt_last_error()
        for _ in range(2):
            assert lib.rt_graph_launch(gr, stream.cuda_stream) == 0
        stream.synchronize()
        assert np.array_equal(_bits(acc.cpu().numpy()), _bits(rgba4)), passes
        assert np.array_equal(host.numpy().view(np.uint32), packed4) and np.array_equal(pk.cpu().numpy().view(np.uint32), packed4)
        if passes == 4:
            lib.rt_graph_destroy(gr)
    for name, (_, _, _, _, y0, y1) in GOLDEN_SPP_CASES.items():
        g = np.load(os.path.join(GOLD, name + ".npz"))
        assert np.array_equal(_bits(rgba4[y0:y1]), _bits(g["acc"])), name
        assert np.array_equal(packed4[y0:y1], g["packed"]), name
    lib.rt_graph_destroy(gr)
    # the 4-spp frame by brute force as well (0.4 s)
    rb4, pb4, _ = _render(scene, w, h, spp=4, cull=False)
    assert np.array_equal(_bits(rb4), _bits(rgba4)) and np.array_equal(pb4, packed4)


def test_cxx_application_shell_with_its_own_window(rt, gpu, tmp_path):
    """examples/headless_app.cpp is the reference's wWinMain loop in plain C++: it
    defines window.h's functions itself (strong symbols beat the library's weak
    offscreen ones), calls onStart()/update() and dumps what setPixelBuff() received."""
    import subprocess
    import oracle_py
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "headless_app")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "ray-tracer-engine_amd", "csrc"), "examples"], check=True)
    out = tmp_path / "frame.ppm"
    r = subprocess.run([exe, "160", "90", "256", "3", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = out.read_bytes()
    assert raw.startswith(b"P6\n160 90\n255\n")
    rgb = np.frombuffer(raw[len(b"P6\n160 90\n255\n"):], dtype=np.uint8).reshape(90, 160, 3).astype(np.uint32)
    got = (rgb[..., 0] << 16) + (rgb[..., 1] << 8) + rgb[..., 2]
    _, want, _ = Inputs(rt, 256).oracle_render(oracle_py, 160, 90)
    assert np.array_equal(got, want)


def test_packed24_is_the_packed_frame_without_its_zero_byte(rt, gpu):
    """rt_launch_opts.packed24 (what a multi-GPU rank sends to the root): 3 bytes per pixel,
    B,G,R, written by the same launch; widened back it equals the 32-bit frame, for every tile
    shape, a row band, interleaved rows and 4 spp. Widths that are not a multiple of 4 are refused."""
    import torch
    from scenes import Inputs
    from ray_tracer_engine_amd import distributed as rd
    sc = Inputs(rt, 256).scene()
    for kw in ({}, {"tile": 16}, {"tile": 32}, {"tile": 64}, {"spp": 4}, {"y0": 24, "y1": 56},
               {"interleave": (3, 1, 16)}, {"cull": False}):
        out = sc.render(160, 88, want_packed24=True, **kw)
        torch.cuda.synchronize()
        wide = rd.unpack_rgb24(out["packed24"], 160)
        assert torch.equal(wide, out["packed"]), kw
        assert int((out["packed"].view(torch.uint8).view(-1, 4)[:, 3] != 0).sum()) == 0
    out = sc.render(164, 40, want_packed24=True)   # a multiple of 4 but not of the tile width
    torch.cuda.synchronize()
    assert torch.equal(rd.unpack_rgb24(out["packed24"], 164), out["packed"])
    with pytest.raises(rt.RtError):
        sc.render(162, 40, want_packed24=True)


# ---------------------------------------------------------------- several GPUs from the C ABI
def _multi_scene(rt, lib, m, inp):
    fp = C.POINTER(C.c_float)
    ptr = lambda a: a.ctypes.data_as(fp)
    assert lib.rt_multi_set_spheres(m, inp.spheres, inp.n) == 0
    h, w = inp.tex[0].shape
    assert lib.rt_multi_set_texture(m, ptr(inp.tex[0]), ptr(inp.tex[1]), ptr(inp.tex[2]), w, h) == 0
    h, w = inp.sky[0].shape
    assert lib.rt_multi_set_sky(m, C.byref(inp.sky_box), ptr(inp.sky[0]), ptr(inp.sky[1]), ptr(inp.sky[2]), w, h) == 0
    assert lib.rt_multi_set_lights(m, inp.lights, 3) == 0


@pytest.mark.parametrize("shares,w,h,n", [(1, 160, 90, 256), (1, 1920, 1080, 256), (2, 160, 90, 1024), (3, 164, 100, 256),
                                          (8, 160, 90, 1024), (8, 1920, 1080, 256), (8, 7680, 4320, 4096)])
def test_multi_device_frame_from_the_c_abi(rt, gpu, shares, w, h, n):
    """rt_multi_*: the frame split into 16-row blocks dealt round-robin, every share rendered as
    24-bit rows, gathered to the first device and scattered home by rt_scatter_rows24 -- equal to
    the single-GPU frame bit for bit. One GPU here, so `shares` > 1 uses the same device several
    times with the peer-copy transport (everything but the RCCL call itself runs) -- up to C5's own
    size, 7680x4320 / 4096 spheres in 8 shares; one share goes through RT_MULTI_RCCL, which runs
    the WHOLE exchange for its single rank: dlopen, ncclCommInitAll, the gather self-test of
    rt_multi_create, and per frame the 24-bit rows, the in-place ncclGather and the scatter kernel."""
    import torch
    lib = rt.load_library()
    inp = Inputs(rt, n)
    want = inp.scene().render(w, h, want_rgba=False)["packed"]
    m = C.c_void_p()
    devs = (C.c_int * shares)(*([0] * shares))
    transport = 1 if shares == 1 else 2
    assert lib.rt_multi_create_ex(devs, shares, transport, C.byref(m)) == 0, lib.rt_last_error()
    assert lib.rt_multi_device_count(m) == shares and lib.rt_multi_transport(m) == transport
    note = lib.rt_multi_note(m).decode()
    assert ("self-test passed" in note) if transport == 1 else ("peer-copy" in note), note
    _multi_scene(rt, lib, m, inp)
    sc = rt.Scene()
    fd = sc.frame_desc(w, h)
    out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    cams = []
    for k in range(5):                                   # frames in flight on both buffer sets, camera moving
        cam = rt.default_camera()
        cam.Org.z = 10.0 + 0.1 * k
        cams.append(cam)
        fd.cam = cam
        assert lib.rt_multi_render(m, C.byref(fd), out.data_ptr() if k == 4 else None) == 0, lib.rt_last_error()
    host = np.zeros((h, w), dtype=np.uint32)
    assert lib.rt_multi_download(m, host.ctypes.data) == 0
    want4 = inp.scene().render(w, h, want_rgba=False, cam=cams[4])["packed"]
    torch.cuda.synchronize()
    assert torch.equal(out, want4) and np.array_equal(host, want4.cpu().numpy().view(np.uint32))
    fd.cam = rt.default_camera()
    assert lib.rt_multi_render(m, C.byref(fd), None) == 0
    assert lib.rt_multi_download(m, host.ctypes.data) == 0
    assert np.array_equal(host, want.cpu().numpy().view(np.uint32))
    assert lib.rt_multi_gathers(m) == (6 if transport == 1 else 0)       # one ncclGather group per frame
    # the same frame in three row bands (what update() does to overlap the copy to the host with the next band)
    out.zero_()
    bands = [(0, (h // 3) & ~15), ((h // 3) & ~15, (2 * h // 3) & ~15), ((2 * h // 3) & ~15, h)]
    for (y0, y1) in bands:
        if y1 > y0:
            fd.opts.y0, fd.opts.y1 = y0, y1
            assert lib.rt_multi_render(m, C.byref(fd), out.data_ptr()) == 0, lib.rt_last_error()
    assert lib.rt_multi_sync(m) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    fd.opts.y0, fd.opts.y1 = 8, 24                                       # not whole blocks: refused
    assert lib.rt_multi_render(m, C.byref(fd), out.data_ptr()) != 0
    lib.rt_multi_destroy(m)
    if n == 4096:   # C5: a band of the assembled frame against the reference's loops as written
        sc5 = inp.scene()
        rb = sc5.render(w, h, y0=2144, y1=2176, cull=False, want_rgba=False)["packed"]
        torch.cuda.synchronize()
        assert torch.equal(rb, want[2144:2176])


def test_update_on_several_shares(rt, gpu):
    """update() with rt_config_set_gpus: the presented frame equals the oracle's (rehearsed on one
    GPU: three shares of the frame on device 0)."""
    import oracle_py
    lib = rt.load_library()
    assert lib.rt_config_set_sphere_count(256) == 0 and lib.rt_config_set_seed(1) == 0
    assert lib.rt_config_set_gpus(-3) == 0
    try:
        lib.rt_on_start()
        inp = Inputs(rt, 256)
        for (w, h) in ((160, 90), (96, 54), (162, 90)):   # 162: no 24-bit rows -> that frame on the first device alone
            assert lib.rt_offscreen_resize(w, h) == 0
            lib.rt_update()
            got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
            _, want, _ = inp.oracle_render(oracle_py, w, h)
            assert np.array_equal(got, want), (w, h)
        # tall enough for update()'s three row bands (each gathered, scattered and copied while the next renders),
        # camera moved between frames; against the single-GPU render of the same frames
        import torch
        w, h = 1920, 1080
        assert lib.rt_offscreen_resize(w, h) == 0
        cam = lib.rt_config_camera()
        sc = inp.scene()
        for z in (10.0, 10.3, 10.0):
            cam.contents.Org.z = z
            lib.rt_update()
            got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
            c = rt.default_camera()
            c.Org.z = z
            want = sc.render(w, h, want_rgba=False, cam=c)["packed"]
            torch.cuda.synchronize()
            assert np.array_equal(got, want.cpu().numpy().view(np.uint32)), z
        # a plane that MOVES between two frames moves on every share (contents are compared, not counts)
        w, h = 160, 96
        assert lib.rt_offscreen_resize(w, h) == 0
        obj = lib.rt_config_object()
        pl = C.cast(lib.rt_managed_alloc(40), C.POINTER(rt.Plane))
        lib.rt_plane_init(pl, 0.0, -4.0, 0.0, 0.0, 1.0, 0.0)
        obj.contents.d_planes, obj.contents.plane_count = pl, 1
        lib.rt_update()
        first = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
        lib.rt_plane_init(pl, 0.0, -1.0, 0.0, 0.0, 1.0, 0.0)
        lib.rt_update()
        moved = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
        _, want, _ = oracle_py.render(inp.spheres, inp.n, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam, w, h, inp.aspect,
                                      nthreads=8, planes=pl, n_planes=1)
        assert np.array_equal(moved, want) and not np.array_equal(moved, first)
        obj.contents.d_planes, obj.contents.plane_count = None, 0
    finally:
        lib.rt_config_set_gpus(0)


@pytest.mark.parametrize("world", [2, 5, 8])
def test_root_side_assembly_kernel(rt, gpu, world):
    """What bench.py's root does per frame at N > 1: the ranks' 24-bit rows (here rendered one after
    the other on this GPU) sit in the gather's receive slots; rt_assemble_rows24 -- one kernel of the
    library -- puts every row in place and widens it; the result is the single-GPU frame."""
    import torch
    from ray_tracer_engine_amd import distributed as rd
    w, h, n = 160, 90, 256
    sc = Inputs(rt, n).scene()
    want = sc.render(w, h, want_rgba=False)["packed"]
    root = rd.InterleavedGather(h, w, world, "cuda", 16, rgb24=True)
    for r in range(world):
        out = sc.render(w, h, interleave=(world, r, 16), want_packed24=True, want_rgba=False)
        root.views[r][: out["packed24"].shape[0]].copy_(out["packed24"])
    frame = root.assemble()
    torch.cuda.synchronize()
    assert torch.equal(frame, want)


def test_fast_mode_stays_within_the_tolerance_away_from_flips(rt, gpu):
    """rt_launch_opts.fast (opt-in, approximate sample construction / shadow tests / texel coordinates):
    everything that enters a pixel continuously -- primary hit, normal, toL -- stays exact, so against the
    exact frame (= the oracle's) a pixel is either bit-identical or one where a discrete decision went the
    other way (a shadow sample: 0.1 of a light's brightness; a neighbouring texel), and those are a few in
    10^5. north_star's tolerance (1e-5 relative per channel) holds on every other pixel with error 0.
    Never the default: the same call without the flag is bit-exact, and configurations the mode does not
    cover (brute force, other tiles, the work counters, ...) ignore the flag and stay exact."""
    w, h, n = 960, 540, 1024
    scene = Inputs(rt, n).scene()
    e, pe, _ = _render(scene, w, h)
    f, pf, _ = _render(scene, w, h, fast=True)
    rel = np.abs(f[..., :3].astype(np.float64) - e[..., :3]) / np.maximum(np.abs(e[..., :3]), 1e-3)
    worst = rel.max(axis=2)
    flipped = worst > 1e-5
    assert flipped.mean() < 3e-4, flipped.mean()
    assert (worst[~flipped] == 0).all()                  # not flipped = not touched
    assert np.array_equal(pe[~flipped], pf[~flipped])
    assert (pe == pf).mean() > 0.9997
    for kw in (dict(cull=False), dict(tile=16), dict(want_stats=True), dict(force_slow=True), dict(table_lds=True)):
        g, pg, _ = _render(scene, w, h, fast=True, **kw)
        assert np.array_equal(_bits(g), _bits(e)) and np.array_equal(pg, pe), kw


@pytest.mark.gpu
def test_tile_order_changes_the_schedule_not_the_pixels(rt, gpu):
    """rt_scene_set_tile_order: launches whose tiles start in the order the library sorts from the wave durations the
    kernel records (blocks of 16 x 16 tiles, longest first; re-sorted after 1, 2, 4, 8, ... launches of an unchanged view
    and every third launch of a moving one) render the same bits as grid-order launches -- whole frame, a row band,
    interleaved rows, two streams sharing the scene, a camera that rests and moves, layouts evicting each other."""
    import torch
    w, h = 640, 360
    scene = rt.Scene.default(256)
    plain = rt.Scene.default(256)
    plain.set_tile_order(0)
    cams = []
    for k in range(80):
        cam = rt.default_camera()
        cam.Org.z += 0.05 * ((k // 12) % 7)        # rests for twelve launches (six per layout), then moves
        cam.Camyaw += 0.5 * ((k // 12) % 5) + (0.25 * (k % 3) if 36 <= k < 48 else 0.0)   # ... and once keeps moving
        cams.append(cam)
    layouts = [dict(), dict(y0=64, y1=200), dict(interleave=(3, 1, 16)), dict(tile=16), dict(y0=0, y1=72), dict(interleave=(2, 0, 16))]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for k, cam in enumerate(cams):
        kw = layouts[(k // 3) % len(layouts)] if k >= 70 else layouts[k % 2]     # two layouts alternating (35 launches each), then all six
        with torch.cuda.stream(streams[k & 1]):
            a = scene.render(w, h, cam=cam, stream=streams[k & 1], **kw)
        b = plain.render(w, h, cam=cam, **kw)
        torch.cuda.synchronize()
        assert torch.equal(a["packed"], b["packed"]), (k, kw)
        assert torch.equal(a["rgba"].view(torch.int32), b["rgba"].view(torch.int32)), (k, kw)
    # No host synchronisation between launches: frames of both streams and several layouts in flight while orders are
    # sorted and slots of the layout cache are recycled (the sort waits on the device for every frame launched so far,
    # later launches on the other stream wait for its event). Every launch has its own output buffers; all of them are
    # compared with the grid-order frames after ONE final synchronize.
    torch.cuda.synchronize()
    outs = []
    for k in range(60):
        cam = cams[(k * 7) % len(cams)] if k % 10 >= 6 else cams[0]          # mostly resting, moving now and then
        kw = layouts[(k // 2) % len(layouts)] if k >= 30 else layouts[k % 3]   # three layouts, then all six (evictions)
        st = streams[k & 1]
        with torch.cuda.stream(st):
            outs.append((k, kw, cam, scene.render(w, h, cam=cam, stream=st, want_rgba=False, **kw)))
    torch.cuda.synchronize()
    for k, kw, cam, a in outs:
        b = plain.render(w, h, cam=cam, want_rgba=False, **kw)
        torch.cuda.synchronize()
        assert torch.equal(a["packed"], b["packed"]), (k, kw)
