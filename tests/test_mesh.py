"""Triangle-mesh path: OBJ loader + flat BVH of the
product against the oracle's restatement (CPU), Moller-Trumbore KATs, and
bit-exact rendering on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import meshes

OBJS = {
    "uv_sphere_quads": meshes.uv_sphere_obj(),
    "uv_sphere_tris": meshes.uv_sphere_obj(quads=False, n_lat=7, n_lon=9),
    "box_bare_indices": meshes.box_obj_no_normals(),
    "normals_only_quad": meshes.normals_only_obj(),
    "with_blank_and_comment_lines": "# c\n\nv 0 0 0\nv 1 0 0\n\nv 0 1 0\r\nf 1 2 3\n",
}


def test_mesh_struct_layouts(rt, oracle):
    assert C.sizeof(rt.Triangle) == 108 == C.sizeof(oracle.OTriangle)      # kernel.cu:1018-1020
    assert C.sizeof(rt.BvhBox) == 40 and rt.BvhBox.length.offset == 32      # Bvhbox, kernel.cu:512-543
    assert rt.Mesh.poly_count.offset == 16 and rt.Mesh.has_normals.offset == 28 and rt.Mesh.h_box.offset == 32


@pytest.mark.parametrize("name", sorted(OBJS))
def test_loader_and_bvh_match_oracle(name, rt, oracle):
    txt = OBJS[name]
    m = rt.mesh_from_obj_text(txt)
    om = oracle.Mesh(txt)
    mm = m.contents
    assert (mm.poly_count, mm.bvhbox_count, bool(mm.has_normals)) == (om.poly_count, om.bvhbox_count, om.has_normals)
    assert mm.bvhLayer_count == 10
    tri = np.ctypeslib.as_array(C.cast(mm.d_tri_arr, C.POINTER(C.c_float)), shape=(mm.poly_count, 27))
    assert np.array_equal(tri.view(np.uint32), om.triangles().view(np.uint32))
    seen = []
    for j, (b, org, idx) in enumerate(om.boxes()):
        bx = mm.d_box[j]
        c = bx.d_bvhbox.contents
        assert [c.bounds[0].x, c.bounds[0].y, c.bounds[0].z, c.bounds[1].x, c.bounds[1].y, c.bounds[1].z] == b
        assert [c.orgin.x, c.orgin.y, c.orgin.z] == org
        assert [bx.d_indexes[i] for i in range(bx.length)] == idx
        seen += idx
    assert sorted(seen) == list(range(mm.poly_count))          # every triangle in exactly one leaf
    rt.load_library().rt_mesh_free(m)


def test_loader_quirks(rt):
    """Quirks of the reference loader that are reproduced (kernel.cu:660-675, 700-712):
    a quad's second half keeps the first half's face normal and repeats vt[2];
    the a//c quad's second half repeats the first three vertex normals."""
    txt = ("v 0 0 0\nv 1 0 0\nv 1 1 0.5\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
           "vn 0 0 1\nvn 0 0.6 0.8\nvn 0.6 0 0.8\nvn 1 0 0\nf 1/1/1 2/2/2 3/3/3 4/4/4\n")
    mm = rt.mesh_from_obj_text(txt).contents
    assert mm.poly_count == 2 and mm.has_normals == 1
    t1, t2 = mm.d_tri_arr[0], mm.d_tri_arr[1]
    assert (t2.points[1].x, t2.points[1].y, t2.points[1].z) == (1, 1, 0.5)      # 0,2,3
    assert (t2.normal.x, t2.normal.y, t2.normal.z) == (t1.normal.x, t1.normal.y, t1.normal.z)
    assert (t2.vt[1].u, t2.vt[1].v) == (1, 1) and (t2.vt[2].u, t2.vt[2].v) == (1, 1)   # vt[pvt[2]] twice
    assert (t2.vecNormal[2].x) == 1.0                                           # vn[pvn[3]]
    mm2 = rt.mesh_from_obj_text(meshes.normals_only_obj()).contents
    q2 = mm2.d_tri_arr[1]
    assert (q2.vecNormal[1].y, q2.vecNormal[2].x) == (np.float32(0.1), np.float32(0.1))   # vn[pvn[1]], vn[pvn[2]]
    bare = rt.mesh_from_obj_text(meshes.box_obj_no_normals()).contents
    assert bare.has_normals == 0
    assert bare.d_tri_arr[0].vt[0].u == np.float32(0.666413)
    lib = rt.load_library()
    assert not lib.rt_mesh_from_obj_text(b"v 0 0 0\n")                     # no faces
    assert not lib.rt_mesh_from_obj_text(b"v 0 0 0\nf 1 2 3\n")            # index out of range
    assert not lib.rt_mesh_load_obj(b"/nonexistent.obj")


def test_bvh_split_rule(rt):
    """createBvhMesh (kernel.cu:752-937): leaves of <= 5 triangles are kept, larger
    ones are cut at the middle of their bounds by the first vertex, the axis
    advancing y, x, z after every cut."""
    lines = []
    for i in range(12):            # 12 small triangles stacked along y
        y = float(i)
        lines += ["v 0 %g 0" % y, "v 1 %g 0" % y, "v 0 %g 1" % (y + 0.25)]
    lines += ["f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3) for i in range(12)]
    mm = rt.mesh_from_obj_text("\n".join(lines) + "\n").contents
    leaves = [[mm.d_box[j].d_indexes[i] for i in range(mm.d_box[j].length)] for j in range(mm.bvhbox_count)]
    # pass 0: y-cut of [0, 11.25] at 5.625 -> 0..5 | 6..11 (axis -> x). pass 1: 0..5 cut along x: every
    # first vertex has x = 0 <= 0.5, the other half is empty, the leaf survives (axis -> z); 6..11 cut
    # along z: same (axis -> y). pass 2: 0..5 cut along y at (0 + 5.25)/2 -> 0,1,2 | 3,4,5 (axis -> x);
    # 6..11 along x: survives (-> z). pass 3: 6..11 along z: survives (-> y). pass 4: y at (6 + 11.25)/2.
    assert leaves == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]]


def test_moller_trumbore_kats(oracle):
    lib = oracle.load()
    tri = oracle.OTriangle()
    for k, p in enumerate(((0, 0, 5), (1, 0, 5), (0, 1, 5))):
        tri.points[k] = oracle.OVec3(*p)
    t, u, v = C.c_float(), C.c_float(), C.c_float()

    def hit(org, d):
        r = oracle.ORay(oracle.OVec3(*org), oracle.OVec3(*d))
        return lib.oracle_triangle_intersect(C.byref(tri), C.byref(r), C.byref(t), C.byref(u), C.byref(v))

    assert hit((0.25, 0.25, 0), (0, 0, 1)) == 1 and (t.value, u.value, v.value) == (5.0, 0.25, 0.25)
    assert hit((0.25, 0.25, 0), (0, 0, -1)) == 0                  # behind: t = -5 fails t > 1e-7
    assert hit((0.75, 0.75, 0), (0, 0, 1)) == 0                   # u + v > 1
    assert hit((-0.1, 0.2, 0), (0, 0, 1)) == 0                    # u < 0
    assert hit((0.25, 0.25, 0), (1, 0, 0)) == 0                   # parallel: |a| < 1e-7
    assert hit((0.25, 0.25, 10), (0, 0, -1)) == 1 and t.value == 5.0   # no back-face culling
    assert hit((0.0, 0.0, 0), (0, 0, 1)) == 1 and (u.value, v.value) == (0.0, 0.0)   # on a vertex: inclusive


def test_mesh_golden(rt, oracle):
    from scenes import Inputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh_160x90.npz"))
    inp = Inputs(rt, 64)
    om = oracle.Mesh(meshes.uv_sphere_obj())
    rgba, packed, cnt = oracle.render(inp.spheres, inp.n, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam,
                                      160, 90, inp.aspect, nthreads=8, mesh=om.handle)
    assert np.array_equal(rgba[..., :3].view(np.uint32), g["rgb"].view(np.uint32)) and np.array_equal(packed, g["packed"])
    assert cnt["hit_pixels"] == int(g["counters"][2]) == 7257


# ------------------------------------------------------------------ GPU
def _render_both(rt, inp, mesh, w, h, **kw):
    import oracle_py
    import torch
    om = oracle_py.Mesh(mesh)
    rgba, packed, cnt = oracle_py.render(inp.spheres, inp.n, inp.tex, inp.sky, inp.sky_box, inp.lights, inp.n_lights,
                                         inp.cam, w, h, inp.aspect, nthreads=16, mesh=om.handle,
                                         cubes=getattr(inp, "cubes", None), n_cubes=getattr(inp, "n_cubes", 0),
                                         planes=getattr(inp, "planes", None), n_planes=getattr(inp, "n_planes", 0))
    sc = inp.scene()
    if getattr(inp, "n_planes", 0):
        sc.set_planes(inp.planes, inp.n_planes)
    if getattr(inp, "n_cubes", 0):
        sc.set_cubes(inp.cubes, inp.n_cubes)
    pm = rt.mesh_from_obj_text(mesh)
    sc.set_mesh(pm)
    # every way a mesh scene can be launched: culled / brute force, the four tile shapes, all shortcuts
    # off, with the work counters, whole-table LDS staging
    for opts in (dict(cull=True), dict(cull=False), dict(cull=True, tile=16), dict(cull=True, tile=32), dict(cull=False, tile=64),
                 dict(cull=True, force_slow=True), dict(cull=True, want_stats=True), dict(cull=True, table_lds=True)) + tuple(kw.get("more", ())):
        out = sc.render(w, h, cam=inp.cam, **opts)
        torch.cuda.synchronize()
        got = out["rgba"].cpu().numpy()
        bad = int((got.view(np.uint32) != rgba.view(np.uint32)).any(axis=2).sum())
        assert bad == 0, (opts, bad)
        assert np.array_equal(out["packed"].cpu().numpy().view(np.uint32), packed), opts
        if "stats" in out:
            assert out["stats"]["hit_pixels"] == cnt["hit_pixels"]
    return cnt


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["uv_sphere_quads", "uv_sphere_tris", "box_bare_indices", "normals_only_quad"])
def test_mesh_render_matches_oracle(name, rt, gpu):
    from scenes import Inputs
    inp = Inputs(rt, 64)
    cnt = _render_both(rt, inp, OBJS[name], 160, 90)
    assert cnt["hit_pixels"] >= 3187         # at least the 64 spheres alone


@pytest.mark.gpu
def test_mesh_with_every_primitive_kind(rt, gpu):
    from scenes import mixed_scene
    inp = mixed_scene(rt)
    _render_both(rt, inp, meshes.uv_sphere_obj(cx=5.0, cy=3.0, cz=6.0, r=1.2, n_lat=8, n_lon=12), 128, 80)


@pytest.mark.gpu
def test_mesh_only_scene_and_shim(rt, gpu):
    """No spheres, only the mesh; then the same through rt_launch_raytrace with
    objs->mesh1 pointing at the reference-layout mesh."""
    import oracle_py
    import torch
    from scenes import Inputs
    from test_gpu_parity import _managed_sprite
    lib = rt.load_library()
    inp = Inputs(rt, 0)
    txt = meshes.uv_sphere_obj(cx=4.0, cy=1.5, cz=6.0, r=2.0, n_lat=12, n_lon=20)
    _render_both(rt, inp, txt, 128, 80)
    w, h = 128, 80
    om = oracle_py.Mesh(txt)
    _, want, _ = oracle_py.render(inp.spheres, 0, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam, w, h,
                                  inp.aspect, nthreads=16, mesh=om.handle)
    obj = C.cast(lib.rt_managed_alloc(C.sizeof(rt.Object)), C.POINTER(rt.Object))
    C.memset(obj, 0, C.sizeof(rt.Object))
    obj.contents.texture = _managed_sprite(rt, inp.tex)
    obj.contents.mesh1 = rt.mesh_from_obj_text(txt)
    sky = C.cast(lib.rt_managed_alloc(C.sizeof(rt.Skybox)), C.POINTER(rt.Skybox))
    box = C.cast(lib.rt_managed_alloc(32), C.POINTER(rt.Sphere))
    C.memmove(box, C.byref(inp.sky_box), 32)
    sky.contents.box = box
    sky.contents.skyboxTex = _managed_sprite(rt, inp.sky)
    pixels = lib.rt_managed_alloc(4 * w * h)
    assert lib.rt_launch_raytrace(pixels, w, h, inp.aspect, obj, inp.lights, 3, inp.cam, sky, None) == 0, lib.rt_last_error()
    torch.cuda.synchronize()
    got = np.ctypeslib.as_array(C.cast(pixels, C.POINTER(C.c_uint32)), shape=(h, w)).copy()
    assert np.array_equal(got, want)
    obj.contents.mesh1 = None                                   # and without it again (cache must notice)
    assert lib.rt_launch_raytrace(pixels, w, h, inp.aspect, obj, inp.lights, 3, inp.cam, sky, None) == 0
    torch.cuda.synchronize()
    got0 = np.ctypeslib.as_array(C.cast(pixels, C.POINTER(C.c_uint32)), shape=(h, w)).copy()
    _, want0, _ = oracle_py.render(inp.spheres, 0, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam, w, h,
                                   inp.aspect, nthreads=16)
    assert np.array_equal(got0, want0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(6)))
def test_mesh_fuzz_triangle_soup(rt, gpu, seed):
    """Random triangle soups (degenerate and tiny triangles included) among random
    spheres, random camera: culled == brute force == oracle."""
    from test_gpu_scenarios import Scn, _cam
    rng = np.random.default_rng(500 + seed)
    nt = int(rng.integers(8, 120))
    lines = []
    for _ in range(nt):
        c = rng.uniform(0, 10, 3)
        for _ in range(3):
            p = c + rng.normal(size=3) * float(rng.choice([0.05, 0.6, 2.0]))
            lines.append("v %.5f %.5f %.5f" % tuple(p))
    if seed % 2:
        lines += ["vn %.4f %.4f %.4f" % tuple(rng.normal(size=3)) for _ in range(7)]
        lines += ["f %d//%d %d//%d %d//%d" % (3 * i + 1, i % 7 + 1, 3 * i + 2, (i + 1) % 7 + 1, 3 * i + 3, (i + 2) % 7 + 1)
                  for i in range(nt)]
    else:
        lines += ["f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3) for i in range(nt)]
    txt = "\n".join(lines) + "\n"
    sph = [tuple(rng.uniform(0, 10, 3)) + (rng.uniform(0.2, 0.9),) for _ in range(int(rng.integers(0, 60)))]
    sc = Scn(rt, sph, cam=_cam(rt, tuple(rng.uniform(-2, 12, 3)), float(rng.uniform(0, 360)), float(rng.uniform(-40, 40))))
    inp = sc
    inp.scene_obj = None
    _render_both(rt, inp, txt, 96, 64)


def _grazing_triangles_obj(rt, cam_org, w, h, aspect, rows, per_row, seed):
    """Triangles lying (almost) in the planes that whole pixel ROWS of rays span: with yaw = pitch = 0 the rays of row
    y are O + s (dx, dy(y), 1/aspect), O = cam.Org + (0, 0, -1/aspect) (kernel.cu:1624-1631), a plane through O. Tilted
    out of that plane by 0 ... 1e-3 rad they are grazed by hundreds of rays -- where Moller-Trumbore's rounding error
    is unbounded and a bounding-sphere cull alone would be unsound (beam_keeps_triangle's edge-on guard)."""
    rng = np.random.default_rng(seed)
    ez = 1.0 / aspect
    O = np.array([cam_org[0], cam_org[1], cam_org[2] - ez])
    lines = []
    for y in rows:
        dy = float(np.float32(aspect * (2 * (y + 0.5) / np.float32(h)) * (np.float32(h) / w) - 1))
        for _ in range(per_row):
            tilt = float(rng.choice([0.0, 1e-6, -1e-5, 1e-4, -3e-4, 1e-3])) + float(rng.normal()) * 1e-6
            f0 = np.array([0.0, dy, ez]); f0 /= np.linalg.norm(f0)                    # forward, in the row plane
            n = np.array([0.0, f0[2], -f0[1]])                                        # the row plane's normal
            f = f0 * np.cos(tilt) + n * np.sin(tilt)                                  # the triangle's plane: tilted about the
            dist, x0, size = rng.uniform(2.5, 9.0), rng.uniform(-3.0, 3.0), float(rng.choice([0.15, 0.5, 1.5]))
            c = O + f0 * dist + np.array([x0, 0.0, 0.0])                              # line {c + a e_x}, which the row's rays meet
            for _ in range(3):
                a, b = rng.normal(size=2) * size
                p = c + np.array([a, 0.0, 0.0]) + f * b
                lines.append("v %.7f %.7f %.7f" % tuple(p))
    nt = len(lines) // 3
    lines += ["f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3) for i in range(nt)]
    return "\n".join(lines) + "\n"


@pytest.mark.gpu
def test_mesh_triangles_grazed_by_whole_pixel_rows(rt, gpu):
    """Culled == brute force == oracle where rays run (almost) inside triangle planes."""
    from test_gpu_scenarios import Scn, _cam
    w, h = 224, 144
    org = (0.5, 0.25, -1.0)
    sc = Scn(rt, [(0.5, 0.0, 7.0, 0.8), (2.5, 1.0, 6.0, 0.6)], cam=_cam(rt, org, 0.0, 0.0))
    txt = _grazing_triangles_obj(rt, org, w, h, sc.aspect, rows=range(8, h - 8, 9), per_row=4, seed=11)
    cnt = _render_both(rt, sc, txt, w, h)
    assert cnt["hit_pixels"] > 500


@pytest.mark.gpu
def test_mesh_grazing_rows_full_hd_cull_equals_brute(rt, gpu):
    """The same at 1920x1080 (thin beams, 300 000 grazing ray/triangle pairs): culling kernel against the brute-force one."""
    import torch
    from test_gpu_scenarios import Scn, _cam
    w, h = 1920, 1080
    org = (0.5, 0.25, -1.0)
    sc = Scn(rt, [(0.5, 0.0, 7.0, 0.8)], cam=_cam(rt, org, 0.0, 0.0))
    txt = _grazing_triangles_obj(rt, org, w, h, sc.aspect, rows=range(20, h - 20, 13), per_row=3, seed=12)
    scene = sc.scene()
    scene.set_mesh(rt.mesh_from_obj_text(txt))
    a = scene.render(w, h, cam=sc.cam, cull=True)
    b = scene.render(w, h, cam=sc.cam, cull=False)
    torch.cuda.synchronize()
    assert torch.equal(a["rgba"].view(torch.int32), b["rgba"].view(torch.int32))
    assert torch.equal(a["packed"], b["packed"])
