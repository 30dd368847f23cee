"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module. The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OVec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class ORay(C.Structure):
    _fields_ = [("Org", OVec3), ("Dir", OVec3)]


class OCamera(C.Structure):
    _fields_ = [("Org", OVec3), ("Dir", OVec3), ("aspect", C.c_float), ("Camyaw", C.c_float), ("Campitch", C.c_float)]


class OLight(C.Structure):
    _fields_ = [("pos", OVec3), ("size", C.c_float), ("r", C.c_float), ("g", C.c_float), ("b", C.c_float)]


class OSphere(C.Structure):
    _fields_ = [("vptr", C.c_void_p), ("orgin", OVec3), ("reflective", C.c_ubyte), ("radius", C.c_float)]


class OPlane(C.Structure):
    _fields_ = [("vptr", C.c_void_p), ("orgin", OVec3), ("reflective", C.c_ubyte), ("normal", OVec3)]


class OCube(C.Structure):
    _fields_ = [("vptr", C.c_void_p), ("orgin", OVec3), ("normals", OVec3 * 3), ("bounds", OVec3 * 2)]


class OTriangle(C.Structure):
    _fields_ = [("points", OVec3 * 3), ("normal", OVec3), ("vecNormal", OVec3 * 3), ("vt", (C.c_float * 2) * 3)]


class OSprite(C.Structure):
    _fields_ = [("r", C.POINTER(C.c_float)), ("g", C.POINTER(C.c_float)), ("b", C.POINTER(C.c_float)),
                ("width", C.c_int), ("height", C.c_int)]


class OFrame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("aspect", C.c_float),
                ("spheres", C.POINTER(OSphere)), ("sphere_count", C.c_int),
                ("texture", C.POINTER(OSprite)), ("lights", C.POINTER(OLight)), ("light_size", C.c_int),
                ("cam", OCamera), ("sky_box", C.POINTER(OSphere)), ("sky_tex", C.POINTER(OSprite)),
                ("y0", C.c_int), ("y1", C.c_int), ("off_x", C.c_double), ("off_y", C.c_double),
                ("cubes", C.POINTER(OCube)), ("cube_count", C.c_int),
                ("planes", C.POINTER(OPlane)), ("plane_count", C.c_int), ("mesh", C.c_void_p)]


_libs = {}


def build(force: bool = False):
    """Compile the oracle (gcc, no GPU involved)."""
    so = os.path.join(_HERE, "librt_oracle.so")
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "librt_oracle_libm.so")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "all"], check=True,
                       stdout=subprocess.DEVNULL)


def load(libm: bool = False):
    key = "libm" if libm else "rt"
    if key in _libs:
        return _libs[key]
    build()
    lib = C.CDLL(os.path.join(_HERE, "librt_oracle_libm.so" if libm else "librt_oracle.so"))
    cf, ci = C.c_float, C.c_int
    lib.oracle_render.restype = ci
    lib.oracle_render.argtypes = [C.POINTER(OFrame), C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint64), ci]
    lib.oracle_sphere_intersect.restype = ci
    lib.oracle_sphere_intersect.argtypes = [C.POINTER(OSphere), C.POINTER(ORay), C.POINTER(cf)]
    lib.oracle_rgb_to_int.restype = C.c_uint32
    lib.oracle_rgb_to_int.argtypes = [ci, ci, ci]
    lib.oracle_f2i.restype = ci
    lib.oracle_f2i.argtypes = [cf]
    lib.oracle_make_sphere.restype = None
    lib.oracle_make_sphere.argtypes = [C.POINTER(OSphere), cf, cf, cf, cf]
    lib.oracle_default_aspect.restype = cf
    lib.oracle_primary_ray.restype = None
    lib.oracle_primary_ray.argtypes = [ci, ci, ci, ci, cf, C.POINTER(OCamera), C.c_double, C.c_double, C.POINTER(ORay)]
    lib.oracle_rotate_dir.restype = None
    lib.oracle_rotate_dir.argtypes = [C.POINTER(OCamera), C.POINTER(OVec3), cf, cf, C.POINTER(OVec3)]
    lib.oracle_cast_light_ray.restype = cf
    lib.oracle_cast_light_ray.argtypes = [C.POINTER(OSphere), ci, C.POINTER(OVec3), C.POINTER(OLight), C.POINTER(OVec3)]
    lib.oracle_light_dirs.restype = None
    lib.oracle_light_dirs.argtypes = [C.POINTER(OVec3), C.POINTER(OLight), C.POINTER(cf)]
    lib.oracle_plane_intersect.restype = ci
    lib.oracle_plane_intersect.argtypes = [C.POINTER(OPlane), C.POINTER(ORay), C.POINTER(cf)]
    lib.oracle_cube_intersect.restype = ci
    lib.oracle_cube_intersect.argtypes = [C.POINTER(OCube), C.POINTER(ORay), C.POINTER(cf)]
    lib.oracle_make_plane.restype = None
    lib.oracle_make_plane.argtypes = [C.POINTER(OPlane)] + [cf] * 6
    lib.oracle_make_cube.restype = None
    lib.oracle_make_cube.argtypes = [C.POINTER(OCube)] + [cf] * 6
    lib.oracle_mesh_from_obj.restype = C.c_void_p
    lib.oracle_mesh_from_obj.argtypes = [C.c_char_p]
    lib.oracle_mesh_free.restype = None
    lib.oracle_mesh_free.argtypes = [C.c_void_p]
    lib.oracle_mesh_counts.restype = ci
    lib.oracle_mesh_counts.argtypes = [C.c_void_p, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    lib.oracle_mesh_triangles.restype = C.POINTER(OTriangle)
    lib.oracle_mesh_triangles.argtypes = [C.c_void_p]
    lib.oracle_mesh_box.restype = ci
    lib.oracle_mesh_box.argtypes = [C.c_void_p, ci, C.POINTER(cf), C.POINTER(cf), C.POINTER(C.POINTER(ci)), C.POINTER(ci)]
    lib.oracle_triangle_intersect.restype = ci
    lib.oracle_triangle_intersect.argtypes = [C.POINTER(OTriangle), C.POINTER(ORay), C.POINTER(cf), C.POINTER(cf), C.POINTER(cf)]
    lib.oracle_msvc_srand.restype = None
    lib.oracle_msvc_srand.argtypes = [C.c_uint]
    lib.oracle_msvc_rand.restype = ci
    lib.oracle_generate_spheres.restype = None
    lib.oracle_generate_spheres.argtypes = [C.POINTER(OSphere), ci, C.c_uint]
    for nm in ("oracle_cosf", "oracle_sinf", "oracle_acosf"):
        getattr(lib, nm).restype = cf
        getattr(lib, nm).argtypes = [cf]
    lib.oracle_atan2f.restype = cf
    lib.oracle_atan2f.argtypes = [cf, cf]
    lib.oracle_uses_libm.restype = ci
    lib.oracle_pack_color.restype = C.c_uint32
    lib.oracle_pack_color.argtypes = [cf, cf, cf]
    _libs[key] = lib
    return lib


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_sprite(planes):
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    h, w = planes[0].shape
    sp = OSprite(_fptr(planes[0]), _fptr(planes[1]), _fptr(planes[2]), w, h)
    sp._keep = planes
    return sp


def render(spheres, n_spheres, texture_planes, sky_planes, sky_box, lights, n_lights, cam, width, height,
           aspect, y0=0, y1=None, off=(0.5, 0.5), nthreads=1, libm=False, want_rgba=True,
           cubes=None, n_cubes=0, planes=None, n_planes=0, mesh=None):
    """Run the oracle on the given inputs. `spheres`/`lights`/`cam`/`sky_box` may be
    the product's ctypes arrays: they are byte-copied into the oracle's own PODs.
    Returns (rgba float32 [rows,W,4], packed uint32 [rows,W], counters dict)."""
    lib = load(libm)
    y1 = height if y1 is None else y1
    osph = (OSphere * max(n_spheres, 1))()
    C.memmove(osph, spheres, 32 * n_spheres)
    olights = (OLight * max(n_lights, 1))()
    C.memmove(olights, lights, 28 * n_lights)
    ocam = OCamera()
    C.memmove(C.byref(ocam), C.byref(cam), 36)
    obox = OSphere()
    C.memmove(C.byref(obox), C.byref(sky_box), 32)
    tex = make_sprite(texture_planes)
    sky = make_sprite(sky_planes)
    ocubes = (OCube * max(n_cubes, 1))()
    if n_cubes:
        C.memmove(ocubes, cubes, 80 * n_cubes)
    oplanes = (OPlane * max(n_planes, 1))()
    if n_planes:
        C.memmove(oplanes, planes, 40 * n_planes)
    fr = OFrame(width, height, aspect, osph, n_spheres, C.pointer(tex), olights, n_lights, ocam,
                C.pointer(obox), C.pointer(sky), y0, y1, off[0], off[1], ocubes, n_cubes, oplanes, n_planes, mesh)
    rows = y1 - y0
    rgba = np.zeros((rows, width, 4), dtype=np.float32) if want_rgba else None
    packed = np.zeros((rows, width), dtype=np.uint32)
    cnt = (C.c_uint64 * 4)()
    rc = lib.oracle_render(C.byref(fr), _fptr(rgba) if want_rgba else None,
                           packed.ctypes.data_as(C.POINTER(C.c_uint32)), cnt, nthreads)
    if rc != 0:
        raise RuntimeError("oracle_render rejected the frame")
    counters = {"primary_tests": cnt[0], "shadow_tests": cnt[1], "hit_pixels": cnt[2], "unshadowed": cnt[3]}
    return rgba, packed, counters


class Mesh:
    """The oracle's restatement of the reference mesh loader + BVH on OBJ text."""

    def __init__(self, obj_text: str, libm: bool = False):
        self.lib = load(libm)
        self.handle = self.lib.oracle_mesh_from_obj(obj_text.encode())
        if not self.handle:
            raise ValueError("OBJ text produced no triangles")
        pc, bc, hn = C.c_int(), C.c_int(), C.c_int()
        self.lib.oracle_mesh_counts(self.handle, C.byref(pc), C.byref(bc), C.byref(hn))
        self.poly_count, self.bvhbox_count, self.has_normals = pc.value, bc.value, bool(hn.value)

    def triangles(self):
        ptr = self.lib.oracle_mesh_triangles(self.handle)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.poly_count, 27)).copy()

    def boxes(self):
        out = []
        for j in range(self.bvhbox_count):
            b, o = (C.c_float * 6)(), (C.c_float * 3)()
            idx, ln = C.POINTER(C.c_int)(), C.c_int()
            self.lib.oracle_mesh_box(self.handle, j, b, o, C.byref(idx), C.byref(ln))
            out.append((list(b), list(o), [idx[i] for i in range(ln.value)]))
        return out
