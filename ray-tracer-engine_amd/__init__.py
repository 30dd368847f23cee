"""ray-tracer-engine_amd -- Python host binding of the MI355X-native ray-tracing
hot path (C ABI: include/rt_engine.h, library: csrc/librt_engine.so).

This package is plumbing around the C ABI: ctypes structures that mirror the
reference's kernel-argument types (/root/reference/kernel.cu:38-40, 225-262,
265-358, 1246-1261; sprite.h:11-47), a `Scene` wrapper over the device-resident
scene, and helpers to build the default scene of the reference
(kernel.cu:1189-1192, 1695-1712). PyTorch is used only for device buffers,
streams and torch.distributed.

There is NO CPU fallback: if the HIP library is missing or no GPU is present the
render calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_ENGINE_LIB") or os.path.join(_HERE, "csrc", "librt_engine.so")   # env: tuning builds only

# torch bundles a HIP runtime with the same soname as /opt/rocm's; it has to be
# in the process first so that the engine binds to that single runtime.
try:  # pragma: no cover - exercised implicitly
    import torch  # noqa: F401
except Exception:  # torch is optional for pure-host helpers
    torch = None

RT_MAX_LIGHTS = 8
RT_MAX_SPP = 16
RT_STATS_COUNT = 24
STAT_NAMES = ("primary_tests", "shadow_tests", "cull_tests", "hit_pixels", "unshadowed",
              "wave_test_slots", "list_entries", "list_overflows",
              "cyc_ray_setup", "cyc_primary_cull", "cyc_primary_tests", "cyc_shade_sky", "cyc_beam_bound",
              "cyc_shadow_cull", "cyc_sample_dirs", "cyc_shadow_tests",
              "clusters", "waves_lt5us", "waves_lt10us", "waves_lt20us", "waves_lt40us", "waves_lt80us", "waves_lt160us",
              "waves_ge160us")


class RtError(RuntimeError):
    pass


# ----------------------------------------------------------------------------
# ctypes mirrors of include/rt_engine.h
# ----------------------------------------------------------------------------
class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Ray(C.Structure):
    _fields_ = [("Org", Vec3), ("Dir", Vec3)]


class Camera(C.Structure):
    _fields_ = [("Org", Vec3), ("Dir", Vec3), ("aspect", C.c_float),
                ("Camyaw", C.c_float), ("Campitch", C.c_float)]


class Light(C.Structure):
    _fields_ = [("pos", Vec3), ("size", C.c_float), ("r", C.c_float), ("g", C.c_float), ("b", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("vptr_slot", C.c_void_p), ("orgin", Vec3), ("reflective", C.c_uint8),
                ("pad_", C.c_uint8 * 3), ("radius", C.c_float), ("tail_pad_", C.c_uint32)]


class Plane(C.Structure):
    _fields_ = [("vptr_slot", C.c_void_p), ("orgin", Vec3), ("reflective", C.c_uint8), ("pad_", C.c_uint8 * 3),
                ("normal", Vec3), ("tail_pad_", C.c_uint32)]


class Cube(C.Structure):
    _fields_ = [("vptr_slot", C.c_void_p), ("orgin", Vec3), ("normals", Vec3 * 3), ("bounds", Vec3 * 2)]


class Vec2(C.Structure):
    _fields_ = [("u", C.c_float), ("v", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("points", Vec3 * 3), ("normal", Vec3), ("vecNormal", Vec3 * 3), ("vt", Vec2 * 3)]


class BvhBox(C.Structure):
    _fields_ = [("bvhbox", C.POINTER(Cube)), ("d_bvhbox", C.POINTER(Cube)), ("indexes", C.POINTER(C.c_int)),
                ("d_indexes", C.POINTER(C.c_int)), ("length", C.c_int)]


class Mesh(C.Structure):
    _fields_ = [("d_tri_arr", C.POINTER(Triangle)), ("h_tri_arr", C.POINTER(Triangle)), ("poly_count", C.c_int),
                ("bvhbox_count", C.c_int), ("bvhLayer_count", C.c_int), ("has_normals", C.c_uint8),
                ("h_box", C.POINTER(BvhBox)), ("d_box", C.POINTER(BvhBox)), ("indexes", C.POINTER(C.c_int))]


class Buffer(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("size", C.c_int)]


class Sprite(C.Structure):
    _fields_ = [("rBuff", C.POINTER(Buffer)), ("gBuff", C.POINTER(Buffer)), ("bBuff", C.POINTER(Buffer)),
                ("width", C.c_int), ("height", C.c_int)]


class Skybox(C.Structure):
    _fields_ = [("box", C.POINTER(Sphere)), ("skyboxTex", C.POINTER(Sprite))]


class Object(C.Structure):
    _fields_ = [("sphere_count", C.c_int), ("plane_count", C.c_int), ("cube_count", C.c_int),
                ("depth", C.c_int), ("s1", C.POINTER(Sphere)), ("d_spheres", C.POINTER(Sphere)),
                ("c1", C.POINTER(Cube)), ("d_cubes", C.POINTER(Cube)), ("planes", C.POINTER(Plane)),
                ("d_planes", C.POINTER(Plane)), ("mesh1", C.POINTER(Mesh)), ("texture", C.POINTER(Sprite)),
                ("mat", C.c_void_p), ("tot_mesh", C.c_void_p), ("meshes", C.c_int)]


class LaunchOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("rgba", C.c_void_p), ("y0", C.c_int), ("y1", C.c_int),
                ("spp", C.c_int), ("sample_base", C.c_int), ("sample_total", C.c_int),
                ("accumulate", C.c_int), ("resolve", C.c_int), ("cull", C.c_int), ("tile", C.c_int),
                ("stats", C.c_void_p), ("force_slow_path", C.c_int), ("profile", C.c_int),
                ("interleave_count", C.c_int), ("interleave_index", C.c_int), ("interleave_rows", C.c_int),
                ("packed24", C.c_void_p), ("table_lds", C.c_int), ("fast", C.c_int)]


class FrameDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int), ("height", C.c_int),
                ("aspect", C.c_float), ("cam", Camera), ("pixels", C.c_void_p), ("opts", LaunchOpts)]


_lib = None


def load_library():
    """dlopen the HIP extension. Raises if it has not been built: the product
    path never substitutes a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C ray-tracer-engine_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fp = C.POINTER(C.c_float)
    sig = {
        "rt_abi_version": (ci, []),
        "rt_last_error": (C.c_char_p, []),
        "rt_device_count": (ci, []),
        "rt_set_soft_errors": (ci, [ci]),
        "rt_check": (None, [ci, C.c_char_p, C.c_char_p, ci]),
        "rt_managed_alloc": (vp, [C.c_size_t]),
        "rt_managed_free": (None, [vp]),
        "rt_launch_raytrace": (ci, [vp, ci, ci, cf, C.POINTER(Object), C.POINTER(Light), ci, Camera,
                                    C.POINTER(Skybox), vp]),
        "rt_launch_raytrace_ex": (ci, [vp, ci, ci, cf, C.POINTER(Object), C.POINTER(Light), ci, Camera,
                                       C.POINTER(Skybox), vp, C.POINTER(LaunchOpts)]),
        "rt_invalidate_textures": (None, []),
        "rt_on_start": (None, []),
        "rt_update": (None, []),
        "rt_config_set_sphere_count": (ci, [ci]),
        "rt_config_set_seed": (ci, [C.c_uint]),
        "rt_config_set_assets": (ci, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "rt_config_camera": (C.POINTER(Camera), []),
        "rt_config_lights": (C.POINTER(Light), [C.POINTER(ci)]),
        "rt_default_aspect": (cf, []),
        "rt_last_frame_ms": (C.c_double, []),
        "rt_offscreen_resize": (ci, [ci, ci]),
        "rt_offscreen_pixels": (C.POINTER(C.c_uint32), []),
        "rt_offscreen_width": (ci, []),
        "rt_offscreen_height": (ci, []),
        "rt_offscreen_write_ppm": (ci, [C.c_char_p]),
        "rt_sphere_init": (None, [C.POINTER(Sphere), cf, cf, cf, cf]),
        "rt_generate_spheres": (ci, [C.POINTER(Sphere), ci, C.c_uint]),
        "rt_mesh_from_obj_text": (C.POINTER(Mesh), [C.c_char_p]),
        "rt_mesh_load_obj": (C.POINTER(Mesh), [C.c_char_p]),
        "rt_mesh_free": (None, [C.POINTER(Mesh)]),
        "rt_scene_set_mesh": (ci, [vp, C.POINTER(Mesh)]),
        "rt_plane_init": (None, [C.POINTER(Plane), cf, cf, cf, cf, cf, cf]),
        "rt_cube_init": (None, [C.POINTER(Cube), cf, cf, cf, cf, cf, cf]),
        "rt_scene_set_planes": (ci, [vp, C.POINTER(Plane), ci]),
        "rt_scene_set_cubes": (ci, [vp, C.POINTER(Cube), ci]),
        "rt_msvc_rand_sequence": (ci, [C.c_uint, C.POINTER(ci), ci]),
        "rt_synth_texture_size": (ci, [ci, C.POINTER(ci), C.POINTER(ci)]),
        "rt_synth_texture": (ci, [ci, fp, fp, fp]),
        "rt_load_ppm": (ci, [C.c_char_p, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), C.POINTER(ci), C.POINTER(ci)]),
        "rt_free_planes": (None, [fp, fp, fp]),
        "rt_sample_offset": (ci, [ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rt_scene_create": (vp, []),
        "rt_scene_destroy": (None, [vp]),
        "rt_scene_set_spheres": (ci, [vp, C.POINTER(Sphere), ci]),
        "rt_scene_set_texture": (ci, [vp, fp, fp, fp, ci, ci]),
        "rt_scene_set_sky": (ci, [vp, C.POINTER(Sphere), fp, fp, fp, ci, ci]),
        "rt_scene_set_lights": (ci, [vp, C.POINTER(Light), ci]),
        "rt_scene_set_tile_order": (ci, [vp, ci]),
        "rt_scene_render": (ci, [vp, C.POINTER(FrameDesc), vp]),
        "rt_graph_capture": (vp, [vp, C.POINTER(FrameDesc), ci, vp, vp]),
        "rt_graph_launch": (ci, [vp, vp]),
        "rt_graph_set_camera": (ci, [vp, C.POINTER(Camera)]),
        "rt_graph_destroy": (None, [vp]),
        "rt_debug_math": (ci, [ci, fp, fp, fp, ci]),
        "rt_debug_intersect": (ci, [C.POINTER(Sphere), C.POINTER(Ray), ci, C.POINTER(ci), fp]),
        "rt_debug_light": (ci, [C.POINTER(Sphere), ci, C.POINTER(Vec3), C.POINTER(Vec3), C.POINTER(Light), ci, fp, fp]),
        "rt_debug_shortcuts": (ci, [ci, C.c_uint, C.c_longlong, C.POINTER(C.c_ulonglong)]),
        "rt_debug_light_prepass": (ci, [C.POINTER(Vec3), C.POINTER(Light), ci, fp, fp, C.POINTER(ci)]),
        "rt_debug_occluder_lists": (ci, [C.POINTER(Sphere), ci, C.POINTER(Light), C.POINTER(ci), fp, C.POINTER(ci), ci]),
        "rt_debug_sphere_beam_slopes": (ci, [C.POINTER(Sphere), ci, C.POINTER(Light), fp, fp]),
        "rt_debug_sphere_beam_slope": (C.c_double, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]),
        "rt_debug_beam_sine": (C.c_double, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rt_debug_occluder_lists_device": (ci, [C.POINTER(Sphere), ci, C.POINTER(Light), C.POINTER(ci), fp, C.POINTER(ci), ci]),
        "rt_debug_occluder_lists_ex": (ci, [C.POINTER(Sphere), ci, C.POINTER(Light), C.POINTER(ci), fp, C.POINTER(ci), ci,
                                            C.POINTER(ci), C.POINTER(ci)]),
        "rt_multi_create": (vp, [ci]),
        "rt_multi_create_ex": (ci, [C.POINTER(ci), ci, ci, C.POINTER(vp)]),
        "rt_multi_destroy": (None, [vp]),
        "rt_multi_device_count": (ci, [vp]),
        "rt_multi_transport": (ci, [vp]),
        "rt_multi_scene": (vp, [vp, ci, C.POINTER(ci)]),
        "rt_multi_set_spheres": (ci, [vp, C.POINTER(Sphere), ci]),
        "rt_multi_set_planes": (ci, [vp, C.POINTER(Plane), ci]),
        "rt_multi_set_cubes": (ci, [vp, C.POINTER(Cube), ci]),
        "rt_multi_set_mesh": (ci, [vp, C.POINTER(Mesh)]),
        "rt_multi_set_texture": (ci, [vp, fp, fp, fp, ci, ci]),
        "rt_multi_set_sky": (ci, [vp, C.POINTER(Sphere), fp, fp, fp, ci, ci]),
        "rt_multi_set_lights": (ci, [vp, C.POINTER(Light), ci]),
        "rt_multi_render": (ci, [vp, C.POINTER(FrameDesc), vp]),
        "rt_multi_sync": (ci, [vp]),
        "rt_multi_stream_wait": (ci, [vp, vp]),
        "rt_multi_note": (C.c_char_p, [vp]),
        "rt_multi_gathers": (C.c_ulonglong, [vp]),
        "rt_config_object": (C.POINTER(Object), []),
        "rt_multi_frame": (vp, [vp]),
        "rt_multi_download": (ci, [vp, vp]),
        "rt_config_set_gpus": (ci, [ci]),
        "rt_assemble_rows24": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RtError(f"{what} failed (status {rc}): {load_library().rt_last_error().decode(errors='replace')}")


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ----------------------------------------------------------------------------
# default scene of the reference (kernel.cu:1189-1192, 1695-1712, 261, 1701)
# ----------------------------------------------------------------------------
def default_camera() -> Camera:
    return Camera(Vec3(4, 3, 10), Vec3(0, 0, 1), 0.0, 180.0, -20.0)


def default_lights():
    arr = (Light * 3)()
    arr[0] = Light(Vec3(20, 20, 20), 20, 1, 0, 0)
    arr[1] = Light(Vec3(0, 20, -20), 20, 0, 0, 1)
    arr[2] = Light(Vec3(0, 20, 0), 20, 0, 1, 0)
    return arr


def default_aspect() -> float:
    return float(load_library().rt_default_aspect())


def generate_spheres(n: int, seed: int = 1):
    arr = (Sphere * max(n, 1))()
    _check(load_library().rt_generate_spheres(arr, n, seed), "rt_generate_spheres")
    return arr


def synth_texture(kind: int):
    """(r, g, b) float32 planes [H, W] of the deterministic stand-in textures."""
    lib = load_library()
    w, h = C.c_int(), C.c_int()
    _check(lib.rt_synth_texture_size(kind, C.byref(w), C.byref(h)), "rt_synth_texture_size")
    planes = [np.empty((h.value, w.value), dtype=np.float32) for _ in range(3)]
    _check(lib.rt_synth_texture(kind, *[_fptr(p) for p in planes]), "rt_synth_texture")
    return planes


def mesh_from_obj_text(text: str):
    """rt_mesh* (reference layout) from OBJ text; raises on failure."""
    lib = load_library()
    m = lib.rt_mesh_from_obj_text(text.encode())
    if not m:
        raise RtError("rt_mesh_from_obj_text: " + lib.rt_last_error().decode(errors="replace"))
    return m


def sky_sphere(size: float = 10000.0) -> Sphere:
    s = Sphere()
    load_library().rt_sphere_init(C.byref(s), 0.0, 0.0, 0.0, size)
    return s


def band_rows(height: int, rank: int, world: int):
    """Contiguous row band [y0, y1) of `rank` out of `world`.
    Bands differ by at most one row; every row belongs to exactly one rank."""
    base, rem = divmod(height, world)
    y0 = rank * base + min(rank, rem)
    return y0, y0 + base + (1 if rank < rem else 0)


def interleaved_rows(height: int, rank: int, world: int, block: int = 16):
    """Global row indices owned by `rank` when row blocks of `block` rows are dealt
    round-robin over `world` ranks (balanced multi-GPU split), in local-row order."""
    rows = []
    k = rank
    while k * block < height:
        rows.extend(range(k * block, min((k + 1) * block, height)))
        k += world
    return rows


class Scene:
    """Device-resident scene (rt_scene). Keeps the host arrays it was built from
    so tests can hand exactly the same inputs to the oracle."""

    def __init__(self):
        self.lib = load_library()
        self.handle = self.lib.rt_scene_create()
        if not self.handle:
            raise RtError("rt_scene_create failed")
        self.spheres = None
        self.n_spheres = 0
        self.texture = None
        self.sky = None
        self.sky_box = None
        self.lights = None
        self.n_lights = 0
        self.planes, self.n_planes = None, 0
        self.cubes, self.n_cubes = None, 0

    def close(self):
        if self.handle:
            self.lib.rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_spheres(self, spheres, n):
        _check(self.lib.rt_scene_set_spheres(self.handle, spheres, n), "rt_scene_set_spheres")
        self.spheres, self.n_spheres = spheres, n

    def set_planes(self, planes, n):
        _check(self.lib.rt_scene_set_planes(self.handle, planes, n), "rt_scene_set_planes")
        self.planes, self.n_planes = planes, n

    def set_cubes(self, cubes, n):
        _check(self.lib.rt_scene_set_cubes(self.handle, cubes, n), "rt_scene_set_cubes")
        self.cubes, self.n_cubes = cubes, n

    def set_mesh(self, mesh):
        _check(self.lib.rt_scene_set_mesh(self.handle, mesh), "rt_scene_set_mesh")
        self.mesh = mesh

    def set_texture(self, planes):
        planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
        h, w = planes[0].shape
        _check(self.lib.rt_scene_set_texture(self.handle, *[_fptr(p) for p in planes], w, h), "rt_scene_set_texture")
        self.texture = planes

    def set_sky(self, box: Sphere, planes):
        planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
        h, w = planes[0].shape
        _check(self.lib.rt_scene_set_sky(self.handle, C.byref(box), *[_fptr(p) for p in planes], w, h), "rt_scene_set_sky")
        self.sky, self.sky_box = planes, box

    def set_lights(self, lights, n):
        _check(self.lib.rt_scene_set_lights(self.handle, lights, n), "rt_scene_set_lights")
        self.lights, self.n_lights = lights, n

    def set_tile_order(self, mode: int):
        """1 (default): launches start their longest tiles first (durations of earlier frames); 0: grid order."""
        _check(self.lib.rt_scene_set_tile_order(self.handle, mode), "rt_scene_set_tile_order")

    @classmethod
    def default(cls, n_spheres: int = 1024, seed: int = 1) -> "Scene":
        s = cls()
        s.set_spheres(generate_spheres(n_spheres, seed), n_spheres)
        s.set_texture(synth_texture(0))
        s.set_sky(sky_sphere(), synth_texture(1))
        s.set_lights(default_lights(), 3)
        return s

    def frame_desc(self, width, height, *, pixels=0, rgba=0, cam=None, aspect=None, y0=0, y1=0, spp=1,
                   sample_base=0, sample_total=0, accumulate=False, resolve=0, cull=True, tile=0,
                   stats=0, force_slow=False, profile=False, interleave=None, packed24=0, table_lds=False, fast=False) -> FrameDesc:
        fd = FrameDesc()
        fd.struct_size = C.sizeof(FrameDesc)
        fd.width, fd.height = width, height
        fd.aspect = default_aspect() if aspect is None else aspect
        fd.cam = cam if cam is not None else default_camera()
        fd.pixels = pixels
        o = fd.opts
        o.struct_size = C.sizeof(LaunchOpts)
        o.rgba = rgba
        o.y0, o.y1 = y0, y1
        o.spp, o.sample_base, o.sample_total = spp, sample_base, sample_total
        o.accumulate = 1 if accumulate else 0
        o.resolve = resolve
        o.cull = 1 if cull else 0
        o.tile = tile
        o.stats = stats
        o.force_slow_path = 1 if force_slow else 0
        o.profile = 1 if profile else 0
        if interleave is not None:          # (count, index, block_rows)
            o.interleave_count, o.interleave_index, o.interleave_rows = interleave
        o.packed24 = packed24
        o.table_lds = 1 if table_lds else 0
        o.fast = 1 if fast else 0
        return fd

    def render_raw(self, fd: FrameDesc, stream=0):
        _check(self.lib.rt_scene_render(self.handle, C.byref(fd), stream), "rt_scene_render")

    def render(self, width, height, *, y0=0, y1=0, want_rgba=True, want_stats=False, want_packed24=False, stream=None, **kw):
        """Render rows [y0,y1) into fresh torch CUDA tensors and return
        {'packed': int32 [rows, W], 'rgba': float32 [rows, W, 4], 'stats': dict}."""
        import torch
        if not torch.cuda.is_available():
            raise RtError("no GPU visible: the ray-tracing path has no CPU fallback")
        rows = (y1 if y1 else height) - y0
        if kw.get("interleave") is not None:
            cnt, idx, blk = kw["interleave"]
            rows = len(interleaved_rows(rows, idx, cnt, blk or 16))   # blocks are dealt from the band's first row
        packed = torch.empty((rows, width), dtype=torch.int32, device="cuda")
        rgba = torch.empty((rows, width, 4), dtype=torch.float32, device="cuda") if want_rgba else None
        stats = torch.zeros(RT_STATS_COUNT, dtype=torch.int64, device="cuda") if want_stats else None
        st = torch.cuda.current_stream() if stream is None else stream
        p24 = torch.zeros((rows, width * 3 // 4), dtype=torch.int32, device="cuda") if want_packed24 else None
        fd = self.frame_desc(width, height, pixels=packed.data_ptr(), rgba=rgba.data_ptr() if want_rgba else 0,
                             y0=y0, y1=y1, stats=stats.data_ptr() if want_stats else 0,
                             packed24=p24.data_ptr() if want_packed24 else 0, **kw)
        self.render_raw(fd, st.cuda_stream)
        out = {"packed": packed, "rgba": rgba}
        if want_packed24:
            out["packed24"] = p24     # [rows, 3*width/4] int32: bytes B,G,R per pixel (rt_launch_opts.packed24)
        if want_stats:
            out["stats"] = dict(zip(STAT_NAMES, stats.cpu().tolist()))
        return out
