"""The C-ABI library: loads, exports every symbol include/rt_engine.h declares,
mirrors the reference's struct layouts, and fails loudly (never silently falls
back to a CPU path) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "rt_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_are_exported(rt):
    lib = rt.load_library()
    names = _declared_functions()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_reference_cxx_symbols_are_exported(rt):
    # kernel.cuh:3-4 and window.h:7-16 keep C++ linkage in the reference
    so = rt.LIB_PATH
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", so], capture_output=True, text=True, check=True).stdout
    for sym in ("onStart()", "update()", "getScreenWidth()", "getScreenHeight()", "setPixelBuff(unsigned int*)",
                "drawPixel(int, int, int)", "Clear_Screen(unsigned int)", "make_inbound(int, int, int)",
                "buffer::buffer(float*, int)", "sprite::sprite("):
        assert sym in out, sym
    # the window functions must be weak so an application's window.cpp wins
    weak = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert re.search(r" W _Z12setPixelBuffPj", weak)


def test_struct_layouts(rt):
    assert C.sizeof(rt.Vec3) == 12 and C.sizeof(rt.Ray) == 24
    assert C.sizeof(rt.Camera) == 36 and C.sizeof(rt.Light) == 28
    assert C.sizeof(rt.Sphere) == 32 and rt.Sphere.orgin.offset == 8 and rt.Sphere.radius.offset == 24
    assert C.sizeof(rt.Buffer) == 16 and C.sizeof(rt.Sprite) == 32 and C.sizeof(rt.Skybox) == 16
    assert C.sizeof(rt.Plane) == 40 and rt.Plane.normal.offset == 24 and C.sizeof(rt.Cube) == 80 and rt.Cube.bounds.offset == 56
    assert C.sizeof(rt.Object) == 104
    assert rt.Object.d_spheres.offset == 24 and rt.Object.texture.offset == 72


def test_abi_version_and_aspect(rt):
    lib = rt.load_library()
    assert lib.rt_abi_version() == 1
    import numpy as np
    assert np.float32(lib.rt_default_aspect()) == np.float32(0.9999537)


def test_sample_offsets(rt):
    lib = rt.load_library()
    ox, oy = C.c_double(), C.c_double()
    assert lib.rt_sample_offset(0, 1, C.byref(ox), C.byref(oy)) == 0
    assert (ox.value, oy.value) == (0.5, 0.5)          # the reference's pixel centre
    got = []
    for k in range(4):
        assert lib.rt_sample_offset(k, 4, C.byref(ox), C.byref(oy)) == 0
        got.append((ox.value, oy.value))
    assert got == [(0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75)]
    assert lib.rt_sample_offset(4, 4, C.byref(ox), C.byref(oy)) != 0


def test_argument_validation_without_gpu_work(rt):
    lib = rt.load_library()
    obj = rt.Object()
    sky = rt.Skybox()
    lights = rt.default_lights()
    obj.cube_count = 1                      # a count without a list is invalid
    rc = lib.rt_launch_raytrace(None, 64, 64, 1.0, C.byref(obj), lights, 3, rt.default_camera(), C.byref(sky), None)
    assert rc == 1
    obj.cube_count = 0
    rc = lib.rt_launch_raytrace(None, 64, 64, 1.0, C.byref(obj), lights, 3, rt.default_camera(), C.byref(sky), None)
    assert rc == 1        # skybox missing
    assert lib.rt_launch_raytrace(None, 64, 64, 1.0, None, lights, 3, rt.default_camera(), C.byref(sky), None) == 1
    assert lib.rt_offscreen_resize(0, 10) != 0
    assert lib.rt_offscreen_resize(16, 8) == 0 and lib.rt_offscreen_width() == 16 and lib.rt_offscreen_height() == 8


def test_no_cpu_fallback(rt):
    """Without a GPU every render path must report an error; nothing may produce
    pixels on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked tests")
    lib = rt.load_library()
    assert lib.rt_device_count() == 0
    s = rt.Scene()
    with pytest.raises(rt.RtError):
        s.set_spheres(rt.generate_spheres(8), 8)       # needs device memory
    with pytest.raises(rt.RtError):
        s.render(64, 64)


def test_window_helpers_behave_as_the_reference_window(rt):
    """window.h:7-16 beyond the three functions the frame driver calls: what an application that
    links the offscreen window observes is what window.cpp:95-129 does -- Set_Background writes
    word(x, y) = y*x/(x+1) in int arithmetic, Clear_Screen one colour, drawPixel a clamped store,
    getBuffSize the size of the buffer POINTER member (sizeof(render.buffmemory))."""
    import numpy as np
    lib = rt.load_library()
    w, h = 37, 11
    assert lib.rt_offscreen_resize(w, h) == 0
    px = lambda: np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(h, w)).copy()
    getattr(lib, "_Z14Set_Backgroundv")()
    yy, xx = np.mgrid[0:h, 0:w]
    assert np.array_equal(px(), (yy * xx // (xx + 1)).astype(np.uint32))
    clear = getattr(lib, "_Z12Clear_Screenj")
    clear.argtypes = [C.c_uint]
    clear(0x00123456)
    assert (px() == 0x00123456).all()
    draw = getattr(lib, "_Z9drawPixeliii")
    draw.argtypes = [C.c_int, C.c_int, C.c_int]
    draw(5, 3, 0xABCDEF)
    draw(-7, 400, 0x010203)                       # clamped to (0, h-1)
    got = px()
    assert got[3, 5] == 0xABCDEF and got[h - 1, 0] == 0x010203 and (got != 0x00123456).sum() == 2
    size = getattr(lib, "_Z11getBuffSizev")
    size.restype = C.c_int
    assert size() == C.sizeof(C.c_void_p)
    inb = getattr(lib, "_Z12make_inboundiii")
    inb.restype = C.c_int
    inb.argtypes = [C.c_int] * 3
    assert (inb(0, 9, -3), inb(0, 9, 4), inb(0, 9, 12)) == (0, 4, 9)
