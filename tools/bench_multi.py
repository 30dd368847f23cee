#!/usr/bin/env python3
"""rt_multi_* (the C++ multi-GPU path) rehearsed on ONE GPU: the frame split into 1, 2, 4, 8 shares that all run on
device 0 (peer-copy transport), so nothing gets faster -- what shows is the overhead of the path itself: more,
smaller launches, the row copies, the scatter kernel, the event plumbing. On 8 real GPUs each share's kernel runs
on its own device."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rt_amd
from _settle import settle

rt = rt_amd.load()
lib = rt.load_library()
out = {}
for (w, h, n) in ((3840, 2160, 1024), (7680, 4320, 4096)):
    ref = rt.Scene.default(n)
    fp = C.POINTER(C.c_float)
    ptr = lambda a: a.ctypes.data_as(fp)
    for shares in (0, 1, 2, 4, 8):   # 0: ONE share through the RCCL transport (24-bit rows, one-rank ncclGather, scatter kernel)
        rccl1 = shares == 0
        shares = max(shares, 1)
        m = C.c_void_p()
        devs = (C.c_int * shares)(*([0] * shares))
        assert lib.rt_multi_create_ex(devs, shares, 1 if rccl1 else 2, C.byref(m)) == 0, lib.rt_last_error()
        lib.rt_multi_set_spheres(m, ref.spheres, n)
        th, tw = ref.texture[0].shape
        lib.rt_multi_set_texture(m, ptr(ref.texture[0]), ptr(ref.texture[1]), ptr(ref.texture[2]), tw, th)
        sh, sw = ref.sky[0].shape
        lib.rt_multi_set_sky(m, C.byref(ref.sky_box), ptr(ref.sky[0]), ptr(ref.sky[1]), ptr(ref.sky[2]), sw, sh)
        lib.rt_multi_set_lights(m, ref.lights, 3)
        fd = ref.frame_desc(w, h)
        assert lib.rt_multi_render(m, C.byref(fd), None) == 0, lib.rt_last_error()
        settle(lambda: lib.rt_multi_render(m, C.byref(fd), None), lambda: lib.rt_multi_sync(m), window=5)
        steps = 50 if n == 1024 else 15
        t0 = time.perf_counter()
        for _ in range(steps):
            lib.rt_multi_render(m, C.byref(fd), None)
        lib.rt_multi_sync(m)
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[f"{w}x{h}_n{n}_" + ("rccl_one_rank" if rccl1 else f"shares{shares}")] = {"ms_per_frame": ms, "Mrays_per_s": w * h / ms / 1e3}
        lib.rt_multi_destroy(m)
print(json.dumps(out, indent=1))
