#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs on one MI355X (bench.py itself
stays on the headline C3 config): C2, C3, C4 (4-spp, hipGraph replay and
in-kernel), one rank's share of C5, and the update() path with its D2H copy."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd
from _settle import settle

rt = rt_amd.load()
lib = rt.load_library()
out = {}


def time_render(scene, w, h, iters=30, **kw):
    rows = len(rt.interleaved_rows(h, kw["interleave"][1], kw["interleave"][0], 16)) if kw.get("interleave") else h
    rgba = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
    pk = torch.empty((rows, w), dtype=torch.int32, device="cuda")
    fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), rgba=rgba.data_ptr(), **kw)
    st = torch.cuda.current_stream()
    settle(lambda: scene.render_raw(fd, st.cuda_stream), torch.cuda.synchronize)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        scene.render_raw(fd, st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, rows


s256 = rt.Scene.default(256)
ms, _ = time_render(s256, 1920, 1080)
out["C2_1920x1080_n256_1spp"] = {"ms": ms, "Mrays_per_s": 1920 * 1080 / ms / 1e3, "fps": 1e3 / ms}

s1024 = rt.Scene.default(1024)
ms, _ = time_render(s1024, 3840, 2160)
out["C3_3840x2160_n1024_1spp"] = {"ms": ms, "Mrays_per_s": 3840 * 2160 / ms / 1e3, "fps": 1e3 / ms}

ms, _ = time_render(s1024, 3840, 2160, spp=4, iters=10)
out["C4_4spp_in_kernel"] = {"ms": ms, "Mrays_per_s": 4 * 3840 * 2160 / ms / 1e3, "fps": 1e3 / ms}

# C4 as BASELINE words it: 4-spp accumulate under a hipGraph-captured frame loop
w, h = 3840, 2160
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
pk = torch.zeros((h, w), dtype=torch.int32, device="cuda")
host = torch.zeros((h, w), dtype=torch.int32).pin_memory()
stream = torch.cuda.Stream()
fd = s1024.frame_desc(w, h, pixels=pk.data_ptr(), rgba=acc.data_ptr())
for passes, with_copy in ((4, False), (4, True), (-4, False)):
    g = lib.rt_graph_capture(s1024.handle, C.byref(fd), passes, host.data_ptr() if with_copy else None, stream.cuda_stream)
    assert g, lib.rt_last_error()
    settle(lambda: lib.rt_graph_launch(g, stream.cuda_stream), stream.synchronize, window=5)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        lib.rt_graph_launch(g, stream.cuda_stream)
    stream.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    out["C4_4spp_hipgraph" + ("_with_d2h" if with_copy else "") + ("_four_progressive_nodes" if passes < 0 else "")] = {
        "ms": ms, "Mrays_per_s": 4 * w * h / ms / 1e3, "fps": 1e3 / ms}
    if not with_copy and passes > 0:
        # the camera nudged before every replay (kernel.cu:1727): node parameters replaced in the instantiated
        # graph, the graph's eye-cone table rebuilt by its own build node -- no re-capture, no synchronisation
        cam = rt.default_camera()
        t0 = time.perf_counter()
        for k in range(n):
            cam.Org.z = 10.0 + 0.1 * (k % 4)
            lib.rt_graph_set_camera(g, C.byref(cam))
            lib.rt_graph_launch(g, stream.cuda_stream)
        stream.synchronize()
        ms_mv = (time.perf_counter() - t0) / n * 1e3
        out["C4_4spp_hipgraph_moving_camera"] = {"ms": ms_mv, "Mrays_per_s": 4 * w * h / ms_mv / 1e3, "fps": 1e3 / ms_mv}
        lib.rt_graph_set_camera(g, C.byref(rt.default_camera()))
    lib.rt_graph_destroy(g)

s4096 = rt.Scene.default(4096)
ms, rows = time_render(s4096, 7680, 4320, iters=5, interleave=(8, 3, 16))
out["C5_7680x4320_n4096_rank3of8"] = {"ms": ms, "rows": rows, "Mrays_per_s_this_rank": 7680 * rows / ms / 1e3,
                                      "note": "camera is enclosed by a sphere at N=4096 (SURVEY F5): every pixel is a negative-t hit"}
ms, _ = time_render(s4096, 7680, 4320, iters=3)
out["C5_7680x4320_n4096_whole_frame_on_1_gpu"] = {"ms": ms, "Mrays_per_s": 7680 * 4320 / ms / 1e3, "fps": 1e3 / ms}

# C5 with the extent scaled so that the sphere density matches the 1024-sphere case
# (at N=4096 the generator encloses the camera in a sphere)
scale = (4096 / 1024) ** (1.0 / 3.0)
sph = rt.generate_spheres(4096, 1)
import numpy as np
for i in range(4096):
    sph[i].orgin.x = float(np.float32(sph[i].orgin.x) * np.float32(scale))
    sph[i].orgin.y = float(np.float32(sph[i].orgin.y) * np.float32(scale))
    sph[i].orgin.z = float(np.float32(sph[i].orgin.z) * np.float32(scale))
s4096s = rt.Scene.default(8)
s4096s.set_spheres(sph, 4096)
ms, _ = time_render(s4096s, 7680, 4320, iters=3)
st = s4096s.render(960, 540, want_stats=True)["stats"]
out["C5_scaled_extent_whole_frame_on_1_gpu"] = {"ms": ms, "Mrays_per_s": 7680 * 4320 / ms / 1e3, "fps": 1e3 / ms,
                                                "hit_fraction_at_960x540": st["hit_pixels"] / (960 * 540),
                                                "note": "sphere centres scaled by 4^(1/3): same density as C3"}

# update(): kernel + D2H into the offscreen window (what the reference's boundary requires)
lib.rt_config_set_sphere_count(1024)
lib.rt_on_start()
lib.rt_offscreen_resize(3840, 2160)
settle(lib.rt_update, lambda: None, window=5)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    lib.rt_update()
ms = (time.perf_counter() - t0) / n * 1e3
out["update_3840x2160_n1024_end_to_end"] = {"ms": ms, "fps": 1e3 / ms, "kernel_ms_last": lib.rt_last_frame_ms(),
                                            "note": "kernel + 33 MB D2H over PCIe + setPixelBuff memcpy"}
print(json.dumps(out, indent=1))
