#!/usr/bin/env python3
"""Cost of bringing the library up in a fresh process: dlopen of librt_engine.so, the first
frame (HIP loads the code object with every instantiation of the frame kernel; tables are built and uploaded),
the second frame, and a steady-state frame. Prints one JSON object; the build time of the library (wall / CPU,
`make -j8` of ray-tracer-engine_amd/csrc in the build container) is passed in by the caller."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--build-wall-s", type=float, default=None)
ap.add_argument("--build-cpu-s", type=float, default=None)
a = ap.parse_args()
t0 = time.perf_counter()
import torch
torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t_torch = time.perf_counter() - t0
import rt_amd
rt = rt_amd.load()
t0 = time.perf_counter()
lib = rt.load_library()
t_dlopen = time.perf_counter() - t0
so = rt.LIB_PATH
t0 = time.perf_counter()
scene = rt.Scene.default(1024)
t_scene = time.perf_counter() - t0
times = []
for k in range(6):
    t0 = time.perf_counter()
    scene.render(3840, 2160, want_rgba=False)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"library": os.path.basename(so), "library_bytes": os.path.getsize(so),
                  "torch_cuda_init_s": t_torch, "dlopen_ms": t_dlopen * 1e3, "scene_upload_ms": t_scene * 1e3,
                  "first_frame_ms": times[0], "second_frame_ms": times[1], "later_frames_ms": times[2:],
                  "build": {"wall_s": a.build_wall_s, "cpu_s": a.build_cpu_s, "how": "make -j8, four translation units of the frame kernel"},
                  "note": "C3 frame (3840x2160, 1024 spheres) incl. allocation of its output tensor and a host synchronise; the first "
                          "frame carries the code-object load, the eye-cone / light-column table builds and the raygen tables"}))
