#!/usr/bin/env python3
"""Host-side cost per frame when the GPU work is small: one rank's share of the C3 frame
(interleaved rows of rank 0 of N) rendered back to back, without any collective. If
ms_per_step stays close to kernel_ms the launch path will not limit multi-GPU scaling."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rt_amd

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
rt = rt_amd.load()
from ray_tracer_engine_amd import distributed as rd  # noqa: E402 (registered by rt_amd.load)
scene = rt.Scene.default(1024, 1)
w, h = 3840, 2160
out = {}
for world in a.world:
    il = (world, 0, 16) if world > 1 else None
    rows = len(rt.interleaved_rows(h, 0, world, 16)) if world > 1 else h
    rgba = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
    packed = torch.zeros((rd.max_interleaved_rows(h, world, 16) if world > 1 else h, w), dtype=torch.int32, device="cuda")
    fd = scene.frame_desc(w, h, pixels=packed.data_ptr(), rgba=rgba.data_ptr(), y0=0, y1=0 if world > 1 else h,
                          spp=1, cull=True, tile=0, interleave=il)
    stream = torch.cuda.current_stream()
    for _ in range(20):
        scene.render_raw(fd, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(a.steps):
        scene.render_raw(fd, stream.cuda_stream)
    e1.record(stream)
    host_issue = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    out[f"rank0_of_{world}"] = {"rows": rows, "gpu_ms_per_frame": e0.elapsed_time(e1) / a.steps,
                                "host_issue_ms_per_frame": host_issue, "wall_ms_per_frame": wall}
print(json.dumps(out, indent=1))
