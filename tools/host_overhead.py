#!/usr/bin/env python3
"""Host-side cost per frame when the GPU work is small: one rank's share of the C3 frame
(interleaved rows of rank 0 of N) rendered back to back, without any collective. If
ms_per_step stays close to kernel_ms the launch path will not limit multi-GPU scaling."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rt_amd
from _settle import settle

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--streams", type=int, default=1, help="frames alternate between this many streams (own output buffers)")
a = ap.parse_args()
rt = rt_amd.load()
from ray_tracer_engine_amd import distributed as rd  # noqa: E402 (registered by rt_amd.load)
scene = rt.Scene.default(1024, 1)
w, h = 3840, 2160
out = {}
for world in a.world:
    il = (world, 0, 16) if world > 1 else None
    rows = len(rt.interleaved_rows(h, 0, world, 16)) if world > 1 else h
    ns = a.streams
    streams = [torch.cuda.Stream() for _ in range(ns)]
    rgbas = [torch.empty((rows, w, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
    packeds = [torch.zeros((rd.max_interleaved_rows(h, world, 16) if world > 1 else h, w), dtype=torch.int32, device="cuda")
               for _ in range(ns)]
    fds = [scene.frame_desc(w, h, pixels=packeds[k].data_ptr(), rgba=rgbas[k].data_ptr(), y0=0, y1=0 if world > 1 else h,
                            spp=1, cull=True, tile=0, interleave=il) for k in range(ns)]
    kk = [0]

    def one():
        scene.render_raw(fds[kk[0] % ns], streams[kk[0] % ns].cuda_stream)
        kk[0] += 1
    settle(one, torch.cuda.synchronize, window=20)
    t0 = time.perf_counter()
    for k in range(a.steps):
        scene.render_raw(fds[k % ns], streams[k % ns].cuda_stream)
    host_issue = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    out[f"rank0_of_{world}"] = {"rows": rows, "host_issue_ms_per_frame": host_issue, "wall_ms_per_frame": wall}
print(json.dumps(out, indent=1))
