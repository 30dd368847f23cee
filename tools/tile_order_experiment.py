#!/usr/bin/env python3
"""How much of a frame's time is the drain at the end of the launch, and what starting the expensive tiles first buys:
C3 rendered one frame at a time with rt_scene_set_tile_order 0 (grid order) and 1 (longest tiles first, from the wave
durations the kernel records), for the whole frame and for one rank's share of 1/2, 1/4, 1/8 of it."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rt_amd
from _settle import settle

rt = rt_amd.load()
lib = rt.load_library()
W, H, N = 3840, 2160, 1024
out = {}
for world in (1, 2, 4, 8):
    scene = rt.Scene.default(N)
    il = (world, 0, 16) if world > 1 else None
    rows = len(rt.interleaved_rows(H, 0, world, 16)) if world > 1 else H
    gx, gy = (W + 7) // 8, (rows + 7) // 8
    rgba = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
    pk = torch.zeros((rows, W), dtype=torch.int32, device="cuda")
    fd = scene.frame_desc(W, H, pixels=pk.data_ptr(), rgba=rgba.data_ptr(), interleave=il)
    st = torch.cuda.current_stream()

    def timed(steps=200):
        settle(lambda: scene.render_raw(fd, st.cuda_stream), torch.cuda.synchronize)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            scene.render_raw(fd, st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    scene.set_tile_order(0)
    base = timed()
    ref = pk.clone()
    scene.set_tile_order(1)
    pk.zero_()
    lpt = timed()
    same = bool(torch.equal(pk, ref))
    out[f"rank0_of_{world}"] = {"tiles": gx * gy, "ms_grid_order": base, "ms_longest_first": lpt, "frame_identical": same}
print(json.dumps(out, indent=1))
