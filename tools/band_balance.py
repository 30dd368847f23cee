#!/usr/bin/env python3
"""Kernel time of each rank's row band when the C3 frame is split N ways
(contiguous bands), measured back to back on one GPU."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rt_amd
rt = rt_amd.load()
w, h, n = 3840, 2160, 1024
scene = rt.Scene.default(n)
res = {}
for world in (1, 2, 4, 8):
    times = []
    for r in range(world):
        y0, y1 = rt.band_rows(h, r, world)
        rows = y1 - y0
        rgba = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
        pk = torch.empty((rows, w), dtype=torch.int32, device="cuda")
        fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), rgba=rgba.data_ptr(), y0=y0, y1=y1)
        st = torch.cuda.current_stream()
        for _ in range(3):
            scene.render_raw(fd, st.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            scene.render_raw(fd, st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 20)
    res[world] = [round(t, 3) for t in times]
    print(world, "bands ms:", res[world], "max", max(times), "sum", round(sum(times), 3), "ideal", round(sum(times) / world, 3))
