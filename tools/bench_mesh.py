#!/usr/bin/env python3
"""Timing of a scene with a triangle mesh on one MI355X."""
import json, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import torch
import rt_amd
import meshes
from _settle import settle

rt = rt_amd.load()
out = {}
for (w, h, n, lat, lon) in ((1920, 1080, 256, 24, 40), (3840, 2160, 1024, 48, 80)):
    scene = rt.Scene.default(n)
    txt = meshes.uv_sphere_obj(cx=4.0, cy=1.5, cz=6.0, r=2.0, n_lat=lat, n_lon=lon)
    m = rt.mesh_from_obj_text(txt)
    scene.set_mesh(m)
    rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    pk = torch.empty((h, w), dtype=torch.int32, device="cuda")
    fd = scene.frame_desc(w, h, pixels=pk.data_ptr(), rgba=rgba.data_ptr())
    st = torch.cuda.current_stream()
    settle(lambda: scene.render_raw(fd, st.cuda_stream), torch.cuda.synchronize, window=5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    it = 20
    for _ in range(it):
        scene.render_raw(fd, st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    out[f"{w}x{h}_n{n}_mesh{m.contents.poly_count}tris_{m.contents.bvhbox_count}leaves"] = {
        "ms": ms, "Mrays_per_s": w * h / ms / 1e3}
print(json.dumps(out, indent=1))
