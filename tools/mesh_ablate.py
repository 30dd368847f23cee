#!/usr/bin/env python3
"""Where the time of a mesh frame goes: the 4K scene of tools/bench_mesh.py rendered by a TUNING build
(RT_ENGINE_LIB=.../librt_engine_tuning.so) with parts of the kernel skipped (RT_ABLATE bits, read once per
scene; the output is wrong when set -- timing only), and once without the mesh."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import rt_amd
from _settle import settle
import meshes

rt = rt_amd.load()
W, H, N = 3840, 2160, 1024


def run(ablate, with_mesh=True, stats=False):
    os.environ["RT_ABLATE"] = str(ablate)
    scene = rt.Scene.default(N)
    if with_mesh:
        scene.set_mesh(rt.mesh_from_obj_text(meshes.uv_sphere_obj(cx=4.0, cy=1.5, cz=6.0, r=2.0, n_lat=48, n_lon=80)))
    rgba = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    pk = torch.empty((H, W), dtype=torch.int32, device="cuda")
    fd = scene.frame_desc(W, H, pixels=pk.data_ptr(), rgba=rgba.data_ptr())
    st = torch.cuda.current_stream()
    settle(lambda: scene.render_raw(fd, st.cuda_stream), torch.cuda.synchronize, window=5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        scene.render_raw(fd, st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    res = {"ms": round(e0.elapsed_time(e1) / 20, 4)}
    if stats:
        res["stats"] = scene.render(W, H, want_stats=True)["stats"]
    return res


out = {"no_mesh": run(0, False)}
for name, bits in (("all", 0), ("no_lights", 16), ("no_lights_no_leaf_walk", 16 + 2048), ("no_lights_no_triangle_tests", 16 + 1024),
                   ("no_sphere_shadow_tests", 1), ("no_sample_construction", 2), 
                   ("no_leaf_walk", 2048), ("no_triangle_tests", 1024), ("no_shadow_mesh_loop", 8192),
                   ("no_shadow_triangle_tests", 16384), ("no_light_box_lists", 32768),
                   ("no_occluder_lists", 65536), ("beam_slope_per_tile_not_per_sphere", 524288)):
    out[name] = run(bits)
print(json.dumps(out, indent=1))
