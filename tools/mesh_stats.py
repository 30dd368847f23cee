#!/usr/bin/env python3
"""Work counters of a mesh frame (the 4K scene of tools/bench_mesh.py): how many leaves a tile lists, visits and how
many triangle tests it executes, for the primary and for the shadow rays (wave-level counts, stats kernel slots 18-23
of a MESH launch)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import rt_amd
import meshes

rt = rt_amd.load()
W, H, N = 3840, 2160, 1024
scene = rt.Scene.default(N)
m = rt.mesh_from_obj_text(meshes.uv_sphere_obj(cx=4.0, cy=1.5, cz=6.0, r=2.0, n_lat=48, n_lon=80))
scene.set_mesh(m)
st = scene.render(W, H, want_stats=True)["stats"]
v = list(st.values())
tiles = (W // 8) * (H // 8)
out = {"tiles": tiles, "triangles": m.contents.poly_count, "leaves": m.contents.bvhbox_count,
       "walked_tile_lights": v[17],
       "primary": {"leaves_listed": v[18], "leaves_with_a_lane_inside_the_box": v[19], "triangle_tests_wave_level": v[20]},
       "shadow": {"leaves_listed_by_walked_tile_lights": v[21], "slab_tests_wave_level": v[22], "triangle_tests_wave_level": v[23]},
       "hit_pixels": st["hit_pixels"], "shadow_sphere_tests_per_lane": st["shadow_tests"], "list_entries": st["list_entries"]}
print(json.dumps(out, indent=1))
