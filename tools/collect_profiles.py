#!/usr/bin/env python3
"""Copy the summaries of a tools/final_pass.sh run (gpurun_out/<tag>/) into profiles/<name>_*: what DESIGN.md quotes."""
import json, os, shutil, subprocess, sys
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
line = [l for l in open(os.path.join(src, "bench.json")).read().strip().splitlines() if l.startswith("{")][-1]
json.dump(json.loads(line), open(f"profiles/{name}_bench_c3_n1.json", "w"), indent=1)
subprocess.check_call([sys.executable, "tools/summarize_profile.py", tag, f"{name}_c3_frame_kernel"], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, "tools/summarize_profile.py", tag + "_tablds", f"{name}_c3_table_lds"], stdout=subprocess.DEVNULL)
for f in ("other_configs", "mesh_scenes", "fast_mode", "host_overhead_single_stream", "host_overhead_two_streams",
          "multi_one_gpu_rehearsal", "build_and_first_launch", "bench_driver_flags", "bench_n2_gloo_one_gpu"):
    s = open(os.path.join(src, f + ".json")).read()
    open(f"profiles/{name}_{f}.json", "w").write(s[s.index("{"):])   # (RCCL prints a banner to stdout)
shutil.copy(os.path.join(src, "ablation_pmc.txt"), f"profiles/{name}_ablation_pmc.txt")
if os.path.exists(os.path.join(src, "pmc_classes.txt")):
    shutil.copy(os.path.join(src, "pmc_classes.txt"), f"profiles/{name}_pmc_instruction_classes.txt")
abl = json.load(open(os.path.join(src, "mesh_ablate.json")))
json.dump({"scene": "3840x2160, 1024 spheres + uv-sphere mesh of 7520 triangles / 756 leaves (tools/bench_mesh.py)",
           "ms_with_parts_skipped_tuning_build": {k: v["ms"] for k, v in abl.items()},
           "work_counters": json.load(open(os.path.join(src, "mesh_stats.json")))},
          open(f"profiles/{name}_mesh_ablation.json", "w"), indent=1)
if os.path.exists(os.path.join(src, "tile_order.json")):
    s = open(os.path.join(src, "tile_order.json")).read()
    open(f"profiles/{name}_tile_order.json", "w").write(s[s.index("{"):])
if os.path.exists(os.path.join(src, "bench_tile_order0.json")):
    def legs(d):
        return {"ms_per_step": d["ms_per_step"], "grid_order_leg": d["grid_order"]["ms_per_step"],
                "moving_camera": d["moving_camera"]["ms_per_step"], "same_positions_held": d["moving_camera"]["same_positions_held_ms"],
                "pipelined": d["pipelined"]["ms_per_step"], "fast_mode": d["fast_mode"]["ms_per_step"]}
    z = json.loads([l for l in open(os.path.join(src, "bench_tile_order0.json")).read().strip().splitlines() if l.startswith("{")][-1])
    json.dump({"what": "python bench.py legs (ms per frame) with the library's launch order (tile_order 1: the bench line of "
                       f"profiles/{name}_bench_c3_n1.json) and with grid order (python bench.py --no-cpu-baseline --tile-order 0), "
                       "same box, same run of tools/final_pass.sh",
               "tile_order_1": legs(json.loads(line)), "tile_order_0": legs(z),
               "note": "ms_per_step is the serial loop with HIP events around every fourth launch; the grid_order leg is the same "
                       "loop without events, always in grid order"}, open(f"profiles/{name}_bench_tile_order_ab.json", "w"), indent=1)
print("profiles/%s_* written" % name)
