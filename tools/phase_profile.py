#!/usr/bin/env python3
"""Per-phase cycle shares of the frame kernel (diagnostic build: STATS=2, s_memtime
stamps). Read the SHARES, not the run time (stamps serialise the phases)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import rt_amd

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--spheres", type=int, default=1024)
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--no-cull", action="store_true")
ap.add_argument("--interleave", type=int, nargs=2, default=None, metavar=("COUNT", "INDEX"))
a = ap.parse_args()
rt = rt_amd.load()
scene = rt.Scene.default(a.spheres)
il = (a.interleave[0], a.interleave[1], 16) if a.interleave else None
from _settle import settle
settle(lambda: scene.render(a.width, a.height, cull=not a.no_cull, tile=a.tile, interleave=il), torch.cuda.synchronize)   # clocks up first
out = scene.render(a.width, a.height, want_stats=True, profile=True, cull=not a.no_cull, tile=a.tile, interleave=il)
torch.cuda.synchronize()
st = out["stats"]
cyc = {k: v for k, v in st.items() if k.startswith("cyc_")}
tot = sum(cyc.values()) or 1
waves = sum(v for k, v in st.items() if k.startswith('waves_')) or (a.width * a.height + 63) // 64
print(json.dumps({"waves": waves, "cycles_per_wave": tot / waves,
                  "share": {k: round(v / tot, 4) for k, v in cyc.items()},
                  "cycles_per_wave_by_phase": {k: round(v / waves, 1) for k, v in cyc.items()},
                  "wave_duration_histogram_cycles": {k: v for k, v in st.items() if k.startswith("waves_")}}, indent=1))
cnt = scene.render(a.width, a.height, want_stats=True, cull=not a.no_cull, tile=a.tile)["stats"]
print(json.dumps({k: v for k, v in cnt.items() if not k.startswith("cyc_")}))
names = ["<=1", "<=2", "<=4", "<=8", "<=16", "<=cap", "all_clear_skips", "full_occluder_skips"]
print("shadow list length histogram:", dict(zip(names, [cnt[k] for k in list(cnt)[8:16]])))
print("mean distinct hit spheres per hit wave:", cnt["clusters"] / max(1, waves))
