"""Of the tile-lights whose ten samples are walked, how many have a lane in a penumbra at all (work counters of the
stats kernel, slots 17-20). At C3: 67 % have none -- every lit lane ends fully shadowed or fully lit -- yet a lane-by-lane
cone-versus-sphere classification decides only 7 % of them (measured, reverted): they are shadowed by UNIONS of spheres."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rt_amd
rt = rt_amd.load()
sc = rt.Scene.default(1024)
st = sc.render(3840, 2160, want_stats=True)["stats"]
names = list(st)
v = [st[k] for k in names]
print("walks", v[17], "no_penumbra", v[18], "penumbra lanes", v[19], "lit lanes in walks", v[20])
print("fraction of walked lights without any penumbra lane:", v[18] / v[17], " mean penumbra lanes per walk:", v[19] / v[17], " of lit lanes", v[20] / v[17])
print("walked lights whose lit lanes are ALL fully shadowed:", v[21], " ALL fully lit:", v[22])
print("shadow list length histogram (<=1,<=2,<=4,<=8,<=16,<=cap, all clear, full occluder):", v[8:16])
need, walks_need, single = v[23] & 0xffffff, (v[23] >> 24) & 0xfffff, v[23] >> 44
print("pre-pass: walks that needed the exact chain:", walks_need, " samples evaluated exactly (wave level):", need,
      " = per such walk", need / max(1, walks_need), " walks whose every sample has ONE entry that all lit pixels hit for sure:", single)
