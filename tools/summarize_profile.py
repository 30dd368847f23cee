#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh run (gpurun_out/prof_<tag>/) into the files
that are committed under profiles/: the rocprofv3 --kernel-trace --stats summary
as-is, and the per-dispatch PMC averages of the frame kernel with the derived
figures DESIGN.md / bench.py quote (HBM bytes with the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md, VALU issue utilisation, mean occupancy)."""
import re, csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join("profiles", f"{name}_kernel_stats.csv"))
rows = list(csv.DictReader(open(ks)))
# the product instantiation (STATS = 0), not the one-off instrumented launch bench.py makes for its work counters
_cands = [r for r in rows if re.search(r"rt_trace_tiles<\d+, (true|false), 0,", r["Name"])] or \
         [r for r in rows if "rt_trace_tiles" in r["Name"]]
main = max(_cands, key=lambda r: float(r["TotalDurationNs"]))
kname = main["Name"].replace("void (anonymous namespace)::", "").split("(")[0]
pm = collections.defaultdict(list)
meta = {}
for d in ("pmc_a", "pmc_b", "pmc_fetch", "pmc_write"):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        if kname in r["Kernel_Name"]:
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
avg = {k: sum(v) / len(v) for k, v in pm.items()}
out = {"kernel": kname, "calls": int(main["Calls"]), "avg_ns": float(main["AverageNs"]),
       "min_ns": float(main["MinNs"]), "max_ns": float(main["MaxNs"]), "dispatch": meta, "pmc_avg_per_dispatch": avg}
dur = float(main["AverageNs"]) * 1e-9
d = {}
if "GRBM_GUI_ACTIVE" in avg:
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0           # summed over the 8 XCDs
    d["cycles_per_dispatch"] = cyc
    if "SQ_INSTS_VALU" in avg:
        d["valu_issue_utilisation"] = avg["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc)   # 1024 SIMDs, 2 cycles per wave64 VALU op
if "SQ_WAVES" in avg and "SQ_INSTS_VALU" in avg:
    d["valu_insts_per_wave"] = avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]
    d["salu_insts_per_wave"] = avg.get("SQ_INSTS_SALU", 0) / avg["SQ_WAVES"]
    d["lds_insts_per_wave"] = avg.get("SQ_INSTS_LDS", 0) / avg["SQ_WAVES"]
if "SQ_WAVE_CYCLES" in avg:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if k in avg:
            d["share_" + k] = avg[k] / avg["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in avg:
    d["hbm_read_bytes"] = avg["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE reads half of wide streams
    d["hbm_read_bytes_uncorrected"] = avg["FETCH_SIZE"] * 1024
if "WRITE_SIZE" in avg:
    d["hbm_write_bytes"] = avg["WRITE_SIZE"] * 1024
if "hbm_read_bytes" in d and "hbm_write_bytes" in d:
    d["hbm_traffic_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    d["hbm_traffic_GBs"] = d["hbm_traffic_bytes"] / dur / 1e9
out["derived"] = d
json.dump(out, open(os.path.join("profiles", f"{name}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
