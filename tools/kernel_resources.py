#!/usr/bin/env python3
"""Register / spill / scratch table of every kernel in csrc/rt_kernels.hip, as the product
Makefile compiles it (gfx950 code object metadata, `llvm-readelf --notes`).

  python3 tools/kernel_resources.py [--extra "-DRT_TUNING"] [--json out.json]
"""
import argparse, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray-tracer-engine_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm"]
KEYS = ("kernarg_segment_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
        "vgpr_spill_count", "group_segment_fixed_size")


def demangle(n):
    try:
        return subprocess.check_output([f"{LLVM}/llvm-cxxfilt", n], text=True).strip()
    except Exception:
        return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--extra", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--src", default="rt_kernels.hip")
    a = ap.parse_args()
    co = os.path.join(tempfile.mkdtemp(prefix="rt_res_"), "k.co")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *a.extra.split(), "-x", "hip", "--cuda-device-only",
                           "--no-gpu-bundle-output", "-c", os.path.join(CSRC, a.src), "-o", co])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    rows, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "name":
            cur["name"] = v
        elif k in KEYS:
            cur[k] = int(v)
        if k == "wavefront_size":      # last key of a kernel record
            rows.append(cur)
            cur = {}
    if cur.get("name"):
        rows.append(cur)
    for r in rows:
        r["name"] = re.sub(r"\(anonymous namespace\)::", "", demangle(r["name"]))
        r["name"] = re.sub(r"\(.*$", "", r["name"]).replace("void ", "")
    rows.sort(key=lambda r: r["name"])
    print(f"{'kernel':58s} karg  scr sgpr sspl vgpr vspl")
    for r in rows:
        print(f"{r['name'][:58]:58s} {r.get('kernarg_segment_size', 0):4d} {r.get('private_segment_fixed_size', 0):4d} "
              f"{r.get('sgpr_count', 0):4d} {r.get('sgpr_spill_count', 0):4d} {r.get('vgpr_count', 0):4d} "
              f"{r.get('vgpr_spill_count', 0):4d}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"flags": FLAGS + a.extra.split(), "kernels": rows}, f, indent=1)


if __name__ == "__main__":
    main()
