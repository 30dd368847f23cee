#!/usr/bin/env python3
"""Static attribution of the frame kernel's instructions to source lines of its body.

Compiles csrc/rt_kernels.hip with -g for gfx950, disassembles one instantiation of
rt_trace_tiles and asks llvm-symbolizer for the inline stack of every instruction; the
outermost frame (a line of the kernel body) gets the instruction. Static counts only:
loops are not weighted. Use together with the ablation timings (tools/ablate.sh).

  python3 tools/static_attrib.py [--inst 'ILi8ELb1ELi0ELb1ELb0E'] [--by-callee]
"""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray-tracer-engine_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm", "-g", "-DRT_QUICK"]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inst", default="ILi8ELb1ELi0ELb0ELi0ELb0E")
    ap.add_argument("--by-callee", action="store_true")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="rt_attrib_")
    obj, co = os.path.join(tmp, "k.o"), os.path.join(tmp, "k.co")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-x", "hip", "--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", co,
                           os.path.join(CSRC, "rt_kernels.hip")], stderr=subprocess.DEVNULL)
    syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", co], text=True)
    sym = [l.split()[-1] for l in syms.splitlines()
           if "rt_trace_tiles" + a.inst in l and " FUNC " in l][0]
    dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn",
                                   f"--disassemble-symbols={sym}", co], text=True)
    insts = []
    for l in dis.splitlines():
        m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-F]{12}):", l)
        if m:
            insts.append((int(m.group(2), 16), m.group(1)))
    inp = "\n".join(hex(ad) for ad, _ in insts) + "\n"
    out = subprocess.check_output([f"{LLVM}/llvm-symbolizer", f"--obj={co}", "-i", "-f", "-s",
                                   "--output-style=LLVM"], input=inp, text=True)
    blocks = out.strip().split("\n\n")
    assert len(blocks) == len(insts), (len(blocks), len(insts))
    per = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
    for (ad, op), b in zip(insts, blocks):
        ls = b.splitlines()
        frames = [(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)]
        outer_line = 0; callee = frames[0][0].split("(")[0].split("::")[-1]
        for fn, loc in frames:
            if "rt_trace_tiles" in fn:
                outer_line = int(loc.split(":")[1])
        cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
               "lds" if op.startswith("ds_") else "mem")
        key = (outer_line, callee) if a.by_callee else outer_line
        per[key] += 1; kinds[key][cls] += 1
    src = open(os.path.join(CSRC, "rt_trace.inc")).read().splitlines()
    tot = collections.Counter()
    for k in kinds: tot.update(kinds[k])
    print(f"{sym}: {len(insts)} instructions: {dict(tot)}")
    for key, n in sorted(per.items(), key=lambda kv: (kv[0] if isinstance(kv[0], int) else kv[0][0])):
        line = key if isinstance(key, int) else key[0]
        text = src[line - 1].strip()[:90] if 0 < line <= len(src) else ""
        k = kinds[key]
        tag = "" if isinstance(key, int) else f" [{key[1]}]"
        print(f"{line:5d} valu {k['valu']:4d} salu {k['salu']:4d} lds {k['lds']:3d} mem {k['mem']:3d}{tag}  {text}")

if __name__ == "__main__":
    main()
