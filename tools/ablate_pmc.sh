#!/bin/bash
# Instruction counts and wave-cycle shares of the frame kernel with parts skipped (RT_ABLATE bits,
# rt_kernels.hip). Needs the TUNING build of the library:
#   make -C ray-tracer-engine_amd/csrc EXTRA=-DRT_TUNING OUT=librt_engine_tuning.so BUILD=build_tuning
# Usage (on the GPU box): tools/ablate_pmc.sh [bits ...]
export TMPDIR=/tmp
export RT_ENGINE_LIB=$PWD/ray-tracer-engine_amd/csrc/librt_engine_tuning.so
for a in ${*:-0 16 24 7 3 1 2 4 4096}; do
  rm -rf /tmp/ab_$a
  RT_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/ab_$a -- python3 bench.py --steps 3 --warmup 1 --no-preroll --no-cpu-baseline --no-extras > /dev/null 2>&1
  ms=$(RT_ABLATE=$a python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['kernel_ms'],4))")
  python3 - $a $ms <<'PY'
import csv,glob,sys,collections
a=sys.argv[1]
f=glob.glob(f'/tmp/ab_{a}/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'rt_trace_tiles<8, true, 0,' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
w=129600
print(f"ablate={a} kernel_ms {sys.argv[2]} VALU/wave {m['SQ_INSTS_VALU']/w:.0f} SALU/wave {m['SQ_INSTS_SALU']/w:.0f} LDS/wave {m['SQ_INSTS_LDS']/w:.0f} SMEM/wave {m['SQ_INSTS_SMEM']/w:.0f} wavecyc/wave {m['SQ_WAVE_CYCLES']/w*4:.0f} active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} stall {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
done
