#!/bin/bash
# instruction counts and wave-cycle shares of the frame kernel with parts skipped
export TMPDIR=/tmp
for a in 0 16 7 3 1 2 4; do
  rm -rf /tmp/ab_$a
  RT_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/ab_$a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
  python3 - $a <<'PY'
import csv,glob,sys,collections
a=sys.argv[1]
f=glob.glob(f'/tmp/ab_{a}/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'rt_trace_tiles<8, true, 0,' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
w=129600
print(f"ablate={a} VALU/wave {m['SQ_INSTS_VALU']/w:.0f} SALU/wave {m['SQ_INSTS_SALU']/w:.0f} LDS/wave {m['SQ_INSTS_LDS']/w:.0f} SMEM/wave {m['SQ_INSTS_SMEM']/w:.0f} wavecyc/wave {m['SQ_WAVE_CYCLES']/w*4:.0f} active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} stall {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
done
