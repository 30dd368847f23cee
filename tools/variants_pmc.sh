#!/bin/bash
# instruction counts and wave-cycle shares of the frame kernel for each tuning build (RT_ENGINE_LIB)
export TMPDIR=/tmp
for lib in ${*:-ray-tracer-engine_amd/csrc/librt_engine*.so}; do
  t=$(basename $lib .so)
  rm -rf /tmp/vp_$t
  RT_ENGINE_LIB=$PWD/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/vp_$t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
  python3 - $t <<'PY'
import csv,glob,sys,collections
a=sys.argv[1]
f=glob.glob(f'/tmp/vp_{a}/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'rt_trace_tiles<8, true, 0,' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
w=129600
print(f"{a}: VALU/wave {m['SQ_INSTS_VALU']/w:.0f} SALU/wave {m['SQ_INSTS_SALU']/w:.0f} LDS/wave {m['SQ_INSTS_LDS']/w:.0f} SMEM/wave {m['SQ_INSTS_SMEM']/w:.0f} wavecyc/wave {m['SQ_WAVE_CYCLES']/w*4:.0f} active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} stall {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
done
