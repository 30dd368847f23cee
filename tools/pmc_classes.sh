#!/bin/bash
# Dynamic instruction mix of the frame kernel by class (rocprofv3 PMC; run on the GPU box). One pass per counter group
# (the SQ has 8 counters per pass). Prints per-wave averages of the product instantiation.
export TMPDIR=/tmp
OUT=${OUT:-/tmp/pmc_classes}
LIBARG=${RT_ENGINE_LIB:+RT_ENGINE_LIB=$RT_ENGINE_LIB}
rm -rf $OUT; mkdir -p $OUT
G1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64"
G2="SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH"
G3="SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SENDMSG SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 bench.py --steps 3 --warmup 1 --no-preroll --no-cpu-baseline --no-extras $* > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $OUT/g$i.log; }
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections,re
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if re.search(r"rt_trace_tiles<8, true, 0,", r['Kernel_Name']):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
w=m.get('SQ_WAVES',129600.0)
for k in sorted(m): print(f"{k:28s} {m[k]/w:10.1f} per wave")
PY
