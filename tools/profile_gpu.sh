#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of the bench
# command. Summaries land under gpurun_out/prof_<tag>/ ; copy what should be
# judged into profiles/.
set -u
TAG=${1:-r1}
shift || true
# flags for every run of the profile (e.g. FLAGS="--table-lds"); the trace run takes 20 steps, the counter runs 3
FLAGS=${FLAGS:-}
BENCH_ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-extras $FLAGS"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $BENCH_ARGS > $OUT/kt.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
   --output-format csv -d $OUT/pmc_a -- python3 bench.py --steps 3 --warmup 1 --no-preroll --no-cpu-baseline --no-extras $FLAGS > $OUT/pmc_a.log 2>&1 || exit 2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM \
   --output-format csv -d $OUT/pmc_b -- python3 bench.py --steps 3 --warmup 1 --no-preroll --no-cpu-baseline --no-extras $FLAGS > $OUT/pmc_b.log 2>&1 || exit 3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-preroll --no-cpu-baseline --no-extras $FLAGS > $OUT/pmc_fetch.log 2>&1 || exit 4
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-preroll --no-cpu-baseline --no-extras $FLAGS > $OUT/pmc_write.log 2>&1 || exit 5
find $OUT -name "*.csv" | head -50
