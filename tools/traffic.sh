#!/bin/bash
# HBM traffic of the frame kernel for each tuning build (FETCH_SIZE / WRITE_SIZE, KB)
export TMPDIR=/tmp
for lib in ray-tracer-engine_amd/csrc/librt_engine*.so; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$c
    RT_ENGINE_LIB=$PWD/$lib rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    v=$(grep -h "rt_trace_tiles<8, true, 0," /tmp/tr_$c/*/*counter_collection.csv | tail -1 | python3 -c "import sys,csv; r=next(csv.reader(sys.stdin)); print(r[-3])")
    echo "$lib $c $v"
  done
done
