#!/bin/bash
# time tuning builds of the engine library (RT_ENGINE_LIB selects the .so)
for lib in ray-tracer-engine_amd/csrc/librt_engine*.so; do
  echo -n "$lib "
  RT_ENGINE_LIB=$PWD/$lib python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $* 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('kernel_ms', round(d['kernel_ms'],3), 'Mrays/s', round(d['value'],1))"
done
