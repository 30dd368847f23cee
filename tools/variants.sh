#!/bin/bash
# time tuning builds of the engine library (RT_ENGINE_LIB selects the .so); run on the GPU box
for lib in ${LIBS:-ray-tracer-engine_amd/csrc/librt_engine*.so}; do
  echo -n "$lib "
  RT_ENGINE_LIB=$PWD/$lib python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $* 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['kernel_ms'],4), 'Mrays/s', round(d['value'],1))"
done
