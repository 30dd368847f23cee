#!/bin/bash
# A/B timing of engine builds on one box (run on the GPU box): every library named in LIBS (default: all
# ray-tracer-engine_amd/csrc/librt_engine*.so except the tuning build) through the serial C3 loop, ROUNDS times, interleaved.
ROUNDS=${ROUNDS:-3}
LIBS=${LIBS:-$(ls ray-tracer-engine_amd/csrc/librt_engine*.so | grep -v tuning)}
for r in $(seq $ROUNDS); do
  for lib in $LIBS; do
    echo -n "$lib "
    RT_ENGINE_LIB=$PWD/$lib python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras $* 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['kernel_ms'],4), 'Mrays/s', round(d['value'],1), 'sclk', d['clocks']['after']['sclk_mhz'])"
  done
done
