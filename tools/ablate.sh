#!/bin/bash
# price parts of the kernel by skipping them (results are wrong; timing only)
for a in 0 1 2 3 4 7 8 16 32 39; do
  echo -n "ablate=$a "
  RT_ABLATE=$a python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('kernel_ms', round(d['kernel_ms'],3))"
done
