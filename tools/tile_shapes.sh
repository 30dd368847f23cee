#!/bin/bash
# frame kernel time for each tile width (64 pixels per wave: 8x8, 16x4, 32x2, 64x1)
for t in 8 16 32 64; do
  echo -n "tile $t: "
  python3 bench.py --tile $t --steps 100 --warmup 10 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4), 'ms')"
done
