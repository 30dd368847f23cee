#!/bin/bash
# run the CPU oracle under AddressSanitizer + UBSan (CPU only).
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
cat > /tmp/rt_asan_run.py <<'PY'
import sys
sys.path[:0] = ['.', 'oracle', 'tests']
import oracle_py
orig = oracle_py.C.CDLL
oracle_py.C.CDLL = lambda p, *a, **k: orig(p.replace('librt_oracle.so', 'librt_oracle_asan.so'), *a, **k)
import rt_amd
rt = rt_amd.load()
from scenes import Inputs, mixed_scene, mixed_oracle_render
for n, (w, h) in ((8, (64, 64)), (256, (96, 54)), (1024, (48, 27))):
    Inputs(rt, n).oracle_render(oracle_py, w, h, nthreads=4)
mixed_oracle_render(mixed_scene(rt), oracle_py, 64, 40)
print("oracle: ASan/UBSan run clean")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python3 /tmp/rt_asan_run.py
