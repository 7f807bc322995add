#!/bin/bash
run() { timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"; }
for a in 0 1 2 3 4; do export AA_V2_ABL=$a; run "ABL=$a"; done
