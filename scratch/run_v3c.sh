#!/bin/bash
run() { timeout -k 5 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"; }
for g in 8 4 6 10; do export AA_V3_G=$g; run "G=$g"; run "G=$g"; done
