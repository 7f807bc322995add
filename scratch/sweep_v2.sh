#!/bin/bash
run() { timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"; }
export AA_V2_LDSMODE=1
for cfg in "4 8 2" "4 16 2" "4 4 2" "2 8 2" "2 4 2" "2 16 2" "8 8 2" "1 8 2" "1 4 2" "4 8 1" "4 8 3" "2 8 1" "2 8 3"; do
  set -- $cfg
  export AA_V2_K=$1 AA_V2_RS=$2 AA_V2_NCONS=$3
  run "K=$1 RS=$2 NCONS=$3"
done
