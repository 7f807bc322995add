#!/bin/bash
for m in 1 0 2; do
  export AA_V2_LDSMODE=$m
  echo "=== LDSMODE=$m"
  timeout -k 5 120 python scratch/dbg_fused.py 2>&1 | grep "bad frac"
  timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"
done
