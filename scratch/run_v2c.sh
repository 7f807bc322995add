#!/bin/bash
timeout -k 5 120 python scratch/dbg_fused.py 2>&1 | grep "bad frac" | awk '{print $NF, $(NF-2)}' | tr '\n' ' '; echo
run() { timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"; }
for g in 8 6 4 10 12; do export AA_V2_G=$g; run "G=$g"; done
export AA_V2_G=8
for nc in 1 3; do export AA_V2_NCONS=$nc; run "G=8 NCONS=$nc"; done
unset AA_V2_NCONS
for yb in 1 2 4 6; do export AA_FUSED_YBANDS=$yb; run "G=8 yb=$yb"; done
