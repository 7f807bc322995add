#!/bin/bash
timeout -k 5 120 python scratch/dbg_fused.py 2>&1 | grep "bad frac" | awk '{print $(NF-6), $NF, $(NF-2)}' | tr '\n' ' '; echo
run() { timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"; }
run "default(spb=5)"
for s in 1 2 3; do export AA_V3_SPB=$s; run "spb=$s"; done
unset AA_V3_SPB
for yb in 1 2 3 4 6; do export AA_FUSED_YBANDS=$yb; run "spb=5 yb=$yb"; done
