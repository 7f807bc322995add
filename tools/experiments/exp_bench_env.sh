#!/bin/bash
# usage: tools/experiments/exp_bench_env.sh "ENV1=.. ENV2=.." "ENV1=.."   (headline bench under each environment, kernel ms)
for e in "$@"; do
  r=$(env $e AA_V3_DEBUG=1 python bench.py --steps 150 --warmup 30 --no-secondary --no-cpu-baseline 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['max_abs_err_vs_oracle'])")
  echo "$e => $r | $(grep 'aa v3' /tmp/err.txt | tail -1 | cut -c1-90)"
done
