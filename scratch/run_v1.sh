#!/bin/bash
python - <<'PY'
import sys
sys.path.insert(0,'.')
from interpolate_antialiasing_amd import _lib
PY
run() { timeout -k 5 100 python - <<PY
import sys, json, subprocess, os
sys.path.insert(0,'.')
from interpolate_antialiasing_amd import _lib
_lib.set_fused(2)
sys.argv=['bench.py','--steps','20','--warmup','5','--no-cpu-baseline']
import runpy
runpy.run_path('bench.py', run_name='__main__')
PY
}
echo "unaligned:"; run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"
export AA_FUSED_ALIGNED_LOADS=1
echo "aligned:"; run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['variant'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['max_abs_err_vs_oracle'])"
