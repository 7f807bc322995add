#!/bin/bash
# Round-end evidence run: full GPU suite, smoke, the bench (default flags), and the bench command under rocprofv3 --kernel-trace --stats.
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 $R/bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $R/gpurun_out/final/bench_prof.json 2> $R/gpurun_out/final/bench_prof.err); echo "prof rc=$?"
f=$(ls gpurun_out/final/prof/*/*kernel_stats.csv | head -1); head -3 $f | cut -c1-40,140-260
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d['roofline'].items() if k!='copy_ceiling_by_kernel_form'})
for s in d.get('secondary',[]): print(s.get('ms'), s.get('GB/s'), s.get('variant'), '|', s['workload'][:70])
print(d.get('cpu_baseline'))
PY
