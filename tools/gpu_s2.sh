#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/s2/pytest.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/s2/bench.json 2> gpurun_out/s2/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for s in d.get('secondary',[]): print(s)
PY
