#!/bin/bash
# GPU session 1 of round 2: parity suite, then A/B of the unaligned-LDS window reads on the headline bench.
set -o pipefail
mkdir -p gpurun_out/s1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1/pytest.log
tail -5 gpurun_out/s1/pytest.log
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/s1/bench_unaligned_$rep.json 2> gpurun_out/s1/bench_unaligned_$rep.err; echo "unaligned rc=$?"
  AA_INTERP_LIB=$PWD/interpolate_antialiasing_amd/csrc/libaa_interp_aligned.so timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/s1/bench_aligned_$rep.json 2> gpurun_out/s1/bench_aligned_$rep.err; echo "aligned rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('copy_ceiling_measured_GBs'), d['max_abs_err_vs_oracle'], d['config']['variant'])
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout -k 10 400 python bench.py > gpurun_out/s1/bench_full.json 2> gpurun_out/s1/bench_full.err; echo "full rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s1/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for s in d.get('secondary',[]): print(s)
print(d.get('cpu_baseline'))
PY
