#!/bin/bash
# usage: tools/experiments/gpu_ab.sh OUTDIR name1 name2 ...   (A/B of libaa_interp_<name>.so builds on the headline bench, 2 rounds)
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for n in "$@"; do
    lib=$PWD/interpolate_antialiasing_amd/csrc/libaa_interp_$n.so
    [ "$n" = "stock" ] && lib=$PWD/interpolate_antialiasing_amd/csrc/libaa_interp.so
    AA_INTERP_LIB=$lib timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-secondary --no-cpu-baseline > $OUT/${n}_$rep.json 2> $OUT/${n}_$rep.err || echo "$n failed"
  done
done
python - "$OUT" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['roofline']['kernel_ms'], d['roofline']['frac'], 'err', d['max_abs_err_vs_oracle'])
    except Exception as e:
        print(f, 'ERR', e)
PY
