#!/bin/bash
# usage: tools/experiments/ab_workloads.sh "lib1 lib2 ..." "workload1 workload2 ..." [launches]   (two rounds; 'stock' = product build)
set -o pipefail
for rep in 1 2; do
  for n in $1; do
    L=$PWD/interpolate_antialiasing_amd/csrc/libaa_interp_$n.so
    [ "$n" = stock ] && L=$PWD/interpolate_antialiasing_amd/csrc/libaa_interp.so
    for w in $2; do
      echo -n "$n: "
      AA_TIME=1 AA_INTERP_LIB=$L timeout -k 10 120 python tools/workload.py $w ${3:-60} 2>&1 | grep -v amdgpu.ids || exit 1
    done
  done
done
