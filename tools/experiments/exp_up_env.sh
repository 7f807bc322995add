#!/bin/bash
# usage: tools/experiments/exp_up_env.sh WORKLOAD "ENV1=.. ENV2=.." "ENV.." ...   (event-timed workload under each environment, twice)
W=$1; shift
for rep in 1 2; do
  for e in "$@"; do
    echo -n "[$e] "
    env $e AA_TIME=1 timeout -k 10 120 python tools/workload.py $W 60 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
  done
done
