#!/bin/bash
# usage: tools/pmc_round.sh PREFIX workload...   : tools/pmc_any.sh for each workload (tags PREFIX_<workload>), one after the other
P=$1; shift
for w in "$@"; do
  echo "=== $w"; bash tools/pmc_any.sh ${P}_$w $w > gpurun_out/pmc_round_$w.log 2>&1; tail -2 gpurun_out/pmc_round_$w.log | cut -c1-300
done
