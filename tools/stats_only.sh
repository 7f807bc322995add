#!/bin/bash
# usage: tools/stats_only.sh TAG WORKLOAD : rocprofv3 --kernel-trace --stats over 600 launches (warm clocks) -> gpurun_out/pmc_TAG/stats
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_$1
rm -rf $R/gpurun_out/pmc_$1/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_$1/stats -- python3 $R/tools/workload.py $2 600 > $R/gpurun_out/pmc_$1/log0.txt 2>&1 || echo "stats pass failed"
