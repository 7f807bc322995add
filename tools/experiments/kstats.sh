#!/bin/bash
# usage: tools/experiments/kstats.sh WORKLOAD [launches] : per-kernel average duration of one workload (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
D=$R/gpurun_out/kstats_$(echo $1 | tr ':' '_')
rm -rf $D; mkdir -p $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/workload.py $1 ${2:-100} > $D/log.txt 2>&1 || { echo "failed"; tail -5 $D/log.txt; exit 1; }
F=$(find $D -name "*kernel_stats.csv" | head -1)
echo "== $1"; cut -d, -f1-4 $F | cut -c1-150
