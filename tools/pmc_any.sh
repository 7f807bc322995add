#!/bin/bash
# usage: tools/pmc_any.sh TAG WORKLOAD : kernel-trace stats + PMC passes (separate passes, counters only with --kernel-trace)
# over tools/workload.py WORKLOAD; CSVs under gpurun_out/pmc_TAG/.  Summaries: tools/pmc_summary.py TAG <kernel substring>.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; WL=$2
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_$TAG
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_$TAG/stats -- python3 $R/tools/workload.py $WL 600 > $R/gpurun_out/pmc_$TAG/log0.txt 2>&1 || echo "stats pass failed"
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/tools/workload.py $WL > $R/gpurun_out/pmc_$TAG/log$i.txt 2>&1 || echo "pass $i failed"
done
cd $R
python3 tools/pmc_summary.py $TAG ${3:-fused} > gpurun_out/pmc_$TAG/summary.json 2>/dev/null
f=$(ls gpurun_out/pmc_$TAG/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -3 "$f" | cut -c1-200
cat gpurun_out/pmc_$TAG/summary.json | python3 -c "
import json,sys
d=json.load(sys.stdin)
print({k:round(v['avg_per_dispatch'],1) for k,v in d.items()})"
