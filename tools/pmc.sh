#!/bin/bash
# usage: tools/pmc.sh <tag> : PMC passes over a short bench run, CSVs under gpurun_out/pmc_<tag>/
R=$GRAFT_REPO_ROOT
TAG=$1
export TMPDIR=/tmp
cd /tmp
i=0
mkdir -p $R/gpurun_out/pmc_$TAG
for CNT in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_$TAG/log$i.txt 2>&1 || echo "pass $i failed" 
done
