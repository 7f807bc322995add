#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for mode in perm noperm; do
  if [ $mode = noperm ]; then export AA_F32_NOPERM=1; fi
  mkdir -p $R/gpurun_out/pmc_c2_$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_c2_$mode/p1 -- python3 $R/scratch/bench_c2.py > $R/gpurun_out/pmc_c2_$mode/log.txt 2>&1
done
