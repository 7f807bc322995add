#!/bin/bash
# The backward kernel's 906-column residue (verdict round 2, item 6): L2 <-> memory write-path counters for input widths W (default:
# 896 906 912), one rocprofv3 --pmc pass per counter group and width (counters only with --kernel-trace), plus a --stats pass for
# the duration.  Output: gpurun_out/bwdw/<W>/..., digest by tools/pmc_bwd_digest.py -> profiles/r03_bwd_906_vs_896.json
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
WIDTHS=${@:-896 906 912}
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*" | sort -u > $R/gpurun_out/bwdw_counters_available.txt
for W in $WIDTHS; do
  D=$R/gpurun_out/bwdw/$W
  mkdir -p $D
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools/workload.py bwdw:$W 300 > $D/log0.txt 2>&1 || echo "stats pass failed ($W)"
  i=0
  for CNT in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_DRAM_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $D/p$i -- python3 $R/tools/workload.py bwdw:$W 6 > $D/log$i.txt 2>&1 || echo "pass $i failed ($W: $CNT)"
  done
done
cd $R
python3 tools/pmc_bwd_digest.py $WIDTHS
