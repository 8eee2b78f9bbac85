#!/bin/bash
# kernel stats of the mid-size product kernels inside a maxG11 solve: LRN_GEMM_MID=1 (three-stage LDS DMA kernel) and 0
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/mid; mkdir -p $O
for m in 1 0; do
  export LRN_GEMM_MID=$m
  rm -rf $O/prof_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python3 tools/e2e_times.py --nocpu maxG11 > $O/prof_$m.log 2>&1
  cp $(find $O/prof_$m -name "*kernel_stats.csv" | head -1) $O/maxG11_mid${m}_kernel_stats.csv
  echo "== LRN_GEMM_MID=$m"; head -8 $O/maxG11_mid${m}_kernel_stats.csv | cut -c1-150
done
find $O -name "*kernel_trace.csv" -delete
