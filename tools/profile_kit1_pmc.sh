#!/bin/bash
# Run on the GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the kit=1 kernels:
# C5 at full size (2 IP iterations) and C3 thetaG11.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd $R
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c5_$cnt $O/pmc_c3_$cnt
  echo "== C5 $cnt"
  timeout -k 10 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_c5_$cnt -- python3 tools/c5_solve.py 10000 20000 4 2 > $O/pmc_c5_$cnt.log 2>&1
  tail -1 $O/pmc_c5_$cnt.log | cut -c1-200
  echo "== C3 $cnt"
  timeout -k 10 200 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_c3_$cnt -- python3 tools/e2e_times.py --nocpu thetaG11 > $O/pmc_c3_$cnt.log 2>&1
  tail -1 $O/pmc_c3_$cnt.log | cut -c1-200
done
python3 tools/pmc_kit1_summary.py
find $O -name "*counter_collection.csv" -size +30M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
