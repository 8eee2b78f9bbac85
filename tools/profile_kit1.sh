#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the kit=1 (PCG) path:
#  C3 thetaG11 with H_alpha, and a few IP iterations of C5 (sparse constraints, H_beta) at full size.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== C3 thetaG11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 tools/e2e_times.py thetaG11 > $O/prof_c3.log 2>&1
tail -1 $O/prof_c3.log
echo "== C5 ${C5_MSZ:-10000} x ${C5_NVAR:-20000}, ${C5_ITERS:-4} IP iterations"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 tools/c5_solve.py ${C5_MSZ:-10000} ${C5_NVAR:-20000} 4 ${C5_ITERS:-4} > $O/prof_c5.log 2>&1
tail -1 $O/prof_c5.log
for d in prof_c3 prof_c5; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
du -sh $O
