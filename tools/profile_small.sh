#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of whole solves of the small BASELINE configs (C2 maxG11, C3 thetaG11).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd $R
for name in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 tools/e2e_times.py --nocpu $name > $O/prof_$name.log 2>&1
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${name}_kernel_stats.csv
  head -40 $O/${name}_kernel_stats.csv | cut -c1-150
  tail -1 $O/prof_$name.log | cut -c1-400
  find $O/prof_$name -name "*.csv" -size +10M -delete
done
