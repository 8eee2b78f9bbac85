#!/bin/bash
# round 4: per-iteration breakdown + rocprofv3 kernel stats of C2 (maxG11) and C3 (thetaG11), resident driver
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for name in maxG11 thetaG11; do
  python3 tools/iter_breakdown.py $name > $O/breakdown_$name.txt 2>&1
  tail -6 $O/breakdown_$name.txt | cut -c1-260
  rm -rf $O/prof_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 tools/e2e_times.py --nocpu $name > $O/prof_$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  head -5 $O/${name}_kernel_stats.csv | cut -c1-140
done
find $O -name "*kernel_trace.csv" -delete
