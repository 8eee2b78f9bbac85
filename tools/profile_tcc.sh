#!/bin/bash
# L2 (TCC) hit rate of the bench kernels: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum on the default bench command.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
cd $R
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_pmc4 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc4.log 2>&1
tail -1 $O/prof_pmc4.log | cut -c1-200
