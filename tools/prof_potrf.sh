#!/bin/bash
# rocprofv3 kernel stats of the blocked Cholesky at the sizes given (default 800 4000)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_potrf
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_potrf -- python3 tools/potrf_probe.py ${@:-800 4000} > gpurun_out/prof_potrf.log 2>&1
grep "^n " gpurun_out/prof_potrf.log
f=$(find gpurun_out/prof_potrf -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-160
