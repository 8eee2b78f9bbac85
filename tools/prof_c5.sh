#!/bin/bash
# rocprofv3 kernel stats of a few IP iterations of C5 (kit=1, H_beta) at full size
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 tools/c5_solve.py 10000 20000 4 ${C5_ITERS:-4} > gpurun_out/prof_c5.log 2>&1
tail -1 gpurun_out/prof_c5.log | cut -c1-300
f=$(find gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-170
find gpurun_out/prof_c5 -name "*kernel_trace.csv" -delete
