#!/bin/bash
# Run on the GPU box: kernel-trace stats of the default bench command + PMC passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd $R
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i -E "mfma|GRBM_GUI|FETCH_SIZE|WRITE_SIZE|SQ_BUSY|SQ_WAVE_CYC|SQ_WAIT" $O/counters_list.txt | head -60 > $O/counters_grep.txt
echo "== kernel trace of default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu-baseline > $O/prof_stats_bench.log 2>&1
tail -2 $O/prof_stats_bench.log
find $O/prof_stats -name "*stats*.csv" | head
echo "== pmc pass 1 (mfma busy / clock)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/prof_pmc1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc1.log 2>&1
tail -2 $O/prof_pmc1.log
echo "== pmc pass 2 (fetch)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc2.log 2>&1
tail -2 $O/prof_pmc2.log
echo "== pmc pass 3 (write)"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc3.log 2>&1
tail -2 $O/prof_pmc3.log
# keep only small summaries (the merge-back limit is 64 MiB)
find $O -name "*.csv" -size +20M -delete
du -sh $O
