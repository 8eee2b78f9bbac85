#!/bin/bash
# Round-2 evidence, one box: rocprofv3 kernel-trace stats of the default bench command, then the PMC passes in their
# own runs (MFMA busy / clock, FETCH_SIZE, WRITE_SIZE, L2 hits), the per-rank replay of both multi-GPU layouts, the
# parity configs end to end and the full C4 / C5 solves.  Summaries are copied to profiles/r02_* by the caller.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02; mkdir -p $O
cd $R
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
echo "== bench (plain)"; step 300 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
echo "== kernel trace of the default bench command"
step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/prof_stats.err
tail -c 300 $O/bench_under_rocprof.json; echo
echo "== pmc 1 (mfma busy / clock)"
step 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/prof_pmc1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc1.log 2>&1
echo "== pmc 2 (fetch)"
step 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc2.log 2>&1
echo "== pmc 3 (write)"
step 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc3.log 2>&1
echo "== pmc 4 (L2 hits)"
step 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_pmc4 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc4.log 2>&1
python3 tools/pmc_summary_r02.py $O > $O/pmc_derived.txt 2>&1; cat $O/pmc_derived.txt
echo "== per-rank replay, both layouts"
step 400 python3 tools/shard_balance.py > $O/shard_balance.txt 2>&1; grep "^\[" $O/shard_balance.txt
echo "== parity configs end to end"
E2E_OUT=$O/e2e_a.json step 600 python3 tools/e2e_times.py --cpu theta1 maxG11 > $O/e2e_a.log 2>&1
E2E_OUT=$O/e2e_b.json step 400 python3 tools/e2e_times.py thetaG11 tru9 vib9 > $O/e2e_b.log 2>&1; grep -h "^[a-zA-Z0-9]* {" $O/e2e_a.log $O/e2e_b.log | cut -c1-400
echo "== full C4 solve"
step 300 python3 tools/c4_full_solve.py > $O/c4_full_solve.log 2>&1; tail -1 $O/c4_full_solve.log | cut -c1-600
cp gpurun_out/c4_full_solve_2000_4000.json $O/ 2>/dev/null
find $O -name "*.csv" -size +12M -delete
du -sh $O
