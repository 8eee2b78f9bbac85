#!/bin/bash
# Round-3 evidence, one box.  Part A (default): bench line, rocprofv3 kernel-trace stats of the default bench command,
# the parity configs end to end with kernel stats of C2 / C3, the full C4 and C5 solves, the prepare_W probe.
# Part B (PMC=1): the counter passes of the bench command, each in its own run.  Summaries are copied to profiles/r03_*.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
if [ -z "$PMC" ]; then
echo "== bench (plain)"; step 300 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
echo "== kernel trace of the default bench command"
rm -rf $O/prof_stats
step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/prof_stats.err
tail -c 300 $O/bench_under_rocprof.json; echo
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "== prepare_W probe (eigen-free against the SVD route)"
COND_X=1e8 step 300 python3 tools/nt_probe.py 800 2000 > $O/prepw_probe.txt 2>&1; grep -v amdgpu $O/prepw_probe.txt
echo "== parity configs end to end"
E2E_OUT=$O/e2e.json step 600 python3 tools/e2e_times.py --nocpu > $O/e2e.log 2>&1; grep -h "^[a-zA-Z0-9]* {" $O/e2e.log | cut -c1-420
for name in maxG11 thetaG11; do
  rm -rf $O/prof_$name
  step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 tools/e2e_times.py --nocpu $name > $O/prof_$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
done
echo "== blocked Cholesky (time, backward error; kernel stats at n = 800 and 4000)"
step 200 python3 tools/potrf_probe.py 145 800 3240 4000 > $O/potrf_probe.txt 2>&1; grep "^n " $O/potrf_probe.txt
rm -rf $O/prof_potrf
step 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_potrf -- python3 tools/potrf_probe.py 800 4000 > $O/prof_potrf.log 2>&1
cp $(find $O/prof_potrf -name "*kernel_stats.csv" | head -1) $O/potrf_kernel_stats.csv
echo "== C3: dense against pattern-restricted operator of the CG mat-vec"
step 200 python3 tools/c3_matvec_ab.py > $O/c3_matvec_ab.txt 2>&1; grep "^matvec" $O/c3_matvec_ab.txt
echo "== full C4 solve"
step 300 python3 tools/c4_full_solve.py > $O/c4_full_solve.log 2>&1; tail -1 $O/c4_full_solve.log | cut -c1-700
cp gpurun_out/c4_full_solve_2000_4000.json $O/ 2>/dev/null
echo "== full C5 solve"
step 600 python3 tools/c5_solve.py > $O/c5_full_solve.log 2>&1; tail -1 $O/c5_full_solve.log | cut -c1-700
else
rm -rf $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3 $O/prof_pmc4
echo "== pmc 1 (mfma busy / clock)"
step 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/prof_pmc1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc1.log 2>&1
echo "== pmc 2 (fetch)"
step 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc2.log 2>&1
echo "== pmc 3 (write)"
step 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc3.log 2>&1
echo "== pmc 4 (L2 hits)"
step 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_pmc4 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_pmc4.log 2>&1
python3 tools/pmc_summary_r02.py $O > $O/pmc_derived.txt 2>&1; cat $O/pmc_derived.txt
fi
find $O -name "*.csv" -size +12M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
