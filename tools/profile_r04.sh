#!/bin/bash
# Round-4 evidence, one box, one batch from the final tree.
#   PART=A (default)  bench line; rocprofv3 kernel-trace stats of the default bench command; parity configs end to end with
#                     kernel stats of C2 / C3; Cholesky probe up to nvar 20000; full C4 solve; full C5 solve + kernel stats
#   PART=B            PMC passes of the bench command (each in its own run) -> pmc_summary.csv, l2_hit_rate.csv, pmc_derived.txt
#   PART=C            HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) of the kit=1 kernels: C5 (5 IP iterations, the
#                     operator goes through the assembled matrix from the second one on) and C3
# Summaries are copied to profiles/r04_* by hand (tools/README.md).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
case "${PART:-A}" in
A)
echo "== bench (plain)"; step 400 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo
echo "== kernel trace of the default bench command"
rm -rf $O/prof_stats
step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/prof_stats.err
tail -c 300 $O/bench_under_rocprof.json; echo
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "== parity configs end to end"
E2E_OUT=$O/e2e.json step 600 python3 tools/e2e_times.py --nocpu > $O/e2e.log 2>&1; grep -h "^[a-zA-Z0-9]* {" $O/e2e.log | cut -c1-420
for name in maxG11 thetaG11; do
  python3 tools/iter_breakdown.py $name > $O/breakdown_$name.txt 2>&1
  rm -rf $O/prof_$name
  step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 tools/e2e_times.py --nocpu $name > $O/prof_$name.log 2>&1
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
done
echo "== blocked Cholesky (time, backward error) up to nvar 20000"
step 400 python3 tools/potrf_probe.py 800 3240 4000 10000 20000 > $O/potrf_probe.txt 2>&1; grep "^n " $O/potrf_probe.txt
echo "== full C4 solve"
step 300 python3 tools/c4_full_solve.py > $O/c4_full_solve.log 2>&1; tail -1 $O/c4_full_solve.log | cut -c1-700
cp gpurun_out/c4_full_solve_2000_4000.json $O/ 2>/dev/null
echo "== full C5 solve"
step 600 python3 tools/c5_solve.py > $O/c5_full_solve.log 2>&1; tail -1 $O/c5_full_solve.log | cut -c1-900
echo "== C5 kernel stats (5 IP iterations)"
rm -rf $O/prof_c5
step 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 tools/c5_solve.py 10000 20000 4 5 > $O/prof_c5.log 2>&1
cp $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv; head -14 $O/c5_kernel_stats.csv | cut -c1-150
;;
B)
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
;;
C)
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_c5_$cnt $R/gpurun_out/pmc_c3_$cnt
  echo "== C5 $cnt"
  step 400 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c5_$cnt -- python3 tools/c5_solve.py 10000 20000 4 5 > $O/pmc_c5_$cnt.log 2>&1
  tail -1 $O/pmc_c5_$cnt.log | cut -c1-200
  echo "== C3 $cnt"
  step 200 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c3_$cnt -- python3 tools/e2e_times.py --nocpu thetaG11 > $O/pmc_c3_$cnt.log 2>&1
  tail -1 $O/pmc_c3_$cnt.log | cut -c1-200
done
KIT1_OUT=$O/kit1_hbm_traffic.csv python3 tools/pmc_kit1_summary.py
find $R/gpurun_out -name "*counter_collection.csv" -size +30M -delete
;;
esac
find $O -name "*.csv" -size +12M -delete
find $O -name "*kernel_trace.csv" -delete
du -sh $O
