#!/bin/bash
# rocprofv3 evidence for the sparse assembly kernels (pair_wave / pair_thread / dense_sparse_gather / lin_schur):
# kernel-trace stats, then L2 hit rate and HBM traffic in separate --pmc passes (never combined with other traces).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sparse${PAIR_LANES:+_lanes$PAIR_LANES}; mkdir -p $O
cd $R
for m in tru9 vib9 c5; do
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 tools/sparse_assembly_probe.py $m 5 > $O/stats_$m.log 2>&1 || { echo "stats $m failed"; tail -3 $O/stats_$m.log; }
  grep "assemble" $O/stats_$m.log
  timeout -k 10 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc_$m -- python3 tools/sparse_assembly_probe.py $m 3 > $O/tcc_$m.log 2>&1 || echo "tcc $m failed"
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$m -- python3 tools/sparse_assembly_probe.py $m 3 > $O/fetch_$m.log 2>&1 || echo "fetch $m failed"
done
python3 tools/pmc_sparse_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
find $O -name "*.csv" -size +8M -delete
