#!/bin/bash
# A/B of the GEMM3' strip schedules on one box, per-kernel durations by rocprofv3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  echo "== $name"
  env "$@" timeout -k 10 100 python3 tools/ab_strip.py 0 > gpurun_out/ab_$name.plain.log 2>&1
  grep "^rep" gpurun_out/ab_$name.plain.log
}
prof() {
  name=$1; shift
  for kv in "$@"; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$name -- python3 tools/ab_strip.py 0 > gpurun_out/ab_$name.log 2>&1
  for kv in "$@"; do unset "${kv%%=*}"; done
  f=$(find gpurun_out/ab_$name -name "*kernel_stats.csv" | head -1)
  grep kseg $f | cut -c1-150
}
run off AB_STRIPS=0
run m0 AB_STRIPS=1
run m1 AB_STRIPS=1 LRN_STRIP_MODE=1
run m2 AB_STRIPS=1 LRN_STRIP_MODE=2
run off_again AB_STRIPS=0
prof m1 AB_STRIPS=1 LRN_STRIP_MODE=1
prof m0 AB_STRIPS=1
prof off AB_STRIPS=0
