#!/bin/bash
# round 2, first GPU call: whole GPU suite on the refactored library, per-rank replay of the new column split,
# the default bench line, and the 2-rank rehearsal of bench.py's self-launch (gloo, both ranks on the one GPU).
# A step killed at its time limit ends the call (no further GPU step behind a hang).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2a; mkdir -p $O
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
step 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
step 400 python tools/shard_balance.py > $O/shard_balance.txt 2>&1; echo "balance rc=$?"; grep "world" $O/shard_balance.txt | grep -v rank
step 300 python bench.py > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"; tail -c 1500 $O/bench1.json
export LRN_BENCH_BACKEND=gloo LRN_BENCH_ONE_GPU=1
step 300 python bench.py --gpus 2 --steps 2 --warmup 1 --msz 512 --nvar 700 --no-cpu-baseline > $O/bench2_rehearsal.json 2> $O/bench2.err; echo "bench2 rc=$?"; tail -c 1200 $O/bench2_rehearsal.json; tail -5 $O/bench2.err
exit 0
