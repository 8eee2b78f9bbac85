#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2c; mkdir -p $O
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
step 300 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_schur.py -x -q > $O/pytest_a.log 2>&1; rc=$?; echo "pytest blocks+schur rc=$rc"; tail -3 $O/pytest_a.log
[ $rc -ne 0 ] && exit 0
step 400 python tools/shard_balance.py --layouts=split --worlds=1,8 > $O/shard_balance.txt 2>&1; echo "balance rc=$?"; grep -v amdgpu.ids $O/shard_balance.txt
step 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_blocks.py --deselect tests/test_gpu_schur.py > $O/pytest_b.log 2>&1; echo "pytest rest rc=$?"; tail -5 $O/pytest_b.log
exit 0
