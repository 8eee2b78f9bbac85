#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2b; mkdir -p $O
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
step 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
step 400 python tools/shard_balance.py --layouts=split --worlds=1,8,4,2 > $O/shard_balance.txt 2>&1; echo "balance rc=$?"; grep -v amdgpu.ids $O/shard_balance.txt
exit 0
