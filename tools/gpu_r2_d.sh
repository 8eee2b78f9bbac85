#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2d; mkdir -p $O
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in: $*"; exit 1; fi; return $rc; }
step 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
step 400 python tools/shard_balance.py --layouts=split --worlds=1,8,4,2 > $O/shard_balance.txt 2>&1; echo "balance rc=$?"; grep "^\[" $O/shard_balance.txt
step 300 python bench.py > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2d/bench1.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["phase_ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("scaled_instance",{}).get("value_scaled_ms"))
PY
exit 0
