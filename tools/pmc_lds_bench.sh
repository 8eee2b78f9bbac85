#!/bin/bash
# round 4: LDS counters of the three assembly GEMMs (one counter pass of the bench command)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/p.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs=glob.glob("gpurun_out/pmc_lds/p/*/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"]
    if "gemm_f64" not in k: continue
    k=k[:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,cs in agg.items():
    print(k, "dispatches", len(n[k]))
    for c,v in sorted(cs.items()): print("   %-24s %.4g" % (c, v))
    if cs.get("SQ_LDS_IDX_ACTIVE"): print("   bank conflict cycles / LDS active cycles = %.3f" % (cs["SQ_LDS_BANK_CONFLICT"]/cs["SQ_LDS_IDX_ACTIVE"]))
PY
find $O -name "*counter_collection.csv" -size +20M -delete
