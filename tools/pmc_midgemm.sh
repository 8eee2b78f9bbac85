#!/bin/bash
# round 4: counters of the mid-size (msz 800) product kernel inside a maxG11 solve
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/midgemm; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p1 -- python3 tools/e2e_times.py --nocpu maxG11 > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/p2 -- python3 tools/e2e_times.py --nocpu maxG11 > $O/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("p1","p2"):
    fs=glob.glob(f"gpurun_out/r04/midgemm/{d}/*/*counter_collection.csv")
    if not fs: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "gemm_f64_kernel<64, 64, false, false" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k][r["Dispatch_Id"]]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k,cs in agg.items():
        print(k[:60], "dispatches", len(dur[k]), "avg us %.1f"%(sum(dur[k].values())/len(dur[k])/1e3))
        for c,v in cs.items(): print("   %-28s mean %.4g"%(c, sum(v)/len(v)))
PY
