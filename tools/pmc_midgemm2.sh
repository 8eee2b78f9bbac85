#!/bin/bash
# round 4: L2 hit rate and L2-miss traffic of the mid-size (msz 800) product kernels inside a maxG11 solve
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/mid/pmc; rm -rf $O; mkdir -p $O
for m in 1 0; do
  export LRN_GEMM_MID=$m
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/h$m -- python3 tools/e2e_times.py --nocpu maxG11 > $O/h$m.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$m -- python3 tools/e2e_times.py --nocpu maxG11 > $O/f$m.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in ("h1","f1","h0","f0"):
    fs=glob.glob(f"gpurun_out/mid/pmc/{d}/*/*counter_collection.csv")
    if not fs: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "gemm_f64_kernel<64, 64, false, false" not in k and "gemm_f64_mid" not in k and "reduce_slabs_kernel" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k][r["Dispatch_Id"]]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k,cs in agg.items():
        print(d, k[:50], "dispatches", len(dur[k]), "avg us %.1f"%(sum(dur[k].values())/len(dur[k])/1e3))
        for c,v in cs.items(): print("   %-28s mean %.4g"%(c, sum(v)/len(v)))
PY
