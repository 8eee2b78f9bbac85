"""Summarise rocprofv3 --pmc CSVs (gpurun_out/prof_pmc*/...) per kernel into profiles/*.csv."""
import collections, csv, glob, os, sys

out_path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01_pmc_summary.csv"
rows_out = []
for d in sorted(glob.glob("gpurun_out/prof_pmc*/")):
    fs = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, cs in agg.items():
        if "gemm_f64" not in k and "potrf" not in k:
            continue
        for cname, vals in cs.items():
            rows_out.append(dict(pass_dir=d.rstrip("/").split("/")[-1], kernel=k, counter=cname, dispatches=len(vals),
                                 mean=sum(vals) / len(vals), mean_duration_ns=sum(dur[k]) / len(dur[k])))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=["pass_dir", "kernel", "counter", "dispatches", "mean", "mean_duration_ns"])
    w.writeheader()
    for r in rows_out:
        w.writerow(r)
# derived numbers for the dominant kernel
def get(kpat, cname):
    for r in rows_out:
        if kpat in r["kernel"] and r["counter"] == cname:
            return r
    return None
for k1, label in (("gemm_f64_kseg_lds_kernel<true>", "GEMM3'"), ("gemm_f64_lds_kernel<false>", "GEMM1'"),
                  ("gemm_f64_lds_kernel<true>", "GEMM2'")):
    mf, gui = get(k1, "SQ_VALU_MFMA_BUSY_CYCLES"), get(k1, "GRBM_GUI_ACTIVE")
    if mf and gui:
        cyc = gui["mean"] / 8.0
        print("%s: clock %.2f GHz, MFMA busy %.1f %%" % (label, cyc / gui["mean_duration_ns"], 100 * mf["mean"] / (cyc * 1024)))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        r = get(k1, c)
        if r:
            print("%s %s = %.1f MB per launch (raw counter, KB units)" % (label, c, r["mean"] / 1024.0))
