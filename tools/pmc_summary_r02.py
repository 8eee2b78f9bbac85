"""Summaries of tools/profile_r02.sh: per kernel and counter (pmc_summary.csv), L2 hit rates (l2_hit_rate.csv) and the
derived numbers (clock, MFMA busy, traffic) for the assembly GEMMs.  usage: pmc_summary_r02.py <out dir>"""
import collections, csv, glob, os, sys
O = sys.argv[1]
rows_out = []
for d in sorted(glob.glob(os.path.join(O, "prof_pmc*/"))):
    fs = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, cs in agg.items():
        if "gemm_f64" not in k and "potrf" not in k and "trsv" not in k:
            continue
        for cname, vals in cs.items():
            rows_out.append(dict(pass_dir=d.rstrip("/").split("/")[-1], kernel=k, counter=cname, dispatches=len(vals),
                                 mean=sum(vals) / len(vals), total=sum(vals), mean_duration_ns=sum(dur[k].values()) / len(dur[k]),
                                 total_duration_ns=sum(dur[k].values())))
with open(os.path.join(O, "pmc_summary.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=["pass_dir", "kernel", "counter", "dispatches", "mean", "total", "mean_duration_ns", "total_duration_ns"])
    w.writeheader()
    for r in rows_out:
        w.writerow(r)


def get(kpat, cname):
    for r in rows_out:
        if kpat in r["kernel"] and r["counter"] == cname:
            return r
    return None


with open(os.path.join(O, "l2_hit_rate.csv"), "w") as fh:
    fh.write("kernel,TCC_HIT_sum,TCC_MISS_sum,hit_rate\n")
    for k1 in ("gemm_f64_lds_kernel<false,", "gemm_f64_lds_kernel<true,", "gemm_f64_kseg_lds_kernel<true, 4, 4>",
               "gemm_f64_kseg_lds_kernel<true, 4, 5>"):
        h, m = get(k1, "TCC_HIT_sum"), get(k1, "TCC_MISS_sum")
        if h and m:
            fh.write('"%s",%.0f,%.0f,%.4f\n' % (k1, h["total"], m["total"], h["total"] / (h["total"] + m["total"])))
# per LAUNCH (the number of assemblies in a pass depends on the bench options: warm-up, steps, the `alongside` solve)
for k1, label in (("gemm_f64_kseg_lds_kernel<true, 4, 4>", "GEMM3' (leading rows, 128 x 128 tiles)"),
                  ("gemm_f64_kseg_lds_kernel<true, 4, 5>", "GEMM3' strip (last 160 rows, 128 x 160 tiles)"),
                  ("gemm_f64_lds_kernel<false,", "GEMM1'"),
                  ("gemm_f64_lds_kernel<true,", "GEMM2'")):
    mf, gui = get(k1, "SQ_VALU_MFMA_BUSY_CYCLES"), get(k1, "GRBM_GUI_ACTIVE")
    if mf and gui:
        cyc = gui["total"] / 8.0
        print("%s: %d launches in the pass, %.2f ms per launch, clock %.2f GHz, MFMA busy %.1f %% (over all its launches)" %
              (label, gui["dispatches"], gui["mean_duration_ns"] / 1e6, cyc / gui["total_duration_ns"], 100 * mf["total"] / (cyc * 1024)))
    f, wr = get(k1, "FETCH_SIZE"), get(k1, "WRITE_SIZE")
    if f and wr:
        launch_bytes = (2.0 * f["mean"] + wr["mean"]) * 1024.0
        print("%s: L2-miss traffic per launch (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.1f GB, %.2f TB/s" %
              (label, launch_bytes / 1e9, launch_bytes / f["mean_duration_ns"] / 1e3))
    h, m = get(k1, "TCC_HIT_sum"), get(k1, "TCC_MISS_sum")
    if h and m:
        print("%s: L2 hit rate %.1f %%" % (label, 100 * h["total"] / (h["total"] + m["total"])))
