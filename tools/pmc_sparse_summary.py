"""Per-kernel summary of tools/profile_sparse.sh: duration (kernel trace), L2 hit rate (TCC_HIT / (HIT + MISS)) and
L2-miss read traffic (2 x FETCH_SIZE x 1024 B: gfx950 reports half the bytes of wide loads; narrower gathers are
uncalibrated -- read it as an upper bound of the fabric bytes, MI355X_MICROARCH.md section HBM)."""
import collections, csv, glob, os, sys
O = sys.argv[1]
KERNELS = ("pair_wave_kernel", "pair_thread_kernel", "dense_sparse_gather_kernel", "lin_schur_kernel")


def counters(d):
    fs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    if not fs:
        return agg, dur
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, dur


print("model,kernel,launches_per_assembly,avg_us,total_ms_per_assembly,l2_hit_rate,fetch_MB_per_launch(2xFETCH_SIZE),fetch_GBps")
for m in ("tru9", "vib9", "c5"):
    st = sorted(glob.glob(os.path.join(O, f"stats_{m}", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    stats = {}
    if st:
        for r in csv.DictReader(open(st[-1])):
            stats[r["Name"]] = r
    tcc, _ = counters(os.path.join(O, f"tcc_{m}"))
    fe, fdur = counters(os.path.join(O, f"fetch_{m}"))
    for name, r in stats.items():
        if not any(k in name for k in KERNELS):
            continue
        calls = int(r["Calls"]); avg_ns = float(r["AverageNs"])
        hit = miss = None
        for k, cs in tcc.items():
            if k == name and "TCC_HIT_sum" in cs:
                hit, miss = sum(cs["TCC_HIT_sum"]), sum(cs["TCC_MISS_sum"])
        fetch_mb = gbps = None
        for k, cs in fe.items():
            if k == name and "FETCH_SIZE" in cs:
                fetch_mb = 2.0 * (sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])) * 1024.0 / 1e6
                d = sum(fdur[k].values()) / max(1, len(fdur[k]))
                gbps = fetch_mb * 1e6 / d if d else None
        print(f"{m},{name.split('(')[0]},{calls / 6.0:.1f},{avg_ns / 1e3:.1f},{calls * avg_ns / 6.0 / 1e6:.3f},"
              f"{'' if hit is None else round(hit / max(1.0, hit + miss), 4)},{'' if fetch_mb is None else round(fetch_mb, 2)},"
              f"{'' if gbps is None else round(gbps, 1)}")
