"""Per-kernel HBM traffic of the kit=1 runs from the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in
KB; the guide's rule for gfx950: HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024)."""
import collections, csv, glob, os
out = []
for cfg in ("c5", "c3"):
    data = collections.defaultdict(lambda: dict(FETCH_SIZE=[], WRITE_SIZE=[], dur=[]))
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = sorted(glob.glob(f"gpurun_out/pmc_{cfg}_{cnt}/*/*counter_collection.csv"), key=os.path.getmtime)
        if not fs:
            continue
        seen = set()
        for r in csv.DictReader(open(fs[-1])):
            if r["Counter_Name"] != cnt:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            data[k][cnt].append(float(r["Counter_Value"]))
            if cnt == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                data[k]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    rows = []
    for k, d in data.items():
        if not d["FETCH_SIZE"] or not d["WRITE_SIZE"] or not d["dur"]:
            continue
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        t = sum(d["dur"]) / len(d["dur"])
        hbm = (2 * f + w) * 1024
        rows.append((sum(d["dur"]), cfg, k, len(d["dur"]), t / 1e3, f / 1024, w / 1024, hbm / 1e6, hbm / t))
    rows.sort(reverse=True)
    out += rows[:14]
os.makedirs("profiles", exist_ok=True)
OUT = os.environ.get("KIT1_OUT", "gpurun_out/kit1_hbm_traffic.csv")
with open(OUT, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["config", "kernel", "launches", "avg_us", "FETCH_SIZE_MB_raw", "WRITE_SIZE_MB_raw", "hbm_MB_per_launch(2F+W)", "hbm_GB_per_s"])
    for r in out:
        w.writerow([r[1], r[2], r[3], f"{r[4]:.1f}", f"{r[5]:.2f}", f"{r[6]:.2f}", f"{r[7]:.2f}", f"{r[8]:.1f}"])
        print(r[1], r[2][:44].ljust(44), r[3], f"{r[4]:9.1f} us  hbm {r[7]:10.2f} MB  {r[8]:8.1f} GB/s")
