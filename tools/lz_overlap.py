"""Timeline of the Lanczos step kernels inside a solve, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/e2e_times.py --nocpu maxG11
    python3 tools/lz_overlap.py $(find gpurun_out/trace -name "*kernel_trace.csv") [kernel-name-substring]

Prints, for the chosen kernel (default lz_fused_kernel): launches, mean duration, the gap between consecutive launches of
one queue (end -> next start), how much of the kernels' time on one queue is overlapped by the same kernel on another
queue (the two runs of eigmin_dev_pair live on two streams), and the share of the kernel's busy span in which the GPU
ran nothing at all."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "lz_fused_kernel"
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    sel = [r for r in rows if pat in r[3]]
    if not sel:
        print("no kernel matches", pat)
        return
    dur = [e - s for s, e, _, _ in sel]
    print(f"{pat}: {len(sel)} launches, mean {sum(dur) / len(dur) / 1e3:.2f} us, min {min(dur) / 1e3:.2f}, max {max(dur) / 1e3:.2f}")
    byq = defaultdict(list)
    for s, e, q, _ in sel:
        byq[q].append((s, e))
    for q, lst in sorted(byq.items()):
        gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        chain = sorted(g for g in gaps if g < 30000)          # consecutive steps of one batch
        if chain:
            print(f"  queue {q}: {len(lst)} launches; gap to the next launch inside a batch: median {chain[len(chain) // 2] / 1e3:.2f} us, "
                  f"p90 {chain[int(0.9 * len(chain))] / 1e3:.2f} us ({len(chain)} gaps < 30 us; {len(gaps) - len(chain)} longer)")
    # overlap between queues
    qs = sorted(byq)
    if len(qs) >= 2:
        a, b = byq[qs[0]], byq[qs[1]]
        j = 0
        ov = 0
        for s, e in a:
            while j < len(b) and b[j][1] <= s:
                j += 1
            k = j
            while k < len(b) and b[k][0] < e:
                ov += min(e, b[k][1]) - max(s, b[k][0])
                k += 1
        ta = sum(e - s for s, e in a)
        tb = sum(e - s for s, e in b)
        print(f"  overlap of the two queues: {ov / 1e3:.0f} us of {ta / 1e3:.0f} + {tb / 1e3:.0f} us of kernel time "
              f"({100.0 * ov / max(1, min(ta, tb)):.1f} % of the shorter)")
    # idle inside the stretches where this kernel runs back to back (any queue): union of ALL kernels vs wall
    stretches = []
    cur_s, cur_e = sel[0][0], sel[0][1]
    for s, e, _, _ in sel[1:]:
        if s - cur_e < 30000:
            cur_e = max(cur_e, e)
        else:
            stretches.append((cur_s, cur_e))
            cur_s, cur_e = s, e
    stretches.append((cur_s, cur_e))
    wall = sum(e - s for s, e in stretches)
    busy = 0
    i = 0
    for ss, se in stretches:
        while i < len(rows) and rows[i][1] <= ss:
            i += 1
        k = i
        last = ss
        while k < len(rows) and rows[k][0] < se:
            s, e = max(rows[k][0], last), min(rows[k][1], se)
            if e > s:
                busy += e - s
                last = e
            k += 1
    print(f"  {len(stretches)} stretches of back-to-back launches: wall {wall / 1e3:.0f} us, some kernel running {busy / 1e3:.0f} us "
          f"({100.0 * busy / max(1, wall):.1f} %), per launch {wall / len(sel) / 1e3:.2f} us of wall")


if __name__ == "__main__":
    main()
