"""Clocks of the workgroups of one launch of gemm_f64_mid_kernel inside a maxG11 solve (LRN_MID_TRACE; 100 MHz ticks)."""
import os, sys, subprocess
import numpy as np
path = "/tmp/lrn_mid_trace.bin"
env = dict(os.environ, LRN_MID_TRACE=path)
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_times.py"), "--nocpu", "maxG11"], env=env,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
r = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
r = r[r[:, 7] > 0].astype(np.int64)
t0 = r[:, 0]; base = t0.min(); us = 0.01
print(f"{len(r)} workgroups; kernel span {(r[:, 7].max() - base) * us:.2f} us")
print(f"start of the workgroups after the first: median {np.median(t0 - base) * us:.2f} us, max {(t0.max() - base) * us:.2f}")
for k, name in ((2, "first K-tile landed (barrier 0 passed)"), (3, "barrier 1"), (4, "barrier 2"), (5, "barrier 3"), (6, "K loop done"), (7, "stores issued")):
    dt = (r[:, k] - t0) * us
    print(f"   {name:40s} since workgroup start: median {np.median(dt):6.2f} us, p10 {np.percentile(dt, 10):6.2f}, p90 {np.percentile(dt, 90):6.2f}, max {dt.max():6.2f}")
cu = ((r[:, 1] >> 8) & 0xff) | (((r[:, 1] >> 32) & 0xf) << 8)
cnt = np.bincount(np.unique(cu, return_inverse=True)[1])
print("workgroups per CU:", dict(zip(*np.unique(cnt, return_counts=True))), "CUs used:", len(cnt))
end = (r[:, 7] - base) * us
print(f"end of the workgroups since kernel start: median {np.median(end):.2f} us, p10 {np.percentile(end, 10):.2f}, max {end.max():.2f}")
