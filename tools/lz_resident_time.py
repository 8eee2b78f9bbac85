"""Time per Lanczos step of a single eigmin search (lrn_dbg_eigmin, upload of M included in both) with launched and with
resident steps (option lz_resident), n = 801, spectra that take ~100 / ~400 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
dev = loraine_jl_amd.Device(0)
n = 801
rng = np.random.default_rng(1)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
for label, lam in (("wide", rng.standard_normal(n)),
                   ("clustered", np.concatenate([[-1.0], -1.0 + 1e-3 * rng.random(n // 4), np.linspace(0.5, 40.0, n - 1 - n // 4)]))):
    M = (Q * lam) @ Q.T
    M = (M + M.T) / 2
    for res in (0, 1, 0, 1):
        dev.set_option("lz_resident", res)
        dev.dbg_eigmin(M)
        t0 = time.perf_counter()
        for _ in range(5):
            val, steps = dev.dbg_eigmin(M)
        dt = (time.perf_counter() - t0) / 5
        print(f"{label}: lz_resident={res}: {steps} steps, {dt * 1e3:.3f} ms per search, lambda_min {val:.12e}")
# upload alone
import ctypes
t0 = time.perf_counter()
for _ in range(5):
    dev.dbg_eigmin(np.eye(n))
print(f"identity (1 batch): {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per search")
