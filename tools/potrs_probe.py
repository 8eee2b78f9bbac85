"""repeated L'\(L\h) at nvar = 4000 (the C4 Schur matrix size) for rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = loraine_jl_amd.Device(0)
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n + 3)); A = M @ M.T + n * 1e-3 * np.eye(n); b = rng.standard_normal(n)
for _ in range(6):
    x, info = dev.dbg_potrs(A, b)
print("resid", np.linalg.norm(A @ x - b) / np.linalg.norm(b))
