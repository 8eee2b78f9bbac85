import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
dev = loraine_jl_amd.Device(0)
rng = np.random.default_rng(0)
for n in (64, 145, 400, 1000):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(rng.uniform(np.log(1e-5), np.log(1e6), n))
    A = (Q * lam) @ Q.T; A = 0.5 * (A + A.T)
    L, info = dev.dbg_potrf(A)
    L = np.tril(L)
    print(n, "info", info, "rel backward err", np.abs(L @ L.T - A).max() / np.abs(A).max(), "numpy", np.abs(np.linalg.cholesky(A) @ np.linalg.cholesky(A).T - A).max() / np.abs(A).max())
