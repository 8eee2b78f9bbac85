"""Time and accuracy of the blocked Cholesky factorisation (lrn_dbg_potrf) at the sizes of the parity configs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
dev = loraine_jl_amd.Device(0)
dev.set_option("profile", 1)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [145, 800, 3240, 4000]:
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    A = G @ G.T + 0.01 * np.eye(n)
    dev.dbg_potrf(A)
    dev.reset_timing()
    L, info = dev.dbg_potrf(A)
    L = np.tril(L)
    print("n %5d info %d  %.3f ms   backward error %.2e" % (n, info, dev.timing("dbg_potrf"), np.abs(L @ L.T - A).max() / np.abs(A).max()), flush=True)
