import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
dev = loraine_jl_amd.Device(0)
import scipy.sparse as sp
# launch overhead microbench
from oracle import loraine_oracle as lo
A = [[sp.csc_matrix((4, 4)), sp.identity(4, format="csc")]]
model = lo.make_model(A, np.ones(1), 0.0, None, None)
dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
dev.set_scaling(0, np.eye(4), np.eye(4)); dev.schur_assemble(0)
t = time.perf_counter()
for _ in range(2000): dev.schur_add_diag(0.0)
import torch; torch.cuda.synchronize()
print("add_diag call: %.1f us" % ((time.perf_counter() - t) / 2000 * 1e6))
for n, cross in ((800, 0), (800, 1), (2000, 0), (2000, 1), (3000, 0), (3000, 1)):
    dev.set_option("jacobi_cross", cross)
    rng = np.random.default_rng(n)
    Mx = rng.standard_normal((n, n)) @ np.diag(np.logspace(0, -4, n)) @ rng.standard_normal((n, n))
    for rep in range(2):
        t = time.perf_counter()
        US, s, V, sw = dev.dbg_svd_jacobi(Mx)
        dt = time.perf_counter() - t
    sref = np.linalg.svd(Mx, compute_uv=False)
    print(f"n={n} cross={cross} sweeps={sw} wall={dt*1e3:.1f} ms  sv relerr={np.max(np.abs(np.sort(s)[::-1]-sref)/sref):.2e}", flush=True)
