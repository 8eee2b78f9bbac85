"""Generates tests/golden/iterates_theta1.npz from the CPU oracle (committed together with this
script): the iterate (X, S, y) after 3 IP iterations of theta1 and the hot-path outputs the
oracle computes from it (W, D, lower triangle of H, makeRHS h, Cholesky solve dely).  The GPU
parity tests compare the C-ABI results against these vectors without running the oracle."""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loraine_oracle as lo  # noqa: E402

path = os.path.join(ROOT, "tests", "golden", "theta1.dat-s")
model = lo.model_from_sdpa(path)
s = lo.MySolver(model, dict(kit=0, eDIMACS=1e-6, initpoint=1, aamat=2, verb=0, maxit=3))
lo.solve(s)
X, S, y = s.X[0].copy(), s.S[0].copy(), s.y.copy()
# hot path on that iterate
lo.find_mu(s)
lo.prepare_W(s)
W, D = s.W[0].copy(), s.D[0].copy()
H = lo.makeBBBBs(model.n, 1, model.A, model.AA, s.W, model.qA, model.sigmaA)
H = np.tril(H)
Rp = model.b - model.AA[0] @ lo.vec(X)
Rd = model.C[0].toarray() - S - lo.mat(model.AA[0].T @ y)
h = lo.makeRHS(1, model.AA, s.W, s.S, Rp, [Rd])
Hs = H + np.tril(H, -1).T
L = np.linalg.cholesky(Hs)
dely = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)
out = os.path.join(ROOT, "tests", "golden", "iterates_theta1.npz")
np.savez_compressed(out, X=X, S=S, y=y, W=W, D=np.sort(D), H_lower=H, Rp=Rp, Rd=Rd, h=h, dely=dely)
print("wrote", out, os.path.getsize(out), "bytes")
