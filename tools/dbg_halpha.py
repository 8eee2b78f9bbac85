import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from loraine_jl_amd import resident
from loraine_jl_amd.synthetic import LowRankProblem
dev = loraine_jl_amd.Device(0)
P = LowRankProblem(60, 120, 4, seed=7)
opts = dict(kit=1, preconditioner=1, erank=4, verb=1, eDIMACS=1e-5)
solver, ha = resident.load(P.model(), opts, device=dev)
_ps, _pcg = dev.prec_setup, dev.pcg
def ps(*a):
    r = _ps(*a); print("   prec_setup", a, "->", r, "z_shift", dev.count("prec_z_shift"), "lz", dev.count("prec_lanczos_steps")); return r
def pcg(h, tol, maxit=10000):
    x, ec, it = _pcg(h, tol, maxit); print("   pcg tol %.1e -> exit %d iters %d |h| %.3e |x| %.3e" % (tol, ec, it, np.linalg.norm(h), np.linalg.norm(x))); return x, ec, it
dev.prec_setup, dev.pcg = ps, pcg
try:
    solver.solve(ha)
    print("status", solver.status)
except Exception as e:
    print("raised", e, "iter", solver.iter, "z_shift", dev.count("prec_z_shift"))
    X, S = dev.ip_get_iterate(0)
    print("eig X", np.linalg.eigvalsh(X)[[0, 1, -2, -1]], "eig S", np.linalg.eigvalsh(S)[[0, 1, -2, -1]])
    info, out = dev.prepare_w(0, X, S)
    W = out["W"]; lam = np.linalg.eigvalsh(W)
    print("eig W", lam[[0, 1, 2, -5, -4, -3, -2, -1]], "cond", lam[-1] / lam[0])
