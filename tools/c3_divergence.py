"""where does the GPU trajectory of C3 (thetaG11, PCG + H_alpha) leave the oracle's?  Both run `maxit` iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
from loraine_jl_amd import solvers
from oracle import loraine_oracle as lo
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "thetaG11.dat-s")
opts = dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5, verb=0)
maxit = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref = lo.MySolver(lo.model_from_sdpa(path), dict(opts, maxit=maxit))
# instrument the oracle
orig_fs = lo.find_step
log_o = []
vec_o = []
def fs(s):
    orig_fs(s)
    vec_o.append((s.dely.copy(), None if s.predict else s.h_corr.copy()))
    log_o.append((bool(s.predict), float(s.alpha[0]), float(s.beta[0]), float(np.linalg.norm(s.dely)), float(s.mu), float(getattr(s, "sigma", 0.0)), float(s.tol_cg)))
lo.find_step = fs
lo.solve(ref)
for pe in (1, 0):
    d = loraine_jl_amd.Device(0); d.set_option("prec_eig", pe)
    o = Optimizer(resident=True, device=d); o.set_silent(True)
    for k, v in opts.items(): o.set_attribute(k, v)
    o.set_attribute("maxit", maxit)
    o.read_from_file(path)
    o._copy_to()
    s = o.solver
    log_g = []
    vec_g = []
    hc = [None]
    orig_rc = s.dev.ip_rhs_corr
    def rc(sm):
        out = orig_rc(sm); hc[0] = np.asarray(out).copy(); return out
    s.dev.ip_rhs_corr = rc
    orig = s.find_step
    def fsg():
        orig()
        vec_g.append((np.asarray(s.dely).copy(), None if s.predict else (s.Rp + hc[0])))
        log_g.append((bool(s.predict), float(s.alpha[0]), float(s.beta[0]), float(np.linalg.norm(s.dely)), float(s.mu), float(getattr(s, "sigma", 0.0)), float(s.tol_cg)))
    s.find_step = fsg
    solvers.solve(s, o.halpha)
    print("prec_eig", pe)
    for a, b in zip(log_o, log_g):
        print("  oracle pred=%d alpha %.12f beta %.12f |dely| %.10e mu %.6e sigma %.6e tol %.3e" % a)
        print("  gpu    pred=%d alpha %.12f beta %.12f |dely| %.10e mu %.6e sigma %.6e tol %.3e" % b)
    for k, (vo, vg) in enumerate(zip(vec_o, vec_g)):
        rd = np.linalg.norm(vo[0] - vg[0]) / np.linalg.norm(vo[0])
        rh = None if vo[1] is None else np.linalg.norm(vo[1] - vg[1]) / np.linalg.norm(vo[1])
        print(f"  step {k}: rel diff dely {rd:.3e}  rel diff h_corr {rh}")
    d.close()
