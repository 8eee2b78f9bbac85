"""thetaG11 (C3: kit=1, H_alpha, erank 1) at tightened tolerances on the GPU path; the oracle's counterpart is
`oracle/make_golden.py thetaG11_tight` (TIGHT_EDIMACS / TIGHT_TOL_CG_MIN)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
name = sys.argv[1] if len(sys.argv) > 1 else "thetaG11"
for ed, tc in ((1e-5, 1e-7), (1e-6, 1e-8), (1e-7, 1e-9), (1e-8, 1e-10)):
    o = Optimizer(resident=True); o.set_silent(True)
    for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=ed, tol_cg_min=tc).items(): o.set_attribute(k, v)
    o.read_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".dat-s"))
    try:
        o.optimize()
        s = o.solver
        print(json.dumps(dict(name=name, eDIMACS=ed, tol_cg_min=tc, status=s.status, iters=s.iter, cg=s.cg_iter_tot,
                              objective=o.objective_value(), dual=o.dual_objective_value(), dimacs=s.trace[-1]["dimacs"])), flush=True)
    except Exception as e:
        print(json.dumps(dict(name=name, eDIMACS=ed, tol_cg_min=tc, error=repr(e)[:200])), flush=True)
