"""print (not assert) the deviations of the GPU path from the C2 / C3 golden fixtures -- to set the tolerances of
tests/test_gpu_golden_c2_c3.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import loraine_jl_amd
from test_gpu_golden_c2_c3 import _iterate, relerr, GOLD
for prec_eig in (1, 0):
    if not os.path.exists(os.path.join(GOLD, "iterate_thetaG11.npz")): break
    dev = loraine_jl_amd.Device(0)
    g, model, m, X, S, Rd = _iterate("thetaG11", 0)
    dev.set_option("prec_eig", prec_eig)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, out = dev.prepare_w(0, X, S)
    V = g["probes"]
    print("prec_eig", prec_eig, "W", relerr(out["W"] @ V, g["W_probe"]), "MyA", relerr(dev.matvec(g["x"]), g["MyA_x"]))
    dev.prec_setup(1, 1, 1)
    print("  MyM", relerr(dev.prec_apply(g["x"]), g["MyM_x"]))
    for tol, xs, ec, it in zip(g["cg_tols"], g["cg_x"], g["cg_exit"], g["cg_iters"]):
        x, exit_code, iters = dev.pcg(g["h"], float(tol))
        print("  cg tol", tol, "oracle", int(ec), int(it), "gpu", exit_code, iters, "rel diff", relerr(x, xs))
    dev.close()
from loraine_jl_amd.optimizer import Optimizer
for name, opts, pe in (("maxG11", dict(kit=0, datarank=-1), 0), ("thetaG11", dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5), 0),
                       ("thetaG11", dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5), 1)):
    f = os.path.join(GOLD, f"trace_{name}.json")
    if not os.path.exists(f): continue
    tr = json.load(open(f))
    d = loraine_jl_amd.Device(0); d.set_option("prec_eig", pe)
    print("prec_eig", pe)
    o = Optimizer(resident=True, device=d); o.set_silent(True)
    for k, v in opts.items(): o.set_attribute(k, v)
    o.read_from_file(os.path.join(GOLD, f"{name}.dat-s")); o.optimize()
    print(name, "iters", o.solver.iter, tr["iterations"], "obj", o.objective_value(), tr["objective"])
    for k, t in enumerate(o.solver.trace):
        if k >= len(tr["primal"]): break
        print(f"  it {k}: primal rel {abs(t['primal_obj']-tr['primal'][k])/max(abs(tr['primal'][k]),1e-300):.2e} dual rel {abs(t['dual_obj']-tr['dual'][k])/max(abs(tr['dual'][k]),1e-300):.2e} "
              f"dimacs {t['dimacs']:.3e} vs {tr['dimacs'][k]:.3e} cg {t['cg_pre']},{t['cg_cor']} vs {tr['cg_pre'][k]},{tr['cg_cor'][k]}")
