"""C3 (thetaG11, kit=1, H_alpha, erank 1) with the dense and the pattern-restricted operator of the CG mat-vec
(option matvec_sparse: 1 dense GEMM route, 0 auto): ms per CG iteration, iterations, objective."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
from loraine_jl_amd import solvers
name = sys.argv[1] if len(sys.argv) > 1 else "thetaG11"
for mode in (1, 0, 1, 0):
    dev = loraine_jl_amd.Device(0)
    o = Optimizer(resident=True, device=dev); o.set_silent(True)
    for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5).items(): o.set_attribute(k, v)
    o.read_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".dat-s"))
    o._copy_to()
    dev.set_option("matvec_sparse", mode)
    dev.set_option("profile", 1)
    s = o.solver
    acc = dict(pcg=0.0, prec=0.0)
    orig = s.myIPstep
    def step(ha):
        dev.reset_timing()
        orig(ha)
        acc["pcg"] += dev.timing("pcg"); acc["prec"] += dev.timing("prec_setup")
    s.myIPstep = step
    t0 = time.perf_counter()
    solvers.solve(s, o.halpha)
    wall = time.perf_counter() - t0
    cg = s.cg_iter_tot if hasattr(s, "cg_iter_tot") else 0
    print(f"matvec_sparse {mode}: {s.iter} iterations, {cg} CG, wall {wall*1e3:.1f} ms = {wall*1e3/max(1,s.iter):.2f} per iteration; "
          f"pcg {acc['pcg']:.1f} ms = {1e3*acc['pcg']/max(1,cg):.1f} us per CG iteration; prec_setup {acc['prec']/max(1,s.iter):.2f} ms per iteration; "
          f"objective {o.objective_value():.10f}", flush=True)
    dev.close()
