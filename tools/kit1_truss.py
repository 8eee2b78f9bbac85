import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from loraine_jl_amd.optimizer import Optimizer
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ["tru9", "vib9"]:
    for opts in [dict(kit=1, preconditioner=2, erank=1, eDIMACS=1e-5), dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5)]:
        o = Optimizer(); o.set_silent(True)
        for k, v in opts.items(): o.set_attribute(k, v)
        o.read_from_file(os.path.join(G, name + ".dat-s"))
        t = time.time()
        try:
            o.optimize(); print(name, opts, o.termination_status(), o.solver.iter, "%.9f" % o.objective_value(), "cg", o.solver.cg_iter_tot, "%.1fs" % (time.time() - t), flush=True)
        except Exception as e:
            print(name, opts, "raised", repr(e)[:120], "iter", o.solver.iter, flush=True)
