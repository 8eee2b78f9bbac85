"""One seed of tools/fuzz_parity.py with traces of the oracle and both drivers (FUZZ_STRICT etc. as there)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuzz_parity as F
from loraine_jl_amd.optimizer import Optimizer
import loraine_jl_amd
from oracle import loraine_oracle as lo

seed = int(sys.argv[1])
A, b, d_lin, C_lin = F.random_problem(np.random.default_rng(seed))
opts = dict(kit=0, datarank=-1) if F.RANK1 else dict(kit=0)
om = lo.make_model([[m.copy() for m in blk] for blk in A], b.copy(), 0.0, None if d_lin is None else d_lin.copy(),
                   None if C_lin is None else C_lin.copy(), datarank=int(opts.get("datarank", 0)))
ref = lo.MySolver(om, dict(opts, verb=0)); lo.solve(ref)
print("oracle status", ref.status, "iter", ref.iter, "regcount", ref.regcount)
for t in ref.trace: print("  o", t["iter"], "%.10e %.3e reg %d adds %d" % (t["primal_obj"], t["dimacs"], t["regcount"], t["reg_adds"]))
d = loraine_jl_amd.Device(0)
if os.environ.get("FUZZ_STRICT"): d.set_option("pivot_boost", 0)
for resident in (True, False):
    o = Optimizer(resident=resident, device=d); o.set_silent(True)
    for k, v in opts.items(): o.set_attribute(k, v)
    o.load_model([[m.copy() for m in blk] for blk in A], b.copy(), 0.0, d_lin, C_lin, max_sense=False)
    try:
        o.optimize()
    except Exception as e:
        import traceback; traceback.print_exc()
    print("resident" if resident else "host", "status", o.solver.status, "iter", o.solver.iter, "regcount", o.solver.regcount)
    for t in o.solver.trace: print("  g", t["iter"], "%.10e %.3e reg %d adds %d" % (t["primal_obj"], t["dimacs"], t["regcount"], t["reg_adds"]))
