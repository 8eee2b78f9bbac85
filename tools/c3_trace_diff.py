"""thetaG11 (C3) resident solve vs the committed oracle trace: relative differences of the first iterations
(the truncated-CG trajectory amplifies rounding ~100x per IP iteration).
usage: c3_trace_diff.py [key=value ...]   (library options, e.g. matvec_h=1 pcg_lookahead=0)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = loraine_jl_amd.Device(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    dev.set_option(k, float(v))
o = Optimizer(resident=True, device=dev)
o.set_silent(True)
for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5).items():
    o.set_attribute(k, v)
o.read_from_file(os.path.join(GOLD, "thetaG11.dat-s"))
o.optimize()
tr = json.load(open(os.path.join(GOLD, "trace_thetaG11.json")))
print("iters", o.solver.iter, tr["iterations"], "cg", o.solver.cg_iter_tot, tr["cg_total"], "obj", o.objective_value(), tr["objective"])
for k in range(min(5, len(o.solver.trace))):
    t = o.solver.trace[k]
    print(k, (t["cg_pre"], t["cg_cor"]), (tr["cg_pre"][k], tr["cg_cor"][k]),
          "primal rel %.2e" % (abs(t["primal_obj"] - tr["primal"][k]) / abs(tr["primal"][k])),
          "dual rel %.2e" % (abs(t["dual_obj"] - tr["dual"][k]) / abs(tr["dual"][k])))
print("hop_assemble", dev.count("hop_assemble"), "hop_matvec", dev.count("hop_matvec"), "matvec", dev.count("matvec"))
