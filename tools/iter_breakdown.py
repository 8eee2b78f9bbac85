"""Per-iteration time breakdown of a resident solve of an SDPA file (gpu_ms phases, find_step, Lyapunov, the rest).

    python3 tools/iter_breakdown.py maxG11 [key=value,key=value]      # library options (lrn_set_option), e.g. eigmin_pair=1
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
name = sys.argv[1] if len(sys.argv) > 1 else "maxG11"
opts = dict(kit=0, datarank=-1) if name == "maxG11" else (dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5) if name == "thetaG11" else dict(kit=0))
dev = loraine_jl_amd.Device(0)
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):
    dev.set_option(kv.split("=")[0], float(kv.split("=")[1]))
o = Optimizer(device=dev, resident=True); o.set_silent(True)
for k, v in opts.items(): o.set_attribute(k, v)
o.read_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".dat-s"))
o.optimize()
for x in o.solver.trace:
    g = x["gpu_ms"]
    known = sum(g.values()) + x.get("find_step_ms", 0.0) + x.get("lyap_ms", 0.0)
    print("it %2d total %6.2f | %s | find_step %5.2f lyap %5.2f (%d) | other %5.2f | lanczos %s in %s runs" % (
        x["iter"], x["itertime"] * 1e3, " ".join("%s %.2f" % (k[:6], v) for k, v in g.items() if v), x.get("find_step_ms", 0.0),
        x.get("lyap_ms", 0.0), x.get("lyap_steps", 0), x["itertime"] * 1e3 - known, x.get("lanczos_steps"), x.get("lanczos_runs")),
        "| prec lanczos %s plain %s dense %s | cg %d+%d hop %s/%s" % (x.get("prec_lanczos_steps"), x.get("lanczos_plain"), x.get("prec_dense_build"),
                                                                   x["cg_pre"], x["cg_cor"], x.get("hop_assemble"), x.get("hop_matvec")))
