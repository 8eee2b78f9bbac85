"""End-to-end solves of the BASELINE parity configs on the GPU path with per-iteration timing,
next to the CPU oracle on the host cores."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
from oracle import loraine_oracle as lo

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cases = [("theta1", dict(kit=0, eDIMACS=1e-6, initpoint=1, aamat=2), 0),
         ("maxG11", dict(kit=0, datarank=-1), -1),
         ("thetaG11", dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5), 0),
         ("tru9", dict(kit=0), 0), ("vib9", dict(kit=0), 0)]
only = [a for a in sys.argv[1:] if not a.startswith("--")]
out = {}
_dev = loraine_jl_amd.Device(0)
for kv in filter(None, os.environ.get("LRN_OPTS", "").split(",")):
    k, v = kv.split("="); _dev.set_option(k, float(v))
for name, opts, dr in cases:
    if only and name not in only:
        continue
    path = os.path.join(G, name + ".dat-s")
    o = Optimizer(resident="--host" not in sys.argv, device=_dev); o.set_silent(True)   # options are per context
    for k, v in opts.items():
        o.set_attribute(k, v)
    o.read_from_file(path)
    t = time.perf_counter(); o.optimize(); tg = time.perf_counter() - t
    tr = o.solver.trace
    its = len(tr)
    g = {k: float(np.mean([x["gpu_ms"][k] for x in tr[1:]])) for k in tr[0]["gpu_ms"]}
    host_it = float(np.mean([x["itertime"] for x in tr[1:]])) * 1e3
    rec = dict(iters=its, obj=o.objective_value(), status=o.termination_status(), wall_s=tg,
               ms_per_iter_total=host_it, gpu_ms=g, cg_tot=o.solver.cg_iter_tot,
               svd_sweeps=[x["svd_sweeps"] for x in tr])
    if "--cpu" in sys.argv or (name in ("theta1", "maxG11") and "--nocpu" not in sys.argv):
        t = time.perf_counter()
        s = lo.MySolver(lo.model_from_sdpa(path, datarank=dr), dict(opts, verb=0)); lo.solve(s)
        rec["cpu_wall_s"] = time.perf_counter() - t
        rec["cpu_iters"] = s.iter; rec["cpu_obj"] = lo.objective_value(s)
        rec["cpu_ms_assembly_solve"] = float(np.mean([x["t_assembly"] + x["t_solve"] for x in s.trace[1:]])) * 1e3
        rec["cpu_ms_prepw"] = float(np.mean([x["t_prepw"] for x in s.trace[1:]])) * 1e3
    print(name, json.dumps(rec), flush=True)
    out[name] = rec
json.dump(out, open(os.environ.get("E2E_OUT", "gpurun_out/e2e_times.json"), "w"), indent=1)
