"""Per-iteration trace of the resident driver with the eigen-free NT scaling (nt_mode 1) beside the SVD route (0).
Usage: python tools/nt_trace.py name [name ...]     names = files of tests/golden without .dat-s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
OPTS = {"maxG11": dict(kit=0, datarank=-1), "thetaG11": dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5)}
dev = loraine_jl_amd.Device(0)
for kv in filter(None, os.environ.get("LRN_OPTS", "").split(",")):
    k, v = kv.split("="); dev.set_option(k, float(v))
for name in sys.argv[1:]:
    tr = {}
    for mode in (0, 1):
        dev.set_option("nt_mode", mode)
        o = Optimizer(resident=True, device=dev); o.set_silent(True)
        for k, v in OPTS.get(name, dict(kit=0)).items():
            o.set_attribute(k, v)
        o.read_from_file(os.path.join(G, name + ".dat-s"))
        o.optimize()
        tr[mode] = o.solver.trace
        print(name, "mode", mode, "iters", len(tr[mode]), "status", o.termination_status(), "obj %.12g" % o.objective_value())
    for i in range(max(len(tr[0]), len(tr[1]))):
        a = tr[0][i] if i < len(tr[0]) else None
        b = tr[1][i] if i < len(tr[1]) else None
        s = "%3d " % (i + 1)
        if a: s += " svd: %.10e %.2e a=%s b=%s |" % (a["primal_obj"], a["dimacs"], np.round(a["alpha"], 4), np.round(a["beta"], 4))
        if a: s += " lz %d/%d fs %.2f |" % (a["lanczos_steps"], a["lanczos_runs"], a["find_step_ms"])
        if b: s += " lz %d/%d ct %d fs %.2f it %.2f" % (b["lanczos_steps"], b["lanczos_runs"], b["eigmin_chol_tests"], b["find_step_ms"], b["itertime"] * 1e3)
        if b: s += " ns: %.10e %.2e a=%s b=%s ns %d lyap %d fb %d/%d pw %.2f ly %.2f" % (
            b["primal_obj"], b["dimacs"], np.round(b["alpha"], 4), np.round(b["beta"], 4), b["ns_steps"], b["lyap_steps"],
            b["ns_fallback"], b["lyap_fallback"], b["gpu_ms"]["prepare_w"], b["lyap_ms"])
        print(s, flush=True)
