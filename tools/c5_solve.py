"""C5: synthetic SDP with sparse constraints and a planted rank-r optimum, solved
with kit=1 (PCG) and the H_beta preconditioner on one MI355X, device-resident driver.
usage: c5_solve.py [msz nvar rank [maxit [budget_s]]]; the planted optimum b'y* is the known answer."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.stdout.reconfigure(line_buffering=True)
import loraine_jl_amd
from loraine_jl_amd import resident
from loraine_jl_amd.synthetic import LowRankProblem

msz = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nvar = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 4
maxit = int(sys.argv[4]) if len(sys.argv) > 4 else 100
budget = float(sys.argv[5]) if len(sys.argv) > 5 else 1e9
prec = int(os.environ.get("C5_PREC", "2"))
t0 = time.perf_counter()
P = LowRankProblem(msz, nvar, rank)
model = P.model()
t_gen = time.perf_counter() - t0
dev = loraine_jl_amd.Device(0)
for kv in filter(None, os.environ.get("LRN_OPTS", "").split(",")):      # e.g. LRN_OPTS=jacobi_inner=3,matvec_sparse=1
    k, v = kv.split("=")
    dev.set_option(k, float(v))
t0 = time.perf_counter()
opts = dict(kit=int(os.environ.get("C5_KIT", "1")), preconditioner=prec, erank=rank, verb=1, maxit=maxit, eDIMACS=float(os.environ.get("C5_EDIMACS", "1e-5")),
            tol_cg_min=float(os.environ.get("C5_TOL_CG_MIN", "1e-7")))
solver, ha = resident.load(model, opts, device=dev)
print("generate %.1f s, upload %.1f s" % (t_gen, time.perf_counter() - t0), flush=True)
solver.time_budget = budget
t0 = time.perf_counter()
solver.solve(ha)
wall = time.perf_counter() - t0
tr = solver.trace
by = float(model.b @ np.ravel(solver.y))
rec = dict(msz=msz, nvar=nvar, rank=rank, preconditioner=prec, iters=len(tr), status=solver.status, wall_s=wall,
           by=by, planted=P.optimum, rel_err=abs(by - P.optimum) / (1 + abs(P.optimum)),
           dimacs=tr[-1]["dimacs"] if tr else None, cg_total=solver.cg_iter_tot,
           cg_per_iter=[x["cg_pre"] + x["cg_cor"] for x in tr],
           ms_per_iter=float(np.mean([x["itertime"] for x in tr[1:]]) * 1e3) if len(tr) > 1 else None,
           gpu_ms={k: float(np.mean([x["gpu_ms"][k] for x in tr[1:]])) for k in tr[0]["gpu_ms"]} if len(tr) > 1 else None,
           find_step_ms=float(np.mean([x.get("find_step_ms", 0.0) for x in tr[1:]])) if len(tr) > 1 else None,
           svd_sweeps=[x["svd_sweeps"] for x in tr])
print(json.dumps(rec), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(rec, trace=[dict(x, errs=[float(e) for e in x["errs"]]) for x in tr]),
          open(f"gpurun_out/c5_solve_{msz}_{nvar}_p{prec}.json", "w"), indent=1)
