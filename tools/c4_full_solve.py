"""Full interior-point solve of the synthetic dense SDP on one MI355X with the device-resident
driver: per-iteration GPU timings of every phase (assembly + solve = the BASELINE metric;
prepare_W, find_step, RHS alongside)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from loraine_jl_amd.synthetic import synthetic_dense_solver

msz = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nvar = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
maxit = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = loraine_jl_amd.Device(0)
t0 = time.perf_counter()
solver, ha = synthetic_dense_solver(dev, msz, nvar, options=dict(kit=0, verb=1, maxit=maxit))
print("setup %.1f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
solver.solve(ha)
wall = time.perf_counter() - t0
tr = solver.trace
rec = dict(msz=msz, nvar=nvar, iters=len(tr), status=solver.status, wall_s=wall,
           primal=tr[-1]["primal_obj"], dual=tr[-1]["dual_obj"], dimacs=tr[-1]["dimacs"],
           ms_per_iter=float(np.mean([x["itertime"] for x in tr[1:]]) * 1e3),
           gpu_ms={k: float(np.mean([x["gpu_ms"][k] for x in tr[1:]])) for k in tr[0]["gpu_ms"]},
           find_step_ms=float(np.mean([x.get("find_step_ms", 0.0) for x in tr[1:]])),
           svd_sweeps=[x["svd_sweeps"] for x in tr],
           schur_chol=[x["schur_chol"] for x in tr], wchol_fail=[x["wchol_fail"] for x in tr])
print(json.dumps(rec), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(rec, trace=[{k: v for k, v in x.items() if k != "errs"} for x in tr]),
          open(f"gpurun_out/c4_full_solve_{msz}_{nvar}.json", "w"), indent=1)
