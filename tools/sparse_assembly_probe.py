"""Repeated kit=0 assemblies of a model whose constraints are sparse (branch 3 of makeBBBBsi, src/makeBBBB.jl:139-213, and
`_dot`, :39-64): the workload for the rocprofv3 passes of tools/profile_sparse.sh.
usage: sparse_assembly_probe.py tru9|vib9|c5 [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = loraine_jl_amd.Device(0)
if name == "c5":
    from loraine_jl_amd.synthetic import LowRankProblem
    model = LowRankProblem(10000, 20000, 4).model()
else:
    from loraine_jl_amd.model import model_from_sdpa
    model = model_from_sdpa(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"{name}.dat-s"))
dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, C_lin=model.C_lin if model.nlin else None)
rng = np.random.default_rng(1)
for i, m in enumerate(model.msizes):
    m = int(m)
    G = rng.standard_normal((m, m)) / np.sqrt(m) + np.eye(m)
    dev.set_scaling(i, G @ G.T, G)
if model.nlin:
    dev.set_lin(np.ones(model.nlin), np.ones(model.nlin))
dev.set_option("profile", 1)
dev.set_option("pair_lanes", int(os.environ.get("PAIR_LANES", "0")))
nnz = [int(a.nnz) for a in model.AA]
terms = 0
for a in model.AA:
    per = np.diff(a.tocsr().indptr).astype(np.float64)
    # pairs (i <= j) of constraints: sum over pairs of nnz_i * nnz_j
    terms += (per.sum() ** 2 + (per ** 2).sum()) / 2.0
dev.schur_assemble(0)
t0 = time.perf_counter()
dev.reset_timing()
for _ in range(reps):
    dev.schur_assemble(0)
dt = (time.perf_counter() - t0) / reps
print(f"{name} (pair_lanes={os.environ.get('PAIR_LANES', '0')}): nvar {model.n} msizes {list(map(int, model.msizes))} nnz(AA) {nnz} pair terms {terms:.3e}  assemble {dt*1e3:.3f} ms "
      f"(sparse {dev.timing('sparse')/reps:.3f} ms, lin {dev.timing('lin')/reps:.3f} ms) -> {terms/dt/1e9:.2f} G terms/s, "
      f"{terms*16/dt/1e12:.3f} TB/s of W gathers (2 x 8 B per term)", flush=True)
