"""Generates tests/golden/iterates_theta1.npz from the CPU oracle (committed together with this
script): the iterate (X, S, y) after 3 IP iterations of theta1 and the hot-path outputs the
oracle computes from it (W, D, lower triangle of H, makeRHS h, Cholesky solve dely).  The GPU
parity tests compare the C-ABI results against these vectors without running the oracle."""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loraine_oracle as lo  # noqa: E402

def golden_theta1():
    path = os.path.join(ROOT, "tests", "golden", "theta1.dat-s")
    model = lo.model_from_sdpa(path)
    s = lo.MySolver(model, dict(kit=0, eDIMACS=1e-6, initpoint=1, aamat=2, verb=0, maxit=3))
    lo.solve(s)
    X, S, y = s.X[0].copy(), s.S[0].copy(), s.y.copy()
    # hot path on that iterate
    lo.find_mu(s)
    lo.prepare_W(s)
    W, D = s.W[0].copy(), s.D[0].copy()
    H = lo.makeBBBBs(model.n, 1, model.A, model.AA, s.W, model.qA, model.sigmaA)
    H = np.tril(H)
    Rp = model.b - model.AA[0] @ lo.vec(X)
    Rd = model.C[0].toarray() - S - lo.mat(model.AA[0].T @ y)
    h = lo.makeRHS(1, model.AA, s.W, s.S, Rp, [Rd])
    Hs = H + np.tril(H, -1).T
    L = np.linalg.cholesky(Hs)
    dely = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)
    out = os.path.join(ROOT, "tests", "golden", "iterates_theta1.npz")
    np.savez_compressed(out, X=X, S=S, y=y, W=W, D=np.sort(D), H_lower=H, Rp=Rp, Rd=Rd, h=h, dely=dely)
    print("wrote", out, os.path.getsize(out), "bytes")


# ======================================================================================================
# BASELINE configs C2 (maxG11, kit=0, datarank=-1) and C3 (thetaG11, kit=1, H_alpha, erank=1): the halves
# of the path the reference's own tests never run pinned at REAL iterates of the
# CPU oracle, plus the per-iteration trace of the whole solves.  Inputs (X, S) are the oracle's iterate
# rounded to float32 -- still a strictly feasible interior point, and a quarter of the bytes (lower
# triangles stored); every output is computed by the oracle in float64 FROM THE ROUNDED INPUTS.  Large
# outputs are stored as digests: products with fixed seeded vectors, diagonals, sampled entries.
def _sym_from_f32_lower(M):
    L = np.tril(M).astype(np.float32).astype(np.float64)
    return L + np.tril(L, -1).T


def _pack_lower_f32(M):
    return M[np.tril_indices(M.shape[0])].astype(np.float32)


def _probe_vectors(n, k=3, seed=2024):
    return np.random.default_rng(seed).standard_normal((n, k))


def _trace_of(solver):
    return dict(primal=[t["primal_obj"] for t in solver.trace], dual=[t["dual_obj"] for t in solver.trace],
                dimacs=[t["dimacs"] for t in solver.trace], cg_pre=[t["cg_pre"] for t in solver.trace],
                cg_cor=[t["cg_cor"] for t in solver.trace], errs=[list(t["errs"]) for t in solver.trace])


def golden_maxG11(after_iterations=6):
    import json
    path = os.path.join(ROOT, "tests", "golden", "maxG11.dat-s")
    model = lo.model_from_sdpa(path, datarank=-1)
    opts = dict(kit=0, datarank=-1, verb=0)
    s = lo.MySolver(model, dict(opts, maxit=after_iterations))
    lo.solve(s)
    X, S, y = _sym_from_f32_lower(s.X[0]), _sym_from_f32_lower(s.S[0]), s.y.copy()
    s.X[0], s.S[0] = X.copy(), S.copy()
    lo.find_mu(s)
    lo.prepare_W(s)
    W, D, G = s.W[0], s.D[0], s.G[0]
    H = lo.makeBBBB_rank1(model.n, 1, model.B, s.G)                      # src/makeBBBB.jl:1-20
    Hg = lo.makeBBBBs(model.n, 1, model.A, model.AA, s.W, model.qA, model.sigmaA)   # general path, same data
    H = np.tril(H)
    Hs = H + np.tril(H, -1).T
    assert np.linalg.norm(np.tril(Hg) - H) <= 1e-12 * np.linalg.norm(H)
    Rp = model.b - model.AA[0] @ lo.vec(X)
    Rd = model.C[0].toarray() - S - lo.mat(model.AA[0].T @ y)
    h = lo.makeRHS(1, model.AA, s.W, [S], Rp, [Rd])
    L = np.linalg.cholesky(Hs)
    dely = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)
    V = _probe_vectors(model.n)
    rng = np.random.default_rng(7)
    ii = rng.integers(0, model.n, 4000)
    jj = rng.integers(0, model.n, 4000)
    ii, jj = np.maximum(ii, jj), np.minimum(ii, jj)
    out = os.path.join(ROOT, "tests", "golden", "iterate_maxG11.npz")
    np.savez_compressed(out, X_lower_f32=_pack_lower_f32(X), S_lower_f32=_pack_lower_f32(S), y=y, after=after_iterations,
                        D_sorted=np.sort(D), W_diag=np.diag(W).copy(), W_probe=W @ V[: W.shape[0]], probes=V,
                        H_diag=np.diag(Hs).copy(), H_probe=Hs @ V, H_fro=np.linalg.norm(Hs),
                        H_sample_i=ii, H_sample_j=jj, H_sample=Hs[ii, jj], Rp=Rp, h=h, dely=dely)
    # whole solve
    full = lo.MySolver(lo.model_from_sdpa(path, datarank=-1), opts)
    lo.solve(full)
    tr = dict(_trace_of(full), status=int(full.status), iterations=int(full.iter), options=opts,
              objective=lo.objective_value(full), dual_objective=lo.dual_objective_value(full))
    with open(os.path.join(ROOT, "tests", "golden", "trace_maxG11.json"), "w") as f:
        json.dump(tr, f)
    print("wrote", out, os.path.getsize(out), "bytes;", full.iter, "iterations, objective", tr["objective"])


def golden_thetaG11(after_iterations=5):
    import json
    path = os.path.join(ROOT, "tests", "golden", "thetaG11.dat-s")
    opts = dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5, verb=0)
    model = lo.model_from_sdpa(path)
    s = lo.MySolver(model, dict(opts, maxit=after_iterations))
    lo.solve(s)
    X, S, y = _sym_from_f32_lower(s.X[0]), _sym_from_f32_lower(s.S[0]), s.y.copy()
    s.X[0], s.S[0] = X.copy(), S.copy()
    tol_cg = float(s.tol_cg)
    lo.find_mu(s)
    lo.prepare_W(s)
    W, D = s.W[0], s.D[0]
    halpha = lo.Halpha(1)
    lo.Prec_for_CG_tilS_prep(s, halpha)                                   # src/Solvers.jl:674-809
    U = halpha.Umat[0][:, 0]
    U = U * np.sign(U[np.argmax(np.abs(U))])                              # eigenvector sign is free
    Z, cholS = halpha.Z[0], halpha.cholS
    tau2 = float(halpha.AAAATtau.diagonal()[0])
    A = lo.MyA(s.W, model.AA, 0, model.C_lin, None, None)
    M = lo.MyM(model.AA, halpha.AAAATtau, halpha.Umat, halpha.Z, halpha.cholS)
    xv = np.random.default_rng(11).standard_normal(model.n)
    Ax, Mx = np.zeros(model.n), np.zeros(model.n)
    A(Ax, xv)
    M(Mx, xv)
    Rp = model.b - model.AA[0] @ lo.vec(X)
    Rd = model.C[0].toarray() - S - lo.mat(model.AA[0].T @ y)
    h = lo.makeRHS(1, model.AA, s.W, [S], Rp, [Rd])
    cgs = {}
    for tol in (tol_cg, 1e-6, 1e-10):
        xs, ec, it = lo.cg(A, h, tol=tol, maxIter=10000, precon=M)
        cgs[tol] = (xs, ec, it)
    m = W.shape[0]
    V = _probe_vectors(m)
    out = os.path.join(ROOT, "tests", "golden", "iterate_thetaG11.npz")
    np.savez_compressed(out, X_lower_f32=_pack_lower_f32(X), S_lower_f32=_pack_lower_f32(S), y=y, after=after_iterations,
                        D_sorted=np.sort(D), W_diag=np.diag(W).copy(), W_probe=W @ V, probes=V,
                        tau=np.sqrt(tau2), Umat=U, Z_diag=np.diag(Z).copy(), Z_probe=Z @ V,
                        cholS_diag=np.diag(cholS).copy(), cholS_probe=cholS @ V,
                        x=xv, MyA_x=Ax, MyM_x=Mx, Rp=Rp, h=h,
                        cg_tols=np.array(list(cgs)), cg_x=np.stack([cgs[t][0] for t in cgs]),
                        cg_exit=np.array([cgs[t][1] for t in cgs]), cg_iters=np.array([cgs[t][2] for t in cgs]))
    full = lo.MySolver(lo.model_from_sdpa(path), opts)
    lo.solve(full)
    tr = dict(_trace_of(full), status=int(full.status), iterations=int(full.iter), options=opts,
              objective=lo.objective_value(full), dual_objective=lo.dual_objective_value(full),
              cg_total=int(full.cg_iter_tot))
    with open(os.path.join(ROOT, "tests", "golden", "trace_thetaG11.json"), "w") as f:
        json.dump(tr, f)
    print("wrote", out, os.path.getsize(out), "bytes;", full.iter, "iterations,", full.cg_iter_tot, "CG iterations, objective",
          tr["objective"])


def golden_tight(name):
    """kit=1 solves carried to where the answer no longer depends on the path: eDIMACS 1e-8 and
    tol_cg_min 1e-10.  A truncated-CG trajectory is not reproducible between two correct implementations (DESIGN.md
    section 2), the optimum both stop on is: thetaG11 (C3: H_alpha, erank 1) and the truss problem tru3 (H_alpha)."""
    import json
    import time
    path = os.path.join(ROOT, "tests", "golden", name + ".dat-s")
    opts = dict(kit=1, preconditioner=1, erank=1, eDIMACS=float(os.environ.get("TIGHT_EDIMACS", "1e-8")),
                tol_cg_min=float(os.environ.get("TIGHT_TOL_CG_MIN", "1e-10")), verb=int(os.environ.get("TIGHT_VERB", "0")))
    t0 = time.time()
    full = lo.MySolver(lo.model_from_sdpa(path), opts)
    lo.solve(full)
    tr = dict(_trace_of(full), status=int(full.status), iterations=int(full.iter), options=opts,
              objective=lo.objective_value(full), dual_objective=lo.dual_objective_value(full),
              cg_total=int(full.cg_iter_tot), wall_s=time.time() - t0)
    with open(os.path.join(ROOT, "tests", "golden", "trace_%s_tight.json" % name), "w") as f:
        json.dump(tr, f)
    print(name, "tight:", full.iter, "iterations,", full.cg_iter_tot, "CG iterations, status", full.status, "objective %.12f" % tr["objective"],
          "dimacs %.2e" % tr["dimacs"][-1], "%.0f s" % tr["wall_s"])


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["theta1"]):
        if name.endswith("_tight"):
            golden_tight(name[:-6])
            continue
        {"theta1": golden_theta1, "maxG11": golden_maxG11, "thetaG11": golden_thetaG11}[name]()
