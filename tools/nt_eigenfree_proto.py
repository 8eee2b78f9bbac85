"""NumPy emulation of the eigen-free NT scaling (round 3), run along whole oracle solves.

The reference's prepare_W (src/prepare_W.jl:28-94) takes an SVD of L_S'L_X to get G, Gi, D.  Everything the
iteration consumes can be written without the singular vectors:
  K = CC'CC = L_X' S L_X = V D^2 V' (CC = L_S'L_X),  Y = K^1/2,  Z = K^-1/2   (coupled Newton-Schulz, products only)
  W  = G G'            = L_X Z L_X'
  B = L_X' dS L_X,  T = Z B Z,  TX = L_X^-1 dX L_X^-T = -I - T (+ sigma mu K^-1 + R),  dX = L_X TX L_X'
  eigmin(DDsi.*(Gi dX Gi').*DDsi) = eigmin(TX),  eigmin(DDsi.*(G' dS G).*DDsi) = eigmin(T)   (orthogonal similarity by V)
  G RNT G' = L_X R L_X',  Y R + R Y = -(N Z + Z N'),  N = L_X^-1 dX dS L_X = TX B        (Lyapunov, CG on products)
  G (G'RdG + D - sm/D - RNT) G' = W Rd W + X - sm Si - L_X R L_X'
This script patches the oracle with that route and compares every iteration's objectives with the SVD route.
Usage: python tools/nt_eigenfree_proto.py [theta1 control1 ...]
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import loraine_oracle as O  # noqa: E402

STATS = dict(ns=[], lyap=[], condK=[], ratio1=[])
LYAP_TOL = float(os.environ.get("LYAP_TOL", "1e-12"))
NS_L0 = float(os.environ.get("NS_L0", "1e-3"))


def ns_sqrt(K, l0=None, maxit=40, tol=3e-8):
    """Y = (K/c)^1/2, Z = (K/c)^-1/2 by the scaled coupled Newton-Schulz iteration."""
    n = K.shape[0]
    c = min(np.abs(K).sum(axis=0).max(), np.linalg.norm(K, "fro"))
    Y = K / c
    Z = np.eye(n)
    ell = np.sqrt(NS_L0 if l0 is None else l0)
    its = 0
    for its in range(1, maxit + 1):
        P = Z @ Y
        P = (P + P.T) / 2
        res = np.linalg.norm(np.eye(n) - P, "fro")
        if res < tol:
            a = 1.0                       # converged: the plain (quadratic) step finishes
        elif ell < 1 - 1e-9:
            a = np.sqrt(3.0 / (1.0 + ell + ell * ell))
            ell = 0.5 * a * ell * (3.0 - a * a * ell * ell)
        else:
            a = 1.0
        T = a * (3.0 * np.eye(n) - a * a * P) / 2.0
        Y = Y @ T; Y = (Y + Y.T) / 2
        Z = T @ Z; Z = (Z + Z.T) / 2
        if res < tol:
            break
    return Y * np.sqrt(c), Z / np.sqrt(c), its, c


def lyap_cg(Y, C, tol=1e-13, maxit=400):
    """R with Y R + R Y = C, Y SPD, C symmetric; CG in the Frobenius inner product."""
    n = Y.shape[0]
    R = C / (2.0 * np.trace(Y) / n)
    op = lambda M: (lambda t: t + t.T)(Y @ M)
    r = C - op(R)
    p = r.copy()
    rr = float(np.vdot(r, r))
    nc = np.linalg.norm(C, "fro")
    k = 0
    for k in range(1, maxit + 1):
        if np.sqrt(rr) <= tol * nc:
            break
        Ap = op(p)
        a = rr / float(np.vdot(p, Ap))
        R += a * p
        r -= a * Ap
        rr2 = float(np.vdot(r, r))
        p = r + (rr2 / rr) * p
        rr = rr2
    return R, k


def prepare_W(solver):
    for i in range(solver.model.nlmi):
        LX = O.try_cholesky(solver, solver.X, i, "X")
        LS = O.try_cholesky(solver, solver.S, i, "S")
        m = LX.shape[0]
        CC = LS.T @ LX                      # prepare_W.jl:39; K = CC'CC, never L_X' S L_X (cancellation at cond(X) >= 1e10)
        K = CC.T @ CC
        Y, Z, its, c = ns_sqrt(K)
        ev = np.linalg.eigvalsh(K)
        STATS["ns"].append(its); STATS["condK"].append(ev[-1] / ev[0]); STATS["ratio1"].append(c / ev[-1])
        solver._LX = getattr(solver, "_LX", {}); solver._LS = getattr(solver, "_LS", {})
        solver._Y = getattr(solver, "_Y", {}); solver._Z = getattr(solver, "_Z", {})
        solver._LX[i], solver._LS[i], solver._Y[i], solver._Z[i] = LX, LS, Y, Z
        W = LX @ Z @ LX.T
        solver.W[i] = (W + W.T) / 2
        Linv = sla.solve_triangular(LS, np.eye(m), lower=True)
        solver.Si[i] = Linv.T @ Linv
        solver.G[i] = None; solver.Gi[i] = None; solver.D[i] = None; solver.DDsi[i] = None
    solver.Si_lin = 1.0 / solver.S_lin if solver.model.nlin > 0 else np.zeros(0)


def makeBBBB_rank1(n, nlmi, B, G_unused, solver=None):
    raise RuntimeError("patched in predictor")


def find_step(solver):
    """Everything in the L_X basis, as ipstep.hip does it (no inverse of L_X anywhere)."""
    m = solver.model
    for i in range(m.nlmi):
        LX, Z = solver._LX[i], solver._Z[i]
        solver.delS[i] = solver.Rd[i] - O.mat(m.AA[i].T @ solver.dely)
        B = LX.T @ solver.delS[i] @ LX
        B = (B + B.T) / 2
        T = Z @ B @ Z
        T = (T + T.T) / 2
        TX = -np.eye(T.shape[0]) - T
        if not solver.predict:
            TX = TX + (solver.sigma * solver.mu) * (Z @ Z) + solver._R[i]
        solver._B = getattr(solver, "_B", {}); solver._TX = getattr(solver, "_TX", {})
        solver._B[i], solver._TX[i] = B, TX
        dX = LX @ TX @ LX.T
        solver.delX[i] = (dX + dX.T) / 2
        mimiX = O._eigmin(TX)
        solver.alpha[i] = 0.99 if mimiX > -1e-6 else min(1.0, -solver.tau / mimiX)
        mimiS = O._eigmin(T)
        solver.beta[i] = 0.99 if mimiS > -1e-6 else min(1.0, -solver.tau / mimiS)
    if m.nlin > 0:
        O.find_step_lin(solver)
    else:
        solver.alpha_lin = 1.0
        solver.beta_lin = 1.0
    if solver.predict:
        solver._Q = getattr(solver, "_Q", {}); solver._R = getattr(solver, "_R", {})
        for i in range(m.nlmi):
            LX = solver._LX[i]
            solver.Xn[i] = solver.X[i] + solver.alpha[i] * solver.delX[i]
            solver.Sn[i] = solver.S[i] + solver.beta[i] * solver.delS[i]
            NZ = solver._TX[i] @ solver._B[i] @ solver._Z[i]
            R, k = lyap_cg(solver._Y[i], -(NZ + NZ.T), tol=LYAP_TOL)
            STATS["lyap"].append(k)
            solver._R[i] = R
            solver._Q[i] = LX @ R @ LX.T
    else:
        solver.yold = solver.y
        bmin = min([*solver.beta, solver.beta_lin])
        amin = min([*solver.alpha, solver.alpha_lin])
        solver.y = solver.y + bmin * solver.dely
        for i in range(m.nlmi):
            Xn = solver.X[i] + amin * solver.delX[i]
            solver.X[i] = (Xn + Xn.T) / 2.0
            Sn = solver.S[i] + bmin * solver.delS[i]
            solver.S[i] = (Sn + Sn.T) / 2.0


def corrector(solver, halpha):
    m = solver.model
    solver.predict = False
    h = solver.Rp.copy()
    for i in range(m.nlmi):
        W = solver.W[i]
        inner = W @ solver.Rd[i] @ W + solver.X[i] - (solver.sigma * solver.mu) * solver.Si[i] - solver._Q[i]
        h = h + m.AA[i] @ O.vec(inner)
    if m.nlin > 0:
        tmp = (solver.delX_lin * solver.delS_lin) * solver.Si_lin - (solver.sigma * solver.mu) * solver.Si_lin
        h = h + m.C_lin @ ((solver.X_lin * solver.Si_lin) * solver.Rd_lin + solver.X_lin + tmp)
    if solver.kit == 0:
        L = solver.cholBBBB
        solver.dely = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)
        if getattr(solver, "chol_is_object", False):
            solver.dely = sla.solve_triangular(L.T, sla.solve_triangular(L, solver.dely, lower=True), lower=False)
    else:
        A = O.MyA(solver.W, m.AA, m.nlin, m.C_lin, solver.X_lin, solver.S_lin_inv)
        if solver.preconditioner == 0:
            M = O.MyM_no()
        elif solver.preconditioner == 1:
            M = O.MyM(m.AA, halpha.AAAATtau, halpha.Umat, halpha.Z, halpha.cholS)
        else:
            M = O.MyM_beta(m.AA, halpha.AAAATtau)
        solver.dely, _, it = O.cg(A, h, tol=solver.tol_cg, maxIter=10000, precon=M)
        solver.cg_iter_cor += it
        solver.cg_iter_tot += it
    find_step(solver)


def rank1_from_W(n, nlmi, B, G_list):
    raise NotImplementedError


def run(name, opts, eigenfree):
    ref = "/root/reference/examples/data/%s.dat-s" % name
    path = ref if os.path.exists(ref) else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".dat-s")
    model = O.model_from_sdpa(path, datarank=opts.get("datarank", 0), kappa=opts.get("datasparsity", 8))
    s = O.MySolver(model, dict(opts, verb=0))
    saved = (O.prepare_W, O.find_step, O.corrector, O.makeBBBB_rank1)
    if eigenfree:
        O.prepare_W, O.find_step, O.corrector = prepare_W, find_step, corrector
        s_ref = s

        def r1(n, nlmi, B, G):              # (B W B').^2 without G
            H = np.zeros((n, n))
            for i in range(nlmi):
                t = B[i] @ s_ref.W[i] @ B[i].T
                H += np.asarray(t) ** 2
            return H
        O.makeBBBB_rank1 = r1
    try:
        O.solve(s)
    finally:
        O.prepare_W, O.find_step, O.corrector, O.makeBBBB_rank1 = saved
    return s


if __name__ == "__main__":
    names = sys.argv[1:] or ["theta1", "control1", "tru3", "vib3"]
    base = dict(kit=0, eDIMACS=1e-7)
    for nm in names:
        opts = dict(base)
        if nm == "maxG11":
            opts["datarank"] = -1
        if nm == "thetaG11":
            opts.update(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5)
        for k in STATS:
            STATS[k].clear()
        a = run(nm, opts, False)
        b = run(nm, opts, True)
        worst = 0.0
        for ta, tb in zip(a.trace, b.trace):
            d = abs(ta["primal_obj"] - tb["primal_obj"]) / (1 + abs(ta["primal_obj"]))
            d2 = abs(ta["dual_obj"] - tb["dual_obj"]) / (1 + abs(ta["dual_obj"]))
            worst = max(worst, d, d2)
        print("%-9s iters %d/%d status %d/%d  worst per-iteration objective gap %.2e  final %.10g / %.10g" %
              (nm, len(a.trace), len(b.trace), a.status, b.status, worst, a.trace[-1]["primal_obj"], b.trace[-1]["primal_obj"]))
        print("   NS iterations %s\n   cond(K) %s\n   c/lmax %s\n   Lyapunov CG %s" %
              (STATS["ns"], ["%.0f" % v for v in STATS["condK"]], ["%.1f" % v for v in STATS["ratio1"]], STATS["lyap"]))
