"""Host side of the solver: the interior-point driver of `Loraine.Solvers`
(reference src/Solvers.jl:169-568, src/predictor_corrector.jl, src/initial_point.jl) with its
hot path -- prepare_W, makeBBBBs / makeBBBB_rank1, makeRHS, the Cholesky factor+solves and
the preconditioned-CG solve -- executed on MI355X through the C ABI (`Device`).

The reference's host language (Julia) is not available in this image, so this file plays
the role of `Solvers.jl`: same option names, same status codes, same iteration logic.
Everything marked [GPU] is one call into libloraine_hip.so; there is no CPU fallback for
those calls.  In THIS driver (host arrays, the reference's literal formulas) the step-length
search, the residuals and the DIMACS errors are host NumPy; `resident.ResidentSolver` -- the
default of `optimizer.Optimizer` -- overrides them with the device-resident `lrn_ip_*` calls
(SURVEY.md section 8f rows 1-3), so that X, S and every msz x msz intermediate stay in HBM.
"""
import math
import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from .device import Device
from .model import MyModel

# src/Solvers.jl:169-185
DEFAULT_OPTIONS = {
    "kit": 0, "tol_cg": 1.0e-2, "tol_cg_up": 0.5, "tol_cg_min": 1.0e-7, "eDIMACS": 1.0e-7,
    "preconditioner": 1, "erank": 1, "aamat": 1, "fig_ev": 0, "verb": 1, "datarank": 0,
    "initpoint": 0, "timing": 1, "maxit": 100, "datasparsity": 8,
}


def _vec(M):
    return np.asarray(M).reshape(-1, order="F")


def _mat(v):
    n = math.isqrt(v.size)
    M = np.asarray(v).reshape(n, n, order="F")
    return 0.5 * (M + M.T)


def _eigmin(M):
    # Julia's `eigmin` hands a matrix with NaN / Inf entries to LAPACK and gets NaN back; SciPy raises.  A diverging run
    # (unbounded problem: fuzz seed 3024, objective -1e18) must end in the reference's own exit -- the step lengths turn
    # NaN, cholesky(X) fails and try_cholesky gives up with status 4 (prepare_W.jl:17-21) -- not in a Python exception.
    if not np.isfinite(M).all():
        return float("nan")
    return float(sla.eigvalsh(M, subset_by_index=[0, 0])[0])


def _step(lam, tau):
    """predictor_corrector.jl:274-278 with Julia's NaN semantics (min(1, NaN) is NaN there, 1.0 in Python)."""
    if lam != lam:
        return float("nan")
    return 0.99 if lam > -1e-6 else min(1.0, -tau / lam)


def _fro(M):
    return float(sp.linalg.norm(M)) if sp.issparse(M) else float(np.linalg.norm(M))


def _dense(M):
    """C[i] is sparse for file/array input and a dense array for the synthetic generators."""
    return M.toarray() if sp.issparse(M) else np.asarray(M)


def _cdot(Cm, X):
    return float(Cm.multiply(X).sum()) if sp.issparse(Cm) else float(np.vdot(Cm, X))


class Halpha:
    """src/Solvers.jl:149-162 -- the preconditioner data lives on the device; this object
    only remembers which one was set up in the predictor so the corrector reuses it."""

    def __init__(self, kit):
        self.kit = kit
        self.ready = False


class MySolver:
    """src/Solvers.jl:18-147 + load() :187-302."""

    def __init__(self, model: MyModel, options=None, device: Device = None, device_index=0):
        opt = dict(DEFAULT_OPTIONS)
        for key, val in (options or {}).items():
            if key not in DEFAULT_OPTIONS:
                raise KeyError(f"unsupported option {key!r}")          # MOI_wrapper.jl:86-103
            opt[key] = val
        self.options = opt
        for key in ("kit", "preconditioner", "erank", "aamat", "fig_ev", "verb", "datarank", "initpoint",
                    "timing", "maxit", "datasparsity"):
            setattr(self, key, int(opt[key]))
        for key in ("tol_cg", "tol_cg_up", "tol_cg_min", "eDIMACS"):
            setattr(self, key, float(opt[key]))
        self.model = model
        self._check_ranges()
        self.cg_iter_tot = 0
        self.dist = None                 # sharding.DistributedHotPath when run with one process per GPU
        self.exact_regularised_solve = False   # True: (H + d I)^-1 h instead of the reference's double solve (:85,90)
        self.status = 0
        self.trace = []
        self.dev = device if device is not None else Device(device_index)
        if not getattr(model, "on_device", False):     # synthetic models are generated in HBM
            self.dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes,
                                  B=model.B if len(model.B) else None,
                                  C_lin=model.C_lin if model.nlin else None)    # [GPU] one-time

    def _say(self, msg):
        if self.verb > 0:
            print(msg)

    def _check_ranges(self):
        # src/Solvers.jl:263-291
        if not 0 <= self.kit <= 1:
            self.kit = 0
            self._say(f" ---Parameter kit out of range, setting kit = {self.kit}")
        if self.tol_cg < self.tol_cg_min and self.kit == 1:
            self.tol_cg = self.tol_cg_min
        if self.tol_cg_min > self.eDIMACS and self.kit == 1:
            self.tol_cg_min = self.eDIMACS
        if self.kit == 1 and not 0 <= self.preconditioner <= 4:
            self.preconditioner = 1
        if self.erank < 0:
            self.erank = 1
        if self.datarank < -1:
            self.datarank = 0
        if not 0 <= self.initpoint <= 1:
            self.initpoint = 1

    # ------------------------------------------------------------------ setup / initial point
    def setup_solver(self):
        """src/Solvers.jl:363-446."""
        m = self.model
        zeros = lambda: [np.zeros((int(s), int(s))) for s in m.msizes]
        self.X, self.S, self.delX, self.delS = zeros(), zeros(), zeros(), zeros()
        self.G, self.Gi, self.W, self.Si = zeros(), zeros(), zeros(), zeros()
        self.Rd, self.Xn, self.Sn, self.RNT = zeros(), zeros(), zeros(), zeros()
        self.D = [np.zeros(int(s)) for s in m.msizes]
        self.DDsi = [np.zeros(int(s)) for s in m.msizes]
        self.alpha = np.zeros(m.nlmi)
        self.beta = np.zeros(m.nlmi)
        self.regcount = 0
        self.chol_is_object = False
        if self.kit == 1:
            if m.nlmi == 0:
                self._say("WARNING: Switching to a direct solver, no LMIs")
                self.kit = 0
            elif self.erank >= int(np.max(m.msizes)) - 1:
                self._say("WARNING: Switching to a direct solver, erank bigger than matrix size")
                self.kit = 0
        if len(m.B) > 0 and any(Bi.nnz == 0 for Bi in m.B):
            self.datarank = 0

    def initial_point(self):
        """src/initial_point.jl:1-81."""
        m = self.model
        self.y = np.zeros(m.n)
        b2 = 1.0 + np.abs(m.b)
        f = 0.0
        for i in range(m.nlmi):
            s = float(m.msizes[i])
            if self.initpoint == 0:
                eps_, eta_ = 1.0, float(m.n)
            else:
                f = np.linalg.norm(b2) / (1.0 + _fro(m.AA[i]))
                eps_ = math.sqrt(s) * max(1.0, math.sqrt(s) * f)
                mf = (1.0 + max(f, _fro(m.C[i]))) / math.sqrt(s)
                eta_ = math.sqrt(s) * max(1.0, mf)
            self.X[i] = eps_ * np.eye(int(s))
            self.S[i] = eta_ * np.eye(int(s))
        nl = m.nlin
        if nl > 0:
            rown = np.sqrt(np.asarray(m.C_lin.multiply(m.C_lin).sum(axis=1)).ravel())
            if self.initpoint == 0:
                ex, es = 1.0, 1.0
            else:
                ex = max(1.0, float(np.max(b2 / (1.0 + rown))))
                es = max(1.0, max(float(np.max(rown)), float(np.linalg.norm(m.d_lin))) / math.sqrt(nl))
            self.X_lin = ex * np.ones(nl)
            self.S_lin = es * np.ones(nl)
        else:
            self.X_lin = np.zeros(0)
            self.S_lin = np.zeros(0)
        self.S_lin_inv = 1.0 / self.S_lin
        self.Si_lin = np.zeros(nl)
        for name in ("delX_lin", "delS_lin", "Xn_lin", "Sn_lin", "RNT_lin", "Rd_lin"):
            setattr(self, name, np.zeros(nl))
        self.sigma, self.tau, self.expon = 3.0, 0.95, 3.0
        self.DIMACS_error, self.iter, self.status = 1.0, 0, 0

    # ------------------------------------------------------------------ hot path on the device
    def prepare_W(self):
        """src/prepare_W.jl:28-94 [GPU]; the 1e-5*I retry loop (:5-26) is replayed here from
        the `info` code."""
        m = self.model
        for i in range(m.nlmi):
            tries = 0
            while True:
                info, out = self.dev.prepare_w(i, self.X[i], self.S[i])
                if info == 0:
                    break
                which = self.X if info == 1 else self.S
                which[i] = which[i] + 1e-5 * np.eye(which[i].shape[0])
                tries += 1
                if tries > 1000:
                    self.status = 4
                    return
            self.D[i], self.G[i], self.Gi[i] = out["D"], out["G"], out["Gi"]
            self.W[i], self.Si[i], self.DDsi[i] = out["W"], out["Si"], out["DDsi"]
        if m.nlin > 0:
            self.Si_lin = 1.0 / self.S_lin

    def _factor_with_regularisation(self):
        """cholesky(BBBB) + the 1e-4*I loop of src/predictor_corrector.jl:55-85 [GPU]."""
        self.chol_is_object = False
        info = self.dev.schur_factor()
        if info == 0:
            nb = self.dev.count("chol_boosted")
            if nb:                                   # not the reference's behaviour: say so
                self._say(f"H numerically singular: {nb} pivot(s) at rounding level boosted")
            return True
        self._say("Matrix H not positive definite, trying to regularize")
        self.regcount += 1
        if self.regcount > 5:
            self.status = 3
            return False
        for k in range(1001):
            self.dev.schur_add_diag(1e-4)
            if self.dev.schur_factor() == 0:
                self.reg_adds = k + 1
                # the reference stores the Cholesky OBJECT in this branch (:85) and the factor L otherwise (:57-58), so
                # its `cholBBBB' \ (cholBBBB \ h)` (:90, :199) solves twice for the rest of this IP iteration
                self.chol_is_object = not self.exact_regularised_solve
                return True
        self.status = 3
        return False

    def _schur_solve(self, h):
        r"""`cholBBBB' \ (cholBBBB \ h)` (src/predictor_corrector.jl:90,199) [GPU]; in a regularised iteration the
        reference's operand is a Cholesky object and the expression is H_reg^-1 (H_reg^-1 h) -- reproduced unless the
        solver was created with exact_regularised_solve=True (divergences)."""
        x = self.dev.schur_solve(h)
        if self.chol_is_object:
            x = self.dev.schur_solve(x)
        return x

    def _cg(self, h, setup, halpha):
        if setup:
            if self.preconditioner == 1:
                info = self.dev.prec_setup(1, self.erank, self.aamat)          # Prec_for_CG_tilS_prep
            elif self.preconditioner in (2, 4):
                info = self.dev.prec_setup(2, self.erank, self.aamat)          # Prec_for_CG_beta
            else:
                info = self.dev.prec_setup(0, self.erank, self.aamat)
            if info != 0:
                raise np.linalg.LinAlgError("PosDefException in preconditioner setup")
            halpha.ready = True
        if self.dist is not None:                                              # multi-GPU: all-reduce PCG
            x, _exit_code, iters = self.dist.pcg(self.dev, h, self.tol_cg, 10000)
        else:
            x, _exit_code, iters = self.dev.pcg(h, self.tol_cg, 10000)        # exit code ignored (:134)
        return x, iters

    # ------------------------------------------------------------------ predictor / corrector
    def predictor(self, halpha):
        """src/predictor_corrector.jl:5-146."""
        m = self.model
        self.predict = True
        Rp = m.b.copy()
        for i in range(m.nlmi):
            Rp -= m.AA[i] @ _vec(self.X[i])
            self.Rd[i] = _dense(m.C[i]) - self.S[i] - _mat(m.AA[i].T @ self.y)
        if m.nlin > 0:
            Rp -= m.C_lin @ self.X_lin
            self.Rd_lin = m.d_lin - self.S_lin - m.C_lin.T @ self.y
            self.dev.set_lin(self.X_lin, self.S_lin_inv)
        self.Rp = Rp
        dev = self.dev
        if self.kit == 0:
            mode = -1 if (self.datarank == -1 and m.nlmi > 0) else 0
            dev.schur_assemble(mode)                                           # [GPU] makeBBBB*
            if self.dist is not None:
                self.dist.allgather(dev)                                       # multi-GPU: column blocks -> all ranks
        if m.nlmi > 0:
            h = dev.make_rhs(self.Rp, [self.Rd[i] + self.S[i] for i in range(m.nlmi)])   # [GPU] makeRHS
        else:
            h = self.Rp.copy()
        if m.nlin > 0:
            h = h + m.C_lin @ ((self.X_lin * self.Si_lin) * self.Rd_lin + self.X_lin)
        if self.kit == 0:
            if not self._factor_with_regularisation():
                return
            self.dely = self._schur_solve(h)                                   # [GPU] L'\(L\h)
        else:
            self.dely, it = self._cg(h, True, halpha)
            self.cg_iter_pre += it
            self.cg_iter_tot += it
        self.find_step()

    def sigma_update(self):
        """src/predictor_corrector.jl:148-179."""
        m = self.model
        step = min(min([*self.alpha, self.alpha_lin]), min([*self.beta, self.beta_lin]))
        if self.mu > 1e-6:
            ex = 1.0 if step < 1.0 / math.sqrt(3.0) else max(self.expon, 3.0 * step * step)
        else:
            ex = max(1.0, min(self.expon, 3.0 * step * step))
        tr = sum(float(np.sum(self.Xn[i] * self.Sn[i])) for i in range(m.nlmi))
        if tr < 0:
            self.sigma = 0.8
            return
        lin = float(self.Xn_lin @ self.Sn_lin) if m.nlin > 0 else 0.0
        ratio = (tr + lin) / (float(np.sum(m.msizes)) + m.nlin) / self.mu
        self.sigma = min(1.0, ratio ** ex)

    def corrector(self, halpha):
        """src/predictor_corrector.jl:181-246."""
        m = self.model
        self.predict = False
        h = self.Rp.copy()
        for i in range(m.nlmi):
            G = self.G[i]
            core = G.T @ self.Rd[i] @ G + np.diag(self.D[i] - (self.sigma * self.mu) / self.D[i]) - self.RNT[i]
            h += m.AA[i] @ _vec(G @ core @ G.T)                                # my_kron(G,G,.)  :186
        if m.nlin > 0:
            t = (self.delX_lin * self.delS_lin) * self.Si_lin - (self.sigma * self.mu) * self.Si_lin
            h += m.C_lin @ ((self.X_lin * self.Si_lin) * self.Rd_lin + self.X_lin + t)
        if self.kit == 0:
            self.dely = self._schur_solve(h)                                   # [GPU] :199
        else:
            self.dely, it = self._cg(h, False, halpha)
            self.cg_iter_cor += it
            self.cg_iter_tot += it
        self.find_step()

    def find_step(self):
        """src/predictor_corrector.jl:248-326 (host)."""
        m = self.model
        for i in range(m.nlmi):
            W, G, Gi, dd = self.W[i], self.G[i], self.Gi[i], self.DDsi[i]
            self.delS[i] = self.Rd[i] - _mat(m.AA[i].T @ self.dely)
            WdSW = W @ self.delS[i] @ W
            if self.predict:
                self.delX[i] = _mat(_vec(-self.X[i] - WdSW))
            else:
                self.delX[i] = _mat(_vec((self.sigma * self.mu) * self.Si[i] - self.X[i] - WdSW + G @ self.RNT[i] @ G.T))
            dSb = G.T @ self.delS[i] @ G
            dXb = Gi @ self.delX[i] @ Gi.T
            for name, Mb in (("alpha", dXb), ("beta", dSb)):
                Q = dd[None, :] * Mb * dd[:, None]
                lam = _eigmin(0.5 * (Q + Q.T))
                getattr(self, name)[i] = _step(lam, self.tau)
        if m.nlin > 0:
            self._find_step_lin()
        else:
            self.alpha_lin = self.beta_lin = 1.0
        if self.predict:
            for i in range(m.nlmi):
                G, Gi = self.G[i], self.Gi[i]
                self.Xn[i] = self.X[i] + self.alpha[i] * self.delX[i]
                self.Sn[i] = self.S[i] + self.beta[i] * self.delS[i]
                dsum = self.D[i][:, None] + self.D[i][None, :]
                self.RNT[i] = -(Gi @ self.delX[i] @ self.delS[i] @ G + G.T @ self.delS[i] @ self.delX[i] @ Gi.T) / dsum
        else:
            a = min([*self.alpha, self.alpha_lin])
            bt = min([*self.beta, self.beta_lin])
            self.y = self.y + bt * self.dely
            for i in range(m.nlmi):
                Xi = self.X[i] + a * self.delX[i]
                Si = self.S[i] + bt * self.delS[i]
                self.X[i] = 0.5 * (Xi + Xi.T)
                self.S[i] = 0.5 * (Si + Si.T)

    def _find_step_lin(self):
        """src/predictor_corrector.jl:329-364."""
        m = self.model
        self.delS_lin = self.Rd_lin - m.C_lin.T @ self.dely
        self.delX_lin = -self.X_lin - self.X_lin * self.Si_lin * self.delS_lin
        if not self.predict:
            self.delX_lin = self.delX_lin + (self.sigma * self.mu) * self.Si_lin + self.RNT_lin
        lx = float(np.min(self.delX_lin / self.X_lin))
        ls = float(np.min(self.delS_lin / self.S_lin))
        self.alpha_lin = 0.99 if lx > -1e-6 else min(1.0, -self.tau / lx)
        self.beta_lin = 0.99 if ls > -1e-6 else min(1.0, -self.tau / ls)
        if self.predict:
            self.Xn_lin = self.X_lin + self.alpha_lin * self.delX_lin
            self.Sn_lin = self.S_lin + self.beta_lin * self.delS_lin
            self.RNT_lin = -(self.delX_lin * self.delS_lin) * self.Si_lin
        else:
            a = min([*self.alpha, self.alpha_lin])
            bt = min([*self.beta, self.beta_lin])
            self.X_lin = self.X_lin + a * self.delX_lin
            self.S_lin = self.S_lin + bt * self.delS_lin
            self.S_lin_inv = 1.0 / self.S_lin

    # ------------------------------------------------------------------ IP step, convergence
    def find_mu(self):
        m = self.model
        tr = sum(float(np.sum(self.X[i] * self.S[i])) for i in range(m.nlmi))
        if m.nlin > 0:
            tr += float(self.X_lin @ self.S_lin)
        self.mu = tr / (float(np.sum(m.msizes)) + m.nlin)

    def myIPstep(self, halpha):
        """src/Solvers.jl:448-478."""
        self.iter += 1
        if self.iter > self.maxit:
            self.status = 4
            self._say("WARNING: Stopped by iteration limit (stopping status = 4)")
        self.cg_iter_pre = self.cg_iter_cor = 0
        self.find_mu()
        self.dev.reset_timing()
        self.prepare_W()
        if self.status == 4 and self.iter <= self.maxit:
            return
        self.predictor(halpha)
        if self.status in (2, 3):
            return
        self.sigma_update()
        self.corrector(halpha)

    def check_convergence(self):
        """src/Solvers.jl:496-568 (norm(M,2) of a matrix is Frobenius in Julia)."""
        m = self.model
        nb = float(np.linalg.norm(m.b))
        by = float(m.b @ self.y)
        e1 = float(np.linalg.norm(self.Rp)) / (1.0 + nb)
        e2 = e3 = e4 = e6 = 0.0
        CX = 0.0
        for i in range(m.nlmi):
            nC = _fro(m.C[i])
            cx = _cdot(m.C[i], self.X[i])
            CX += cx
            e2 += max(0.0, -_eigmin(self.X[i]) / (1.0 + nb))
            e3 += float(np.linalg.norm(self.Rd[i])) / (1.0 + nC)
            e4 += max(0.0, -_eigmin(self.S[i]) / (1.0 + nC))
            e6 += float(np.sum(self.S[i] * self.X[i])) / (1.0 + abs(cx) + abs(by))
        e5 = (CX - by) / (1.0 + abs(CX) + abs(by))
        dX = 0.0
        if m.nlin > 0:
            nd = float(np.linalg.norm(m.d_lin))
            dX = float(m.d_lin @ self.X_lin)
            e2 += max(0.0, -float(np.min(self.X_lin)) / (1.0 + nb))
            e3 += float(np.linalg.norm(self.Rd_lin)) / (1.0 + nd)
            e4 += max(0.0, -float(np.min(self.S_lin)) / (1.0 + nd))
            e5 = (CX + dX - by) / (1.0 + abs(CX) + abs(by))
            e6 += float(self.S_lin @ self.X_lin) / (1.0 + abs(dX) + abs(by))
        self.err1, self.err2, self.err3, self.err4, self.err5, self.err6 = e1, e2, e3, e4, e5, e6
        self.DIMACS_error = (e1 if m.nlmi > 0 else 0.0) + e2 + e3 + e4 + abs(e5) + e6
        self.primal_obj = -by + m.b_const
        self.dual_obj = -CX - dX
        if self.verb > 0 and self.status == 0:
            if self.kit == 0:
                print("%3d %16.8e %9.2e %8.2f" % (self.iter, self.primal_obj, self.DIMACS_error, self.itertime), flush=True)
            else:
                print("%3d %16.8e %9.2e %9d %8.2f" % (self.iter, self.primal_obj, self.DIMACS_error,
                                                      self.cg_iter_pre + self.cg_iter_cor, self.itertime), flush=True)
        if self.DIMACS_error < self.eDIMACS:
            self.status = 1
            self._say(f"Primal objective: {self.primal_obj}")
            self._say(f"Dual objective:   {self.dual_obj}")
        if self.DIMACS_error > 1e55:
            self.status = 2
        elif abs(by) > 1e55:
            self.status = 3

    def solve(self, halpha=None):
        """src/Solvers.jl:304-361."""
        halpha = halpha or Halpha(self.kit)
        t1 = time.perf_counter()
        if self.verb > 0:
            print(" *** IP STARTS")
            print(" it        obj         error     CPU/it" if self.kit == 0
                  else " it        obj         error     cg_iter   CPU/it")
        self.setup_solver()
        self.initial_point()
        self.dev.set_option("profile", 1)
        while self.status == 0:
            t2 = time.perf_counter()
            self.myIPstep(halpha)
            self.itertime = time.perf_counter() - t2
            self.tol_cg = max(self.tol_cg * self.tol_cg_up, self.tol_cg_min)
            if self.status in (2, 3):
                break
            self.check_convergence()
            d = self.dev
            self.trace.append(dict(
                iter=self.iter, primal_obj=self.primal_obj, dual_obj=self.dual_obj, dimacs=self.DIMACS_error,
                errs=(self.err1, self.err2, self.err3, self.err4, self.err5, self.err6), mu=self.mu,
                sigma=self.sigma, cg_pre=self.cg_iter_pre, cg_cor=self.cg_iter_cor, itertime=self.itertime,
                gpu_ms=dict(prepare_w=d.timing("prepare_w"), assemble=d.timing("assemble"),
                            factor=d.timing("factor"), solve=d.timing("solve"),
                            prec_setup=d.timing("prec_setup"), pcg=d.timing("pcg"), svd=d.timing("prepw_svd")),
                svd_sweeps=d.count("svd_sweeps"), find_step_ms=d.timing("find_step"),
                rhs_ms=d.timing("rhs"), residual_d_ms=d.timing("residual_d"), stats_ms=d.timing("stats"),
                hop_assemble=d.count("hop_assemble"), hop_matvec=d.count("hop_matvec"),
                prec_lanczos_steps=d.count("prec_lanczos_steps"), lanczos_plain=d.count("lanczos_plain"),
                prec_dense_build=d.count("prec_dense_build"),
                ns_steps=d.count("ns_steps"), lyap_steps=d.count("lyap_steps"), lyap_ms=d.timing("lyap"),
                ns_fallback=d.count("ns_fallback"), lyap_fallback=d.count("lyap_fallback"),
                lanczos_steps=d.count("lanczos_steps"), lanczos_runs=d.count("lanczos_runs"),
                stats_chol=d.count("stats_chol"), eigmin_chol_tests=d.count("eigmin_chol_tests"),
                schur_chol=d.count("schur_chol"), schur_via_l=d.count("schur_via_l"), wchol_fail=d.count("wchol_fail"),
                alpha=[float(a) for a in self.alpha] + [float(self.alpha_lin)],
                beta=[float(b) for b in self.beta] + [float(self.beta_lin)], regcount=self.regcount,
                reg_adds=getattr(self, "reg_adds", 0), chol_boosted=d.count("chol_boosted")))
            if time.perf_counter() - t1 > getattr(self, "time_budget", float("inf")):
                self.status = 4            # tools/c5_solve.py: wall-clock cap for exploratory runs
            if self.preconditioner == 4:
                n_ = self.model.n
                if ((self.cg_iter_cor / 2 > self.erank * self.model.nlmi * math.sqrt(n_) / 20
                     and self.iter > math.sqrt(n_) / 60) or self.cg_iter_cor > 100):
                    self.preconditioner, self.aamat = 1, 2
                    self._say("Switching to preconditioner 1")
        self.tottime = time.perf_counter() - t1
        if self.verb > 0:
            if self.kit == 1:
                print(" *** Total CG iterations: %8d " % self.cg_iter_tot)
            if self.status == 1:
                print(" *** Optimal solution found in %8.2f seconds" % self.tottime)
        return self


def load(model, options=None, device=None):
    """src/Solvers.jl:187-302 -> (solver, halpha)."""
    solver = MySolver(model, options, device=device)
    if solver.verb > 0:
        print("\n *** Loraine.jl v0.2.5 hot path on MI355X (loraine.jl_amd) ***")
        print(" Number of variables: %5d" % model.n)
        print(" LMI constraints    : %5d" % model.nlmi)
        if model.nlmi > 0:
            print(" Matrix size(s)     :" + "".join("%6d" % s for s in model.msizes))
        print(" Linear constraints : %5d" % model.nlin)
        print((" Preconditioner     : %5d" % solver.preconditioner) if solver.kit > 0
              else " Preconditioner     :  none, using direct solver")
    return solver, Halpha(solver.kit)


def solve(solver, halpha=None):
    return solver.solve(halpha)
