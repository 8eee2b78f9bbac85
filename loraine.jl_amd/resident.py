"""Device-resident variant of the interior-point driver: the matrix variables X, S and every
msz x msz intermediate of predictor / corrector / find_step / check_convergence stay in HBM
(SURVEY.md section 8f ranks 1-3).  The host keeps the nvar- and nlin-vectors (y, Rp, dely,
X_lin, ...) and the scalar logic of `Solvers.jl`; per IP iteration only a handful of
nvar-vectors and scalars cross PCIe.

Same iteration as `solvers.MySolver` (reference src/Solvers.jl:448-568,
src/predictor_corrector.jl), same options, same status codes.
"""
import math

import numpy as np

from . import solvers
from .solvers import MySolver, _dense, _fro


class ResidentSolver(MySolver):
    # ------------------------------------------------------------------ setup
    def setup_solver(self):
        super().setup_solver()
        m = self.model
        if getattr(m, "on_device", False):          # C was built in HBM (synthetic.py)
            self._normC = list(m.normC)
        else:
            for i in range(m.nlmi):
                self.dev.ip_set_c(i, _dense(m.C[i]))
            self._normC = [_fro(m.C[i]) for i in range(m.nlmi)]

    def initial_point(self):
        super().initial_point()
        for i in range(self.model.nlmi):
            self.dev.ip_set_iterate(i, self.X[i], self.S[i])
        self.X = self.S = None                      # the iterate now lives on the device

    def fetch_iterate(self):
        """Bring X, S back (results: constraint duals, dual objective)."""
        out = [self.dev.ip_get_iterate(i) for i in range(self.model.nlmi)]
        self.X = [o[0] for o in out]
        self.S = [o[1] for o in out]

    # ------------------------------------------------------------------ hot path
    def prepare_W(self):
        m = self.model
        for i in range(m.nlmi):
            tries = 0
            while True:
                info = self.dev.ip_prepare_w(i)                             # [GPU]
                if info == 0:
                    break
                self.dev.ip_add_diag(i, info, 1e-5)                          # prepare_W.jl:14
                tries += 1
                if tries > 1000:
                    self.status = 4
                    return
        if m.nlin > 0:
            self.Si_lin = 1.0 / self.S_lin

    def find_mu(self):
        m = self.model
        # <X,S> of the current iterate: already reduced on the device by the previous
        # iteration's check_convergence (X, S have not changed since)
        if getattr(self, "_stats", None) is None:
            self._stats = self.dev.ip_stats() if m.nlmi else np.zeros((0, 5))   # [GPU]
        tr = float(self._stats[:, 0].sum())
        if m.nlin > 0:
            tr += float(self.X_lin @ self.S_lin)
        self.mu = tr / (float(np.sum(m.msizes)) + m.nlin)

    def predictor(self, halpha):
        """src/predictor_corrector.jl:5-146."""
        m = self.model
        dev = self.dev
        self.predict = True
        Rp = m.b.copy()
        if m.nlmi > 0:
            dev.ip_residual_d(self.y)                                        # [GPU] Rd
        if m.nlin > 0:
            Rp -= m.C_lin @ self.X_lin
            self.Rd_lin = m.d_lin - self.S_lin - m.C_lin.T @ self.y
            dev.set_lin(self.X_lin, self.S_lin_inv)
        if self.kit == 0:
            mode = -1 if (self.datarank == -1 and m.nlmi > 0) else 0
            dev.schur_assemble(mode)                                         # [GPU] (+ the exchange when sharded)
        rhs = 0.0
        if m.nlmi > 0:
            # Rp = b - AA*vec(X) (:12) is not used before makeRHS (:44): both products in ONE pass over the constraint
            # data (dense data: 128 GB per pass at C4)
            aax, rhs = dev.ip_rhs_pred2()                                    # [GPU] AA*vec(X), makeRHS
            Rp -= aax
        self.Rp = Rp
        h = self.Rp + rhs
        if m.nlin > 0:
            h = h + m.C_lin @ ((self.X_lin * self.Si_lin) * self.Rd_lin + self.X_lin)
        if self.kit == 0:
            if not self._factor_with_regularisation():
                return
            self.dely = self._schur_solve(h)                                 # [GPU]
        else:
            self.dely, it = self._cg(h, True, halpha)
            self.cg_iter_pre += it
            self.cg_iter_tot += it
        self.find_step()

    def sigma_update(self):
        m = self.model
        step = min(min([*self.alpha, self.alpha_lin]), min([*self.beta, self.beta_lin]))
        if self.mu > 1e-6:
            ex = 1.0 if step < 1.0 / math.sqrt(3.0) else max(self.expon, 3.0 * step * step)
        else:
            ex = max(1.0, min(self.expon, 3.0 * step * step))
        tr = float(np.sum(self._trXnSn)) if m.nlmi else 0.0
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
        if m.nlmi > 0:
            h += self.dev.ip_rhs_corr(self.sigma * self.mu)                  # [GPU] my_kron term
        if m.nlin > 0:
            t = (self.delX_lin * self.delS_lin) * self.Si_lin - (self.sigma * self.mu) * self.Si_lin
            h += m.C_lin @ ((self.X_lin * self.Si_lin) * self.Rd_lin + self.X_lin + t)
        if self.kit == 0:
            self.dely = self._schur_solve(h)
        else:
            self.dely, it = self._cg(h, False, halpha)
            self.cg_iter_cor += it
            self.cg_iter_tot += it
        self.find_step()

    def find_step(self):
        """src/predictor_corrector.jl:248-326 with the matrix work on the device."""
        m = self.model
        if m.nlmi > 0:
            a, b = self.dev.ip_find_step(self.predict, self.sigma * self.mu, self.tau, self.dely)   # [GPU]
            self.alpha[:] = a
            self.beta[:] = b
        if m.nlin > 0:
            self._find_step_lin()
        else:
            self.alpha_lin = self.beta_lin = 1.0
        if self.predict:
            self._trXnSn = self.dev.ip_update(True, self.alpha, self.beta) if m.nlmi else np.zeros(0)
        else:
            a = min([*self.alpha, self.alpha_lin])
            bt = min([*self.beta, self.beta_lin])
            self.y = self.y + bt * self.dely
            if m.nlmi > 0:
                self.dev.ip_update(False, [a], [bt])

    def check_convergence(self):
        """src/Solvers.jl:496-568; the matrix reductions and eigmins come from the device."""
        m = self.model
        st = self.dev.ip_stats() if m.nlmi else np.zeros((0, 5))            # [GPU]
        self._stats = st
        nb = float(np.linalg.norm(m.b))
        by = float(m.b @ self.y)
        e1 = float(np.linalg.norm(self.Rp)) / (1.0 + nb)
        e2 = e3 = e4 = e6 = 0.0
        CX = 0.0
        for i in range(m.nlmi):
            xs, lx, ls, nrd, cx = st[i]
            nC = self._normC[i]
            CX += cx
            e2 += max(0.0, -lx / (1.0 + nb))
            e3 += nrd / (1.0 + nC)
            e4 += max(0.0, -ls / (1.0 + nC))
            e6 += xs / (1.0 + abs(cx) + abs(by))
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
            print("%3d %16.8e %9.2e %8.2f" % (self.iter, self.primal_obj, self.DIMACS_error, self.itertime))
        if self.DIMACS_error < self.eDIMACS:
            self.status = 1
        if self.DIMACS_error > 1e55:
            self.status = 2
        elif abs(by) > 1e55:
            self.status = 3

    def solve(self, halpha=None):
        super().solve(halpha)
        self.fetch_iterate()
        return self


def load(model, options=None, device=None):
    solver = ResidentSolver(model, options, device=device)
    return solver, solvers.Halpha(solver.kit)
