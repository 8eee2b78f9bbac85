"""Builder-defined synthetic dense SDP: the constraint
matrices A_k = (R_k + R_k')/2 are generated in HBM by a counter-based Philox stream
(lrn_synthetic_dense_model, 128 GB at msz=2000 / nvar=4000 -- more than the host holds), and
completed to a strictly feasible problem on the device (lrn_synthetic_dense_problem):
X0 = I + QQ'/msz, b = AA vec(X0), y0 ~ N(0,1)/sqrt(nvar), S0 = I, C = S0 + mat(AA' y0)."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from ._capi import ptr
from .resident import ResidentSolver
from .solvers import Halpha


class SyntheticDenseModel:
    """Duck-types `MyModel` for the device-resident driver; the matrix data never exists on
    the host."""
    on_device = True

    def __init__(self, msz, nvar, b, normC, y0):
        self.n = int(nvar)
        self.msizes = np.array([int(msz)], dtype=np.int64)
        self.nlmi, self.nlin = 1, 0
        self.b = np.asarray(b, float)
        self.b_const = 0.0
        self.normC = [float(normC)]
        self.y0 = np.asarray(y0, float)
        self.B = []
        self.C = None
        self.AA = None
        self.d_lin = np.zeros(0)
        self.C_lin = sp.csr_matrix((self.n, 0))


def synthetic_dense_solver(dev, msz, nvar, seed=20250614, options=None):
    """-> (ResidentSolver, Halpha) for the synthetic dense SDP, data generated on `dev`."""
    dev.synthetic_dense_model(msz, nvar, seed)
    b = np.zeros(nvar); y0 = np.zeros(nvar); nc = C.c_double(0.0)
    dev._chk(dev.lib.lrn_synthetic_dense_problem(dev.h, C.c_uint64(seed + 1), ptr(b), ptr(y0), C.byref(nc)),
             "lrn_synthetic_dense_problem")
    model = SyntheticDenseModel(msz, nvar, b, nc.value, y0)
    opts = dict(options or {})
    if int(opts.get("initpoint", 0)) != 0:
        raise ValueError("the synthetic model supports initpoint = 0 only (AA is not on the host)")
    solver = ResidentSolver(model, opts, device=dev)
    return solver, Halpha(solver.kit)


# ------------------------------------------------------------------------------------------------
# C5: builder-defined synthetic SDP with sparse constraints and a planted low-rank optimum
# (SURVEY.md section 8d).  Host-side NumPy: the data is O(nvar) apart from the dense C.
class LowRankProblem:
    """max b'y  s.t.  S = C - sum_k y_k M_k >= 0   (dual: <M_k, X> = b_k, X >= 0), with

      * M_k: symmetric, traceless, supported on a random 3 x 3 principal block (9 nnz, like the
        3 x 3 blocks of thetaG11); AA[k, :] = vec(M_k), i.e. F_k = -M_k in SDPA terms;
      * planted pair  X* = Q diag(lam) Q'  (rank r, trace sqrt(msz): with trace msz the iteration stalls
        at <X,S> ~ 2e-5 in FP64 -- the CPU restatement does the same -- before DIMACS 1e-5 is met),  S* = I - Q Q',  y* ~ 0.1 N(0,1),
        b_k = <M_k, X*>,  C = S* + sum_k y*_k M_k.   X* S* = 0 and rank X* + rank S* = msz, so
        (X*, y*, S*) is optimal and strictly complementary:  optimum  b'y* = <C, X*>.
    """

    def __init__(self, msz, nvar, rank=4, seed=20250615, xtrace=None):
        rng = np.random.default_rng(seed)
        self.msz, self.nvar, self.rank, self.seed = int(msz), int(nvar), int(rank), int(seed)
        i0 = rng.integers(0, msz, nvar)
        d1 = rng.integers(1, msz, nvar)
        d2 = rng.integers(1, msz - 1, nvar)
        d2 = d2 + (d2 >= d1)
        self.idx = np.stack([i0, (i0 + d1) % msz, (i0 + d2) % msz], axis=1)       # nvar x 3, distinct
        R = rng.standard_normal((nvar, 3, 3))
        blk = 0.5 * (R + R.transpose(0, 2, 1))
        tr = np.trace(blk, axis1=1, axis2=2) / 3.0
        blk[:, [0, 1, 2], [0, 1, 2]] -= tr[:, None]
        self.blocks = blk                                                          # nvar x 3 x 3
        Q, _ = np.linalg.qr(rng.standard_normal((msz, rank)))
        lam = 1.0 + np.arange(rank) / rank
        self.Q, self.lam = Q, lam * ((np.sqrt(msz) if xtrace is None else xtrace) / lam.sum())
        Qk = Q[self.idx]                                                           # nvar x 3 x r
        Xk = np.einsum("kia,a,kja->kij", Qk, self.lam, Qk)
        self.b = np.einsum("kij,kij->k", blk, Xk)
        self.ystar = 0.1 * rng.standard_normal(nvar)
        self.optimum = float(self.b @ self.ystar)

    def AA(self):
        m, n = self.msz, self.nvar
        r = np.repeat(self.idx, 3, axis=1).reshape(n, 3, 3)          # row index of entry (a, b): idx[a]
        c = np.tile(self.idx, (1, 3)).reshape(n, 3, 3)               # col index: idx[b]
        cols = (c.astype(np.int64) * m + r).reshape(n, 9)
        order = np.argsort(cols, axis=1)
        cols = np.take_along_axis(cols, order, axis=1)
        vals = np.take_along_axis(self.blocks.reshape(n, 9), order, axis=1)
        indptr = np.arange(0, 9 * n + 1, 9, dtype=np.int64)
        return sp.csr_matrix((vals.ravel(), cols.ravel(), indptr), shape=(n, m * m))

    def C_dense(self):
        m = self.msz
        Cd = np.eye(m) - self.Q @ self.Q.T
        r = np.repeat(self.idx, 3, axis=1).ravel()
        c = np.tile(self.idx, (1, 3)).ravel()
        np.add.at(Cd, (r, c), (self.ystar[:, None, None] * self.blocks).ravel())
        return np.asfortranarray(Cd)

    def constraint(self, k):
        """M_k as a sparse msz x msz matrix (tests / the CPU oracle)."""
        ii = self.idx[k]
        return sp.csc_matrix((self.blocks[k].ravel(), (np.repeat(ii, 3), np.tile(ii, 3))), shape=(self.msz, self.msz))

    def model(self, kappa=8):
        from .model import MyModel
        n = self.nvar
        nzA = np.full((n, 1), 9, dtype=np.int64)
        sigmaA = np.arange(n, dtype=np.int64).reshape(n, 1)
        qA = np.full((2, 1), n if 9 > kappa else 0, dtype=np.int64)
        return MyModel(None, [self.AA()], [], [self.C_dense()], nzA, sigmaA, qA, self.b.copy(), 0.0, np.zeros(0),
                       sp.csr_matrix((n, 0)), n, np.array([self.msz], dtype=np.int64), 0, 1)
