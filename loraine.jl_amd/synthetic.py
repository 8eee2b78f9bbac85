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
