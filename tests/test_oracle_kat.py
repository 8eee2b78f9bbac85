"""Oracle pinned against the reference's own known answers.

All of these run the kit=0 general path on CPU.  Known answers are quoted from the
reference's @test lines (file:line next to each)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import loraine_oracle as lo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _solve(model, **opts):
    opts.setdefault("verb", 0)
    s = lo.MySolver(model, opts)
    lo.solve(s)
    return s


def test_theta1_known_answer():
    # examples/solve_sdpa.jl:43-54,61  -> objective_value ~ 23 rtol 1e-6
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"), datarank=0, kappa=8)
    assert model.n == 104 and model.msizes.tolist() == [50]
    assert model.qA[0, 0] == 1          # one dense-path constraint (50 nnz), 103 with 2 nnz
    s = _solve(model, kit=0, tol_cg=1e-2, tol_cg_min=1e-6, eDIMACS=1e-6, preconditioner=1,
               erank=1, aamat=2, datarank=0, initpoint=1, maxit=100, datasparsity=8)
    assert s.status == 1
    assert lo.objective_value(s) == pytest.approx(23.0, rel=1e-6)
    assert lo.dual_objective_value(s) == pytest.approx(23.0, rel=1e-5)


def _sym_units(n):
    """PSD-triangle variable bridge: y_k <-> upper-triangle entry (i<=j), column-major."""
    mats, idx = [], {}
    for j in range(n):
        for i in range(j + 1):
            E = sp.lil_matrix((n, n))
            E[i, j] = 1.0
            E[j, i] = 1.0
            idx[(i, j)] = len(mats)
            idx[(j, i)] = len(mats)
            mats.append(E.tocsc())
    return mats, idx


def _lin_rows(nvar, rows):
    """rows: list of (coef dict {var: c}, const) meaning coef.y + const >= 0.
    Returns C_lin (nvar x nlin) = -coef^T, d_lin = const (MOI_wrapper.jl:145-149,217)."""
    C = sp.lil_matrix((nvar, len(rows)))
    d = np.zeros(len(rows))
    for r, (coef, const) in enumerate(rows):
        for v, c in coef.items():
            C[v, r] = -c
        d[r] = const
    return C.tocsr(), d


def _corr_model(sense_max):
    units, idx = _sym_units(3)
    nvar = 6
    A = [[sp.csc_matrix((3, 3))] + units]
    rows = []
    for i in range(3):                       # rho_ii == 1  -> two inequalities
        rows.append(({idx[(i, i)]: 1.0}, -1.0))
        rows.append(({idx[(i, i)]: -1.0}, 1.0))
    rows.append(({idx[(0, 1)]: 1.0}, 0.2))   # -0.2 <= rho_AB <= -0.1
    rows.append(({idx[(0, 1)]: -1.0}, -0.1))
    rows.append(({idx[(1, 2)]: 1.0}, -0.4))  # 0.4 <= rho_BC <= 0.5
    rows.append(({idx[(1, 2)]: -1.0}, 0.5))
    C_lin, d_lin = _lin_rows(nvar, rows)
    b0 = np.zeros(nvar)
    b0[idx[(0, 2)]] = 1.0
    b = b0 if sense_max else -b0             # MOI_wrapper.jl:206
    return lo.make_model(A, b, 0.0, d_lin, C_lin), idx


def test_ex_corr_known_answers():
    # examples/ex_corr.jl:30-31
    model, idx = _corr_model(True)
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s, max_sense=True) == pytest.approx(0.8719210472, rel=1e-6)
    model, idx = _corr_model(False)
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s, max_sense=False) == pytest.approx(-0.9779977649, rel=1e-6)


def test_ex_dist_known_answer():
    # examples/ex_dist.jl:27-40
    D = np.array([[0, 1, 1, 1], [1, 0, 2, 2], [1, 2, 0, 2], [1, 2, 2, 0]], float)
    units, idx = _sym_units(4)
    nvar = 11                                # y0 = c2, y1.. = Q triangle
    A = [[sp.csc_matrix((4, 4)), sp.csc_matrix((4, 4))] + units]
    q = lambda i, j: 1 + idx[(i, j)]
    rows = [({0: 1.0}, -1.0)]                # c2 >= 1
    for i in range(4):
        for j in range(i + 1, 4):
            e = {q(i, i): 1.0, q(j, j): 1.0, q(i, j): -2.0}
            rows.append((dict(e), -D[i, j] ** 2))                        # D^2 <= expr
            e2 = {k: -v for k, v in e.items()}
            e2[0] = D[i, j] ** 2
            rows.append((e2, 0.0))                                       # expr <= c2 D^2
    rows.append(({q(0, 0): 1.0}, 0.0))       # fix(Q11, 0)
    rows.append(({q(0, 0): -1.0}, 0.0))
    C_lin, d_lin = _lin_rows(nvar, rows)
    b0 = np.zeros(nvar)
    b0[0] = 1.0
    model = lo.make_model(A, -b0, 0.0, d_lin, C_lin)
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s) == pytest.approx(4.0 / 3.0, abs=1e-4)
    Q = np.zeros((4, 4))
    for (i, j), k in idx.items():
        Q[i, j] = s.y[1 + k]
    Qref = np.array([[0, 0, 0, 0], [0, 4, -2, -2], [0, -2, 4, -2], [0, -2, -2, 4]]) / 3.0
    assert np.linalg.norm(Q - Qref) <= 1e-5 * np.linalg.norm(Qref)


def test_ex_maxcut_known_answer():
    # examples/ex_maxcut.jl:43-47 -- cut {1,4} | {2,3}: the graph is bipartite for this
    # cut, so the SDP value equals the total weight 17 and X* = v v' with v = (1,-1,-1,1).
    w = np.array([[0, 1, 5, 0], [1, 0, 0, 9], [5, 0, 0, 2], [0, 9, 2, 0]], float)
    L = np.diag(w.sum(axis=1)) - w
    units, idx = _sym_units(4)
    nvar = 10
    A = [[sp.csc_matrix((4, 4))] + units]
    rows = []
    for i in range(4):
        rows.append(({idx[(i, i)]: 1.0}, -1.0))
        rows.append(({idx[(i, i)]: -1.0}, 1.0))
    C_lin, d_lin = _lin_rows(nvar, rows)
    b0 = np.zeros(nvar)
    for (i, j), k in idx.items():
        if i <= j:
            b0[k] = 0.25 * L[i, j] * (1.0 if i == j else 2.0)
    model = lo.make_model(A, b0, 0.0, d_lin, C_lin)      # Max sense
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s, max_sense=True) == pytest.approx(17.0, rel=1e-5)
    X = np.zeros((4, 4))
    for (i, j), k in idx.items():
        X[i, j] = s.y[k]
    v = np.sign(X[:, 0])
    S = sorted((np.where(v > 0)[0] + 1).tolist())
    T = sorted((np.where(v < 0)[0] + 1).tolist())
    assert (S, T) == ([1, 4], [2, 3])


def test_lp_known_answer():
    # examples/k.jl:8-38 -- pure LP (nlmi = 0): max 2x, 1 <= x <= 2 -> 4, x = 2
    C_lin, d_lin = _lin_rows(1, [({0: 1.0}, -1.0), ({0: -1.0}, 2.0)])
    model = lo.make_model([], np.array([2.0]), 0.0, d_lin, C_lin)
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s, max_sense=True) == pytest.approx(4.0, rel=1e-6)
    assert s.y[0] == pytest.approx(2.0, rel=1e-6)
    # shadow prices: X_lin are the duals of the two rows
    assert s.X_lin[0] == pytest.approx(0.0, abs=1e-6)
    assert s.X_lin[1] == pytest.approx(2.0, rel=1e-6)


def test_control1_external_optimum():
    # SDPLIB README (external, not in the reference): control1 optimum 17.78463
    model = lo.model_from_sdpa(os.path.join(GOLD, "control1.dat-s"))
    assert model.nlmi == 2 and model.nlin == 0
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s) == pytest.approx(17.78463, rel=1e-6)


@pytest.mark.parametrize("name", ["tru3", "vib3"])
def test_truss_primal_dual_agree(name):
    # blocks with negative size -> C_lin rows (nlin > 0) next to LMI blocks
    model = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"))
    assert model.nlin > 0 and model.nlmi >= 1
    s = _solve(model, kit=0)
    assert s.status == 1
    assert lo.objective_value(s) == pytest.approx(lo.dual_objective_value(s), rel=1e-5, abs=1e-6)
