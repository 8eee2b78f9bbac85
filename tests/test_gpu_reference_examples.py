"""The reference's own example problems (examples/ex_corr.jl, ex_dist.jl,
ex_maxcut.jl, k.jl) through `Optimizer.load_model` on the GPU path, both drivers.  Known answers are
the reference's @test lines (file:line next to each); the problem builders are shared with
tests/test_oracle_kat.py, which pins the CPU oracle on the same data."""
import numpy as np
import pytest
import scipy.sparse as sp

import test_oracle_kat as kat

pytestmark = pytest.mark.gpu


def _solve(A, b, d_lin, C_lin, max_sense, resident, **opts):
    from loraine_jl_amd.optimizer import Optimizer
    o = Optimizer(resident=resident)
    o.set_silent(True)
    o.set_attribute("kit", 0)
    for k, v in opts.items():
        o.set_attribute(k, v)
    o.load_model(A, b, 0.0, d_lin, C_lin, max_sense=max_sense)
    o.optimize()
    assert o.termination_status() == "OPTIMAL"
    return o


def _corr(sense_max):
    units, idx = kat._sym_units(3)
    rows = []
    for i in range(3):
        rows.append(({idx[(i, i)]: 1.0}, -1.0))
        rows.append(({idx[(i, i)]: -1.0}, 1.0))
    rows += [({idx[(0, 1)]: 1.0}, 0.2), ({idx[(0, 1)]: -1.0}, -0.1), ({idx[(1, 2)]: 1.0}, -0.4), ({idx[(1, 2)]: -1.0}, 0.5)]
    C_lin, d_lin = kat._lin_rows(6, rows)
    b0 = np.zeros(6)
    b0[idx[(0, 2)]] = 1.0
    return [[sp.csc_matrix((3, 3))] + units], (b0 if sense_max else -b0), d_lin, C_lin


@pytest.mark.parametrize("resident", [True, False])
def test_ex_corr(resident):
    # examples/ex_corr.jl:30-31
    A, b, d, C = _corr(True)
    assert _solve(A, b, d, C, True, resident).objective_value() == pytest.approx(0.8719210472, rel=1e-6)
    A, b, d, C = _corr(False)
    assert _solve(A, b, d, C, False, resident).objective_value() == pytest.approx(-0.9779977649, rel=1e-6)


@pytest.mark.parametrize("resident", [True, False])
def test_ex_dist(resident):
    # examples/ex_dist.jl:27-40
    D = np.array([[0, 1, 1, 1], [1, 0, 2, 2], [1, 2, 0, 2], [1, 2, 2, 0]], float)
    units, idx = kat._sym_units(4)
    q = lambda i, j: 1 + idx[(i, j)]
    rows = [({0: 1.0}, -1.0)]
    for i in range(4):
        for j in range(i + 1, 4):
            e = {q(i, i): 1.0, q(j, j): 1.0, q(i, j): -2.0}
            rows.append((dict(e), -D[i, j] ** 2))
            e2 = {k: -v for k, v in e.items()}
            e2[0] = D[i, j] ** 2
            rows.append((e2, 0.0))
    rows += [({q(0, 0): 1.0}, 0.0), ({q(0, 0): -1.0}, 0.0)]
    C_lin, d_lin = kat._lin_rows(11, rows)
    b0 = np.zeros(11)
    b0[0] = 1.0
    A = [[sp.csc_matrix((4, 4)), sp.csc_matrix((4, 4))] + units]      # y0 = c2 has an empty LMI matrix (nnz = 0)
    o = _solve(A, -b0, d_lin, C_lin, False, resident)
    assert o.objective_value() == pytest.approx(4.0 / 3.0, abs=1e-4)
    y = o.variable_primal()
    Q = np.zeros((4, 4))
    for (i, j), k in idx.items():
        Q[i, j] = y[1 + k]
    Qref = np.array([[0, 0, 0, 0], [0, 4, -2, -2], [0, -2, 4, -2], [0, -2, -2, 4]]) / 3.0
    assert np.linalg.norm(Q - Qref) <= 1e-5 * np.linalg.norm(Qref)


@pytest.mark.parametrize("resident", [True, False])
def test_ex_maxcut(resident):
    # examples/ex_maxcut.jl:43-47: cut {1,4} | {2,3}, value 17
    w = np.array([[0, 1, 5, 0], [1, 0, 0, 9], [5, 0, 0, 2], [0, 9, 2, 0]], float)
    L = np.diag(w.sum(axis=1)) - w
    units, idx = kat._sym_units(4)
    rows = []
    for i in range(4):
        rows.append(({idx[(i, i)]: 1.0}, -1.0))
        rows.append(({idx[(i, i)]: -1.0}, 1.0))
    C_lin, d_lin = kat._lin_rows(10, rows)
    b0 = np.zeros(10)
    for (i, j), k in idx.items():
        if i <= j:
            b0[k] = 0.25 * L[i, j] * (1.0 if i == j else 2.0)
    o = _solve([[sp.csc_matrix((4, 4))] + units], b0, d_lin, C_lin, True, resident)
    assert o.objective_value() == pytest.approx(17.0, rel=1e-5)
    y = o.variable_primal()
    X = np.zeros((4, 4))
    for (i, j), k in idx.items():
        X[i, j] = y[k]
    v = np.sign(X[:, 0])
    assert sorted((np.where(v > 0)[0] + 1).tolist()) == [1, 4]


@pytest.mark.parametrize("resident", [True, False])
def test_pure_lp(resident):
    # examples/k.jl:8-38 -- nlmi = 0: max 2x, 1 <= x <= 2 -> 4 at x = 2, shadow prices (0, 2)
    C_lin, d_lin = kat._lin_rows(1, [({0: 1.0}, -1.0), ({0: -1.0}, 2.0)])
    o = _solve([], np.array([2.0]), d_lin, C_lin, True, resident)
    assert o.objective_value() == pytest.approx(4.0, rel=1e-6)
    assert o.variable_primal()[0] == pytest.approx(2.0, rel=1e-6)
    lam = o.constraint_dual_lin()
    assert lam[0] == pytest.approx(0.0, abs=1e-6) and lam[1] == pytest.approx(2.0, rel=1e-6)


@pytest.mark.parametrize("resident", [True, False])
def test_tiny_blocks(resident):
    """Edge sizes: a 1x1 and a 2x2 LMI block, an empty constraint matrix in one block.
        min y0 + y1   s.t.  [y0 - 1] >= 0,   [[y1, 1], [1, y1]] >= 0     ->  y = (1, 1), value 2"""
    A1 = [sp.csc_matrix([[1.0]]), sp.csc_matrix([[1.0]]), sp.csc_matrix((1, 1))]          # F0 = 1, F1 = 1, F2 = 0
    A2 = [sp.csc_matrix(-np.array([[0.0, 1.0], [1.0, 0.0]])), sp.csc_matrix((2, 2)), sp.identity(2, format="csc")]
    o = _solve([A1, A2], -np.ones(2), None, None, False, resident)
    assert o.objective_value() == pytest.approx(2.0, rel=1e-6)
    assert o.variable_primal() == pytest.approx([1.0, 1.0], rel=1e-5)
