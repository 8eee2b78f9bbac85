"""Host-side eigenvalues of the Lanczos tridiagonal matrices (csrc/tridiag.h) against LAPACK through SciPy.

After every batch of Lanczos steps the drivers of the step-length rule (reference: eigmin(XXX),
src/predictor_corrector.jl:272,285) and of the H_alpha setup (eigen(W), src/Solvers.jl:642,706) take the extreme
eigenvalues of T_m on the host: bisection on a division-free Sturm count, bracketed from the previous batch's value.
No GPU needed: lrn_dbg_tridiag_eig has no context argument."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
from scipy.linalg import eigvalsh_tridiagonal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def lib():
    from loraine_jl_amd import _capi
    return _capi.load_library()


def kth(lib, a, b, k, upper=None, width=0.0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = C.c_double(0.0)
    ev = C.c_int64(0)
    up = C.c_double(upper) if upper is not None else None
    rc = lib.lrn_dbg_tridiag_eig(len(a), a.ctypes.data, b.ctypes.data if len(b) else None, int(k),
                                 C.addressof(up) if up is not None else None, float(width), C.byref(out), C.byref(ev))
    assert rc == 0
    return out.value, ev.value


def lanczos_T(lam, m, seed):
    """plain Lanczos (no re-orthogonalisation: ghost copies of converged Ritz values appear) on diag(lam)"""
    rng = np.random.default_rng(seed)
    n = lam.size
    q = rng.standard_normal(n)
    q /= np.linalg.norm(q)
    qp = np.zeros(n)
    bp = 0.0
    a, b = [], []
    for _ in range(m):
        w = lam * q
        al = q @ w
        w = w - al * q - bp * qp
        be = np.linalg.norm(w)
        a.append(al)
        b.append(be)
        qp, q, bp = q, w / be, be
    return np.array(a), np.array(b)


@pytest.mark.parametrize("m", [1, 2, 3, 17, 64, 240, 1500])
def test_kth_eigenvalue_matches_lapack(lib, m):
    rng = np.random.default_rng(m)
    a = rng.standard_normal(m) * 3.0
    b = np.abs(rng.standard_normal(max(0, m - 1))) + 0.01
    ref = eigvalsh_tridiagonal(a, b) if m > 1 else a.copy()
    scale = np.abs(ref).max()
    for k in sorted({0, 1 % m, m // 2, m - 1}):
        v, _ = kth(lib, a, b, k)
        assert abs(v - ref[k]) <= 1e-13 * scale, (m, k)


def test_lanczos_matrix_with_a_clustered_low_end_and_ghosts(lib):
    """The H_alpha setup late in thetaG11: 300 eigenvalues within 10 % above lambda_min, an outlier at 3e4."""
    lam = np.concatenate([0.0132 * (1.0 + 3e-4 * np.arange(300)), np.exp(np.random.default_rng(1).standard_normal(298)) * 0.3,
                          [39.3, 33694.0]])
    a, b = lanczos_T(lam, 240, 2)
    prev_min = prev_top = None
    move = 0.0
    for m in range(24, 241, 24):
        ref = eigvalsh_tridiagonal(a[:m], b[:m - 1])
        scale = abs(ref[-1])
        cold, ev_cold = kth(lib, a[:m], b[:m - 1], 0)
        warm, ev_warm = kth(lib, a[:m], b[:m - 1], 0, upper=prev_min, width=move)
        assert abs(cold - ref[0]) <= 4e-16 * scale * m            # (LAPACK and bisection agree to the backward error of either)
        assert abs(warm - cold) <= 8e-16 * scale                   # the bracket does not change the answer
        if prev_min is not None:
            assert warm <= prev_min + 1e-15 * scale                # Cauchy interlacing, what the bracket relies on
            assert ev_warm < ev_cold                               # ... and it does save Sturm counts
            move = abs(warm - prev_min)
        prev_min = warm
        # largest eigenvalue = - smallest of -T, bracketed from below
        top, _ = kth(lib, -a[:m], b[:m - 1], 0, upper=None if prev_top is None else -prev_top)
        assert abs(-top - ref[-1]) <= 1e-14 * scale
        prev_top = -top
        second, _ = kth(lib, a[:m], b[:m - 1], 1)
        assert abs(second - ref[1]) <= 4e-16 * scale * m


@pytest.mark.parametrize("hint", ["below", "far_above", "outside", "exact", "nan"])
def test_a_wrong_bracket_costs_evaluations_not_the_answer(lib, hint):
    rng = np.random.default_rng(11)
    m = 120
    a = rng.standard_normal(m)
    b = np.abs(rng.standard_normal(m - 1)) + 0.1
    ref = eigvalsh_tridiagonal(a, b)
    up = {"below": ref[0] - 0.3, "far_above": ref[0] + 2.0, "outside": 1e9, "exact": ref[0], "nan": float("nan")}[hint]
    for k in (0, 1):
        v, _ = kth(lib, a, b, k, upper=up, width=1e-9)
        assert abs(v - ref[k]) <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("scale", [1e-250, 1e-20, 1.0, 1e40, 1e250])
def test_the_polynomial_recurrence_is_rescaled(lib, scale):
    """1500 rows at entries of 1e-250 .. 1e250: the recurrence runs on T / (the power of two at its norm), and the
    characteristic polynomials, which leave the double range after some hundred rows, are rescaled: the count only needs
    their signs."""
    rng = np.random.default_rng(5)
    m = 1500
    a = (rng.standard_normal(m) + 4.0) * scale
    b = (np.abs(rng.standard_normal(m - 1)) + 0.5) * scale
    ref = eigvalsh_tridiagonal(a / scale, b / scale) * scale
    for k in (0, m - 1):
        v, _ = kth(lib, a, b, k)
        assert abs(v - ref[k]) <= 1e-12 * np.abs(ref).max()


def test_zero_off_diagonal_entries_and_repeated_eigenvalues(lib):
    a = np.array([2.0, 2.0, -1.0, 5.0, 2.0])
    b = np.array([0.0, 0.0, 0.0, 0.0])
    ref = np.sort(a)
    for k in range(5):
        v, _ = kth(lib, a, b, k)
        assert abs(v - ref[k]) <= 1e-14 * 5.0
    # W = c I at the initial point: T = (c), and the block matrix diag(c, c, c)
    v, _ = kth(lib, np.array([3.5]), np.zeros(0), 0)
    assert v == 3.5
    v, _ = kth(lib, np.full(3, 3.5), np.zeros(2), 2)
    assert abs(v - 3.5) < 1e-15


def test_argument_errors(lib):
    a = np.ones(3)
    b = np.ones(2)
    out = C.c_double(0.0)
    assert lib.lrn_dbg_tridiag_eig(3, a.ctypes.data, b.ctypes.data, 3, None, 0.0, C.byref(out), None) != 0      # k >= m
    assert lib.lrn_dbg_tridiag_eig(0, a.ctypes.data, b.ctypes.data, 0, None, 0.0, C.byref(out), None) != 0
    assert lib.lrn_dbg_tridiag_eig(3, None, b.ctypes.data, 0, None, 0.0, C.byref(out), None) != 0
