"""CPU: the mathematics of the eigen-free NT scaling (csrc/prepw.hip::prepare_w_ns,
csrc/ipstep.hip) restated in NumPy (tools/nt_eigenfree_proto.py) against the oracle's SVD route
(src/prepare_W.jl:28-94, src/predictor_corrector.jl:186,248-326): whole solves must walk the same iterates.  Also the
stability property the device code depends on: the coupled Newton-Schulz iteration is stable only with its products
taken literally."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import nt_eigenfree_proto as P  # noqa: E402


@pytest.mark.parametrize("name,tol", [("theta1", 1e-10), ("tru3", 1e-10), ("control1", 1e-7)])
def test_eigenfree_route_walks_the_same_iterates_as_the_svd_route(name, tol):
    opts = dict(kit=0, eDIMACS=1e-7)
    a = P.run(name, opts, False)
    b = P.run(name, opts, True)
    assert a.status == b.status == 1 and len(a.trace) == len(b.trace)
    for ta, tb in zip(a.trace, b.trace):
        assert tb["primal_obj"] == pytest.approx(ta["primal_obj"], rel=tol, abs=tol)
        assert tb["dual_obj"] == pytest.approx(ta["dual_obj"], rel=tol, abs=tol)
    assert max(P.STATS["ns"]) <= 12 and max(P.STATS["lyap"]) <= 80


def test_newton_schulz_is_stable_only_with_literal_products():
    """P = Z Y, Y <- Y T, Z <- T Z as written holds its residual at rounding level for ever; with Z Y' (what an A B'
    kernel computes from the stored, nearly symmetric matrices) the residual leaves the floor geometrically."""
    rng = np.random.default_rng(0)
    n = 120
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    K = (Q * np.exp(rng.uniform(0, np.log(1e3), n))) @ Q.T
    K = (K + K.T) / 2
    c = min(np.abs(K).sum(0).max(), np.linalg.norm(K))
    res = {}
    np.seterr(all="ignore")
    for mode in ("literal", "transposed", "P-mirrored"):
        Y, Z, ell, hist = K / c, np.eye(n), np.sqrt(2e-3), []
        for _ in range(30):
            Pm = Z @ Y if mode != "transposed" else Z @ Y.T
            if mode == "P-mirrored":                        # what the device does: lower tiles of P + mirror, T symmetric,
                Pm = np.tril(Pm) + np.tril(Pm, -1).T        # Y T and T Z literal
            hist.append(np.linalg.norm(np.eye(n) - Pm))
            if 1 - ell > 1e-9:
                a = np.sqrt(3 / (1 + ell + ell * ell)); ell = .5 * a * ell * (3 - a * a * ell * ell)
            else:
                a = 1.0
            T = a * (3 * np.eye(n) - a * a * Pm) / 2
            Y, Z = (Y @ T, T @ Z) if mode != "transposed" else (Y @ T.T, T @ Z.T)
        res[mode] = hist
    assert max(res["literal"][14:]) < 1e-11
    assert max(res["P-mirrored"][14:]) < 1e-10
    grown = np.nanmax(res["transposed"][12:]) if np.isfinite(res["transposed"][12:]).any() else np.inf
    assert not np.isfinite(res["transposed"][-1]) or grown > 1e3 * min(res["transposed"])       # (it overflows to NaN)
