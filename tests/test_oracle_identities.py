"""Algebraic identities that pin the parts of the oracle the reference's own tests never
exercise (kit=1, datarank=-1) -- SURVEY.md section 8c."""
import os
import types

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import loraine_oracle as lo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _spd(m, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((m, m)) / np.sqrt(m) + np.eye(m)
    return G @ G.T, G


def _rand_model(msz, nvar, seed, kappa=3):
    rng = np.random.default_rng(seed)
    A = [[sp.csc_matrix((msz, msz))]]
    for k in range(nvar):
        nn = int(rng.integers(1, 7))
        r = rng.integers(0, msz, nn); c = rng.integers(0, msz, nn); v = rng.standard_normal(nn)
        M = sp.coo_matrix((v, (r, c)), shape=(msz, msz)).toarray()
        A[0].append(sp.csc_matrix(M + M.T))
    return lo.make_model(A, rng.standard_normal(nvar), 0.0, None, None, kappa=kappa)


@pytest.mark.parametrize("kappa", [0, 3, 100])
def test_vectorised_assembly_equals_literal_loops(kappa):
    model = _rand_model(12, 17, 1, kappa)
    W, _ = _spd(12, 2)
    H1 = lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA)
    H2 = lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA, literal=True)
    assert np.allclose(H1, H2, rtol=1e-13, atol=1e-14)
    # lower triangle equals the brute-force trace formula
    Am = np.stack([model.A[0][k + 1].toarray() for k in range(model.n)])
    T = np.stack([W @ a @ W for a in Am])
    Href = Am.reshape(model.n, -1) @ T.reshape(model.n, -1).T
    assert np.allclose(np.tril(H1), np.tril(Href), rtol=1e-12, atol=1e-13)


def test_dot_literal_is_trace_formula():
    model = _rand_model(9, 4, 3)
    W, _ = _spd(9, 4)
    A1, A2 = model.A[0][1], model.A[0][2]
    assert lo._dot(A1, A2, W) == pytest.approx(np.trace(A1.toarray() @ W @ A2.toarray() @ W), rel=1e-12)


def test_theta1_literal_path():
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    W, _ = _spd(50, 5)
    H1 = lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA)
    H2 = lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA, literal=True)
    assert np.allclose(H1, H2, rtol=1e-12, atol=1e-13)


def test_rank1_equals_general_on_rank1_data():
    rng = np.random.default_rng(6)
    msz, nvar = 10, 8
    A = [[sp.csc_matrix((msz, msz))]]
    for _ in range(nvar):
        b = np.zeros(msz)
        idx = rng.choice(msz, 3, replace=False)
        b[idx] = rng.standard_normal(3)
        A[0].append(sp.csc_matrix(np.outer(b, b)))
    model = lo.make_model(A, np.ones(nvar), 0.0, None, None, datarank=-1)
    W, G = _spd(msz, 7)
    H1 = lo.makeBBBB_rank1(nvar, 1, model.B, [G])
    H0 = lo.makeBBBBs(nvar, 1, model.A, model.AA, [W], model.qA, model.sigmaA)
    H0 = np.tril(H0) + np.tril(H0, -1).T
    assert np.allclose(H1, H0, rtol=1e-10, atol=1e-12)


def _state(model, W, G, erank=1, aamat=1, nlin=False):
    s = types.SimpleNamespace(model=model, W=[W], G=[G], erank=erank, aamat=aamat,
                              X_lin=np.zeros(0), S_lin_inv=np.zeros(0))
    return s


def test_MyA_equals_assembled_H():
    model = _rand_model(11, 9, 8)
    W, G = _spd(11, 9)
    H = lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA)
    H = np.tril(H) + np.tril(H, -1).T
    x = np.random.default_rng(0).standard_normal(model.n)
    y = np.zeros(model.n)
    lo.MyA([W], model.AA, 0, model.C_lin, np.zeros(0), np.zeros(0))(y, x)
    assert np.allclose(y, H @ x, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("erank", [1, 2, 3])
def test_smw_identity_of_H_alpha(erank):
    """(D + V V') * MyM(x) == x with V = D^0 AA (U (x) Z)  (SURVEY 8a row 13)."""
    model = _rand_model(10, 14, 10)
    W, G = _spd(10, 11)
    s = _state(model, W, G, erank=erank)
    ha = lo.Halpha(1)
    lo.Prec_for_CG_tilS_prep(s, ha)
    M = lo.MyM(model.AA, ha.AAAATtau, ha.Umat, ha.Z, ha.cholS)
    TT = np.kron(ha.Umat[0], ha.Z[0])                      # the reference's "slow formula" operand (:759)
    V = model.AA[0] @ TT
    Hal = ha.AAAATtau.toarray() + V @ V.T
    x = np.random.default_rng(1).standard_normal(model.n)
    y = np.zeros(model.n)
    M(y, x)
    assert np.allclose(Hal @ y, x, rtol=1e-9, atol=1e-10)


def test_cg_solves_and_exit_codes():
    rng = np.random.default_rng(2)
    Q = rng.standard_normal((30, 30))
    Amat = Q @ Q.T + 30 * np.eye(30)

    def A(out, x):
        out[:] = Amat @ x
    b = rng.standard_normal(30)
    x, ec, it = lo.cg(A, b, tol=1e-12, maxIter=500)
    assert ec == 30 and np.allclose(Amat @ x, b, atol=1e-9)
    assert lo.cg(A, np.zeros(30))[1:] == (1, 0)
    assert lo.cg(A, 1e-9 * np.ones(30), tol=1e-6)[1:] == (2, 0)
    assert lo.cg(A, b, tol=1e-30, maxIter=3)[1:] == (-2, 3)

    def Aneg(out, x):
        out[:] = -x
    assert lo.cg(Aneg, b, tol=1e-8)[1] == -13


def test_nt_scaling_identities():
    m = 14
    X, _ = _spd(m, 20)
    S, _ = _spd(m, 21)
    model = _rand_model(m, 2, 22)
    s = types.SimpleNamespace(model=model, X=[X.copy()], S=[S.copy()], D=[None], G=[None], Gi=[None], W=[None],
                              Si=[None], DDsi=[None], S_lin=np.zeros(0), status=0)
    lo.prepare_W(s)
    W, G, Gi, D = s.W[0], s.G[0], s.Gi[0], s.D[0]
    assert np.allclose(W @ S @ W, X, rtol=1e-10, atol=1e-11)
    assert np.allclose(G.T @ S @ G, np.diag(D), atol=1e-10)
    assert np.allclose(Gi @ X @ Gi.T, np.diag(D), atol=1e-10)
    assert np.allclose(s.Si[0] @ S, np.eye(m), atol=1e-10)
    assert np.allclose(s.DDsi[0], 1 / np.sqrt(D), rtol=1e-8)


def test_package_host_model_matches_oracle_model():
    """The product's own model construction (loraine.jl_amd/model.py) and the oracle's agree."""
    import loraine_jl_amd  # noqa: F401
    from loraine_jl_amd import model as pm
    for name, dr in (("theta1", 0), ("control1", 0), ("tru3", 0), ("maxG11", -1)):
        a = pm.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"), datarank=dr)
        b = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"), datarank=dr)
        assert a.n == b.n and a.nlin == b.nlin and a.nlmi == b.nlmi
        assert np.array_equal(a.sigmaA, b.sigmaA) and np.array_equal(a.qA, b.qA)
        assert np.array_equal(a.b, b.b) and np.array_equal(a.d_lin, b.d_lin)
        for i in range(a.nlmi):
            assert (a.AA[i] != b.AA[i]).nnz == 0 and (a.C[i] != b.C[i]).nnz == 0
        assert (a.C_lin != b.C_lin).nnz == 0
        if dr == -1:
            assert (a.B[0] != b.B[0]).nnz == 0


def test_golden_iterate_fixture_is_reproducible():
    """tests/golden/iterates_theta1.npz is what oracle/make_golden.py produces."""
    g = np.load(os.path.join(GOLD, "iterates_theta1.npz"))
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    s = lo.MySolver(model, dict(kit=0, eDIMACS=1e-6, initpoint=1, aamat=2, verb=0, maxit=3))
    lo.solve(s)
    assert np.allclose(s.X[0], g["X"], rtol=1e-9, atol=1e-12) and np.allclose(s.y, g["y"], rtol=1e-9, atol=1e-12)
    H = lo.makeBBBBs(model.n, 1, model.A, model.AA, [g["W"]], model.qA, model.sigmaA)
    assert np.allclose(np.tril(H), g["H_lower"], rtol=1e-10, atol=1e-12)


def _unpack_f32_lower(v, m):
    M = np.zeros((m, m))
    M[np.tril_indices(m)] = v.astype(np.float64)
    return M + np.tril(M, -1).T


def test_golden_c2_fixture_is_what_the_oracle_computes():
    """tests/golden/iterate_maxG11.npz (oracle/make_golden.py): from the stored float32 iterate the oracle reproduces
    the stored rank-one Schur matrix digests, right-hand side and solve -- rank-one and general assembly agreeing."""
    import scipy.linalg as sla
    g = np.load(os.path.join(GOLD, "iterate_maxG11.npz"))
    model = lo.model_from_sdpa(os.path.join(GOLD, "maxG11.dat-s"), datarank=-1)
    m = int(model.msizes[0])
    s = lo.MySolver(model, dict(kit=0, datarank=-1, verb=0))
    lo.setup_solver(s, lo.Halpha(0))
    lo.initial_point(s)
    s.X[0], s.S[0], s.y = _unpack_f32_lower(g["X_lower_f32"], m), _unpack_f32_lower(g["S_lower_f32"], m), g["y"].copy()
    lo.find_mu(s)
    lo.prepare_W(s)
    assert np.allclose(np.sort(s.D[0]), g["D_sorted"], rtol=1e-10)
    H = lo.makeBBBB_rank1(model.n, 1, model.B, s.G)
    Hs = np.tril(H) + np.tril(H, -1).T
    assert np.allclose(Hs @ g["probes"], g["H_probe"], rtol=1e-11, atol=1e-11 * np.abs(g["H_probe"]).max())
    assert np.allclose(Hs[g["H_sample_i"], g["H_sample_j"]], g["H_sample"], rtol=1e-11, atol=1e-14 * np.abs(g["H_sample"]).max())
    L = np.linalg.cholesky(Hs)
    dely = sla.solve_triangular(L.T, sla.solve_triangular(L, g["h"], lower=True), lower=False)
    assert np.allclose(dely, g["dely"], rtol=1e-8, atol=1e-10 * np.abs(g["dely"]).max())


def test_golden_c3_fixture_is_what_the_oracle_computes():
    """tests/golden/iterate_thetaG11.npz: MyA(x), MyM(x) and cg from the stored iterate (src/Solvers.jl:572-904)."""
    g = np.load(os.path.join(GOLD, "iterate_thetaG11.npz"))
    model = lo.model_from_sdpa(os.path.join(GOLD, "thetaG11.dat-s"))
    m = int(model.msizes[0])
    s = lo.MySolver(model, dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5, verb=0))
    halpha = lo.Halpha(1)
    lo.setup_solver(s, halpha)
    lo.initial_point(s)
    s.X[0], s.S[0], s.y = _unpack_f32_lower(g["X_lower_f32"], m), _unpack_f32_lower(g["S_lower_f32"], m), g["y"].copy()
    lo.find_mu(s)
    lo.prepare_W(s)
    lo.Prec_for_CG_tilS_prep(s, halpha)
    assert float(np.sqrt(halpha.AAAATtau.diagonal()[0])) == pytest.approx(float(g["tau"]), rel=1e-11)
    U = halpha.Umat[0][:, 0]
    assert np.allclose(U * np.sign(U[np.argmax(np.abs(U))]), g["Umat"], rtol=1e-8, atol=1e-10)
    assert np.allclose(halpha.Z[0] @ g["probes"], g["Z_probe"], rtol=1e-9, atol=1e-11)
    A = lo.MyA(s.W, model.AA, 0, model.C_lin, None, None)
    M = lo.MyM(model.AA, halpha.AAAATtau, halpha.Umat, halpha.Z, halpha.cholS)
    Ax, Mx = np.zeros(model.n), np.zeros(model.n)
    A(Ax, g["x"])
    M(Mx, g["x"])
    assert np.allclose(Ax, g["MyA_x"], rtol=1e-11, atol=1e-12 * np.abs(Ax).max())
    assert np.allclose(Mx, g["MyM_x"], rtol=1e-9, atol=1e-11 * np.abs(Mx).max())
    x, ec, it = lo.cg(A, g["h"], tol=float(g["cg_tols"][0]), maxIter=10000, precon=M)
    assert (ec, it) == (int(g["cg_exit"][0]), int(g["cg_iters"][0]))
    assert np.linalg.norm(x - g["cg_x"][0]) <= 1e-7 * np.linalg.norm(x)
