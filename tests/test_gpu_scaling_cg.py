"""GPU parity: NT scaling (prepare_W), Jacobi SVD, CG operator, preconditioners and PCG
through the C ABI against the CPU oracle (reference src/prepare_W.jl, src/Solvers.jl:572-904)."""
import os
import types

import numpy as np
import pytest

from oracle import loraine_oracle as lo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    import loraine_jl_amd
    d = loraine_jl_amd.Device(0)
    yield d
    d.close()


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _spd(m, seed, cond=1e4):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    lam = np.logspace(0, np.log10(cond), m)
    return (Q * lam[None, :]) @ Q.T


@pytest.mark.parametrize("n", [5, 50, 96, 97, 300])
def test_jacobi_svd(dev, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) @ np.diag(np.logspace(0, -6, n)) @ rng.standard_normal((n, n))
    US, s, V, sweeps = dev.dbg_svd_jacobi(A)
    assert 0 < sweeps < 40
    assert relerr(A @ V, US) < 1e-12
    assert np.linalg.norm(V.T @ V - np.eye(n)) < 1e-12 * n
    sref = np.linalg.svd(A, compute_uv=False)
    assert np.allclose(np.sort(s)[::-1], sref, rtol=1e-10, atol=1e-14 * sref[0])
    U = US / s[None, :]
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 1e-9 * n


@pytest.mark.parametrize("n", [130, 257, 500])
def test_jacobi_svd_wide_blocks(dev, n):
    """32-column block kernels (auto for n >= 3000) forced at small sizes, incl. ragged last blocks."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) @ np.diag(np.logspace(0, -4, n)) @ rng.standard_normal((n, n))
    dev.set_option("jacobi_block", 32)
    try:
        US, s, V, sweeps = dev.dbg_svd_jacobi(A)
    finally:
        dev.set_option("jacobi_block", 0)
    assert 0 < sweeps < 40
    assert relerr(A @ V, US) < 1e-12
    assert np.linalg.norm(V.T @ V - np.eye(n)) < 1e-12 * n
    sref = np.linalg.svd(A, compute_uv=False)
    assert np.allclose(np.sort(s)[::-1], sref, rtol=1e-9, atol=1e-14 * sref[0])


@pytest.mark.parametrize("m,cond", [(10, 1e2), (50, 1e4), (130, 1e6), (400, 1e8)])
def test_prepare_w_matches_oracle_and_identities(dev, m, cond):
    X = _spd(m, 1, cond)
    S = _spd(m, 2, cond)
    # a one-block model just to size the context
    import scipy.sparse as sp
    A = [[sp.csc_matrix((m, m)), sp.identity(m, format="csc")]]
    model = lo.make_model(A, np.ones(1), 0.0, None, None)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, out = dev.prepare_w(0, X, S)
    assert info == 0
    sol = types.SimpleNamespace(model=model, X=[X.copy()], S=[S.copy()], D=[None], G=[None], Gi=[None], W=[None],
                                Si=[None], DDsi=[None], S_lin=np.zeros(0), status=0)
    lo.prepare_W(sol)
    tol = 1e-14 * cond * m
    assert np.allclose(np.sort(out["D"]), np.sort(sol.D[0]), rtol=1e-10)
    assert relerr(out["W"], sol.W[0]) < tol                       # W is unique
    assert relerr(out["Si"], sol.Si[0]) < tol
    G, Gi, W, D = out["G"], out["Gi"], out["W"], out["D"]
    # identities: W S W = X, G'SG = Gi X Gi' = diag(D), Gi = inv(G)
    assert relerr(W @ S @ W, X) < tol
    assert relerr(G.T @ S @ G, np.diag(D)) < tol
    assert relerr(Gi @ X @ Gi.T, np.diag(D)) < tol
    assert relerr(G @ Gi, np.eye(m)) < tol
    assert np.array_equal(W, W.T)
    assert np.allclose(np.sort(out["DDsi"]), np.sort(sol.DDsi[0]), rtol=1e-8)


@pytest.mark.parametrize("m,cond", [(400, 1e8), (700, 1e4)])
def test_jacobi_early_stop_leaves_the_same_scaling(dev, m, cond):
    """Option "jacobi_early": a sweep whose rotated column pairs were all closer to orthogonal than 3e-8 is the last one
    (what it leaves is of second order) -- same W, D and NT identities as with the confirming sweep, one sweep fewer."""
    import scipy.sparse as sp
    X = _spd(m, 3, cond)
    S = _spd(m, 4, cond)
    A = [[sp.csc_matrix((m, m)), sp.identity(m, format="csc")]]
    model = lo.make_model(A, np.ones(1), 0.0, None, None)
    res = {}
    try:
        dev.set_option("jacobi_warm", 0)            # both runs from the same (cold) start
        for early in (0.0, 3e-8):
            dev.set_option("jacobi_early", early)
            dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
            info, out = dev.prepare_w(0, X, S)
            assert info == 0
            res[early] = (out, dev.count("svd_sweeps"))
    finally:
        dev.set_option("jacobi_early", 3e-8)
        dev.set_option("jacobi_warm", 1)
    (full, sw_full), (early, sw_early) = res[0.0], res[3e-8]
    assert sw_full - 2 <= sw_early <= sw_full, (sw_early, sw_full)      # (one fewer wherever the last sweep only confirms)
    tol = 1e-14 * cond * m
    assert relerr(early["W"], full["W"]) < 1e-13 * cond ** 0.5
    assert np.allclose(np.sort(early["D"]), np.sort(full["D"]), rtol=1e-12)
    W, G, D = early["W"], early["G"], early["D"]
    assert relerr(W @ S @ W, X) < tol
    assert relerr(G.T @ S @ G, np.diag(D)) < tol


def test_prepare_w_reports_not_pd(dev):
    import scipy.sparse as sp
    m = 20
    A = [[sp.csc_matrix((m, m)), sp.identity(m, format="csc")]]
    model = lo.make_model(A, np.ones(1), 0.0, None, None)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    X = _spd(m, 1); S = _spd(m, 2)
    Xb = X.copy(); Xb[3, 3] = -1.0
    assert dev.prepare_w(0, Xb, S)[0] == 1
    Sb = S.copy(); Sb[7, 7] = -1.0
    assert dev.prepare_w(0, X, Sb)[0] == 2


def _iterate(model, kit_opts, iters=3):
    """run a few oracle IP iterations to get a realistic (X, S, W, G) iterate"""
    s = lo.MySolver(model, dict(kit_opts, verb=0, maxit=iters))
    lo.solve(s)
    return s


@pytest.mark.parametrize("name", ["theta1", "control1", "tru3"])
def test_matvec_equals_H_times_x(dev, name):
    model = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"))
    s = _iterate(model, dict(kit=0), 4)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, C_lin=model.C_lin if model.nlin else None)
    for i in range(model.nlmi):
        dev.set_scaling(i, s.W[i], s.G[i])
    if model.nlin:
        dev.set_lin(s.X_lin, s.S_lin_inv)
    H = dev.schur_assemble(0, want_H=True)
    x = np.random.default_rng(0).standard_normal(model.n)
    y = dev.matvec(x)
    assert relerr(y, H @ x) < 1e-12
    # against the oracle operator (Solvers.jl:582-614)
    yo = np.zeros(model.n)
    lo.MyA(s.W, model.AA, model.nlin, model.C_lin, s.X_lin, s.S_lin_inv)(yo, x)
    assert relerr(y, yo) < 1e-12


def test_make_rhs_matches_oracle(dev):
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    s = _iterate(model, dict(kit=0), 3)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, s.W[0], s.G[0])
    h = dev.make_rhs(s.Rp, [s.Rd[0] + s.S[0]])
    href = lo.makeRHS(1, model.AA, s.W, s.S, s.Rp, s.Rd)
    assert relerr(h, href) < 1e-12


@pytest.mark.parametrize("n,k", [(40, 3), (600, 1), (600, 5), (1800, 2)])
def test_lanczos_extremes(dev, n, k):
    """What prec_setup consumes of eigen(W) (Solvers.jl:642-650,706-722): k largest pairs,
    lambda_min, trace -- on a spectrum shaped like W late in an IP run."""
    rng = np.random.default_rng(n + k)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(rng.uniform(np.log(1e-3), np.log(1.0), n))
    lam[-6:] = [30.0, 80.0, 200.0, 900.0, 4000.0, 2.0e4]
    W = (Q * lam) @ Q.T
    W = 0.5 * (W + W.T)
    ev, V = np.linalg.eigh(W)
    lt, U, lmin, tr, steps = dev.dbg_lanczos(W, k)
    assert steps <= n
    assert np.allclose(lt, ev[-k:], rtol=1e-9)
    assert abs(tr - np.trace(W)) <= 1e-11 * abs(np.trace(W))
    # Ritz value from above; its accuracy is absolute in ||W||, which is all tau needs (:646-650)
    assert ev[0] - 1e-9 * ev[-1] <= lmin <= ev[0] + (1e-12 if n == steps else 5e-2)
    assert np.allclose(np.abs(np.sum(U * V[:, -k:], axis=0)), 1.0, atol=1e-8)
    assert np.allclose(U.T @ U, np.eye(k), atol=1e-10)


@pytest.mark.parametrize("n", [600, 801])
def test_lanczos_extremes_resident_steps_are_the_launched_ones(dev, n):
    """The plain-recurrence route of the H_alpha setup (k = 1) with resident launches of 24 steps (option lz_resident,
    default) and with one launch per step: same Ritz values, same Ritz vector, same step count, bit for bit -- also on a
    spectrum whose low end is a cluster (the 240-step cap of the route, late in thetaG11)."""
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    for lam in (np.concatenate([np.exp(rng.uniform(np.log(1e-3), np.log(1.0), n - 1)), [2.0e4]]),
                np.concatenate([0.0132 * (1.0 + 3e-4 * np.arange(n // 3)), np.exp(rng.standard_normal(n - n // 3 - 2)) * 0.3, [39.3, 33694.0]])):
        W = (Q * lam) @ Q.T
        W = 0.5 * (W + W.T)
        out = []
        for res in (0, 1):
            dev.set_option("lz_resident", res)
            try:
                out.append(dev.dbg_lanczos(W, 1))
            finally:
                dev.set_option("lz_resident", 1)
        assert dev.count("lz_persist_abort") == 0
        (lt0, U0, lmin0, tr0, st0), (lt1, U1, lmin1, tr1, st1) = out
        assert st0 == st1 and lmin0 == lmin1 and tr0 == tr1
        assert np.array_equal(lt0, lt1) and np.array_equal(U0, U1)


def test_lanczos_invariant_subspace_restart(dev):
    """W = c I at the initial point (Solvers.jl:448-460): every Krylov space is 1-dimensional."""
    lt, U, lmin, tr, steps = dev.dbg_lanczos(3.0 * np.eye(50), 2)
    assert np.allclose(lt, 3.0) and lmin == pytest.approx(3.0) and tr == pytest.approx(150.0)
    assert np.allclose(U.T @ U, np.eye(2), atol=1e-12)
    W = np.diag([2.0] * 30 + [7.0, 11.0])           # three distinct eigenvalues, k = 2
    lt, U, lmin, tr, steps = dev.dbg_lanczos(W, 2)
    assert np.allclose(lt, [7.0, 11.0]) and lmin == pytest.approx(2.0)
    assert np.allclose(np.abs(U[30, 0]), 1.0) and np.allclose(np.abs(U[31, 1]), 1.0)


@pytest.mark.parametrize("prec,erank,eig", [(0, 1, 1), (2, 1, 1), (1, 1, 1), (1, 3, 1), (2, 1, 2), (1, 1, 2), (1, 3, 2)])
def test_preconditioner_apply_and_pcg(dev, prec, erank, eig):
    dev.set_option("prec_eig", eig)        # 1: Jacobi eigendecomposition, 2: Lanczos extremes
    try:
        _preconditioner_apply_and_pcg(dev, prec, erank, 1e-9 if eig == 1 else 1e-7)
    finally:
        dev.set_option("prec_eig", 0)


def _preconditioner_apply_and_pcg(dev, prec, erank, tol_apply):
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    s = _iterate(model, dict(kit=1, preconditioner=max(prec, 1) if prec else 0, erank=erank), 6)
    X, S = s.X[0], s.S[0]
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, out = dev.prepare_w(0, X, S)
    assert info == 0
    # oracle state with the device's scaling (W is unique; G is not -> use the device's)
    s.W[0], s.G[0] = out["W"], out["G"]
    s.preconditioner, s.erank = prec, erank
    ha = lo.Halpha(1)
    if prec == 1:
        lo.Prec_for_CG_tilS_prep(s, ha)
        Mo = lo.MyM(model.AA, ha.AAAATtau, ha.Umat, ha.Z, ha.cholS)
    elif prec == 2:
        lo.Prec_for_CG_beta(s, ha)
        Mo = lo.MyM_beta(model.AA, ha.AAAATtau)
    else:
        Mo = lo.MyM_no()
    assert dev.prec_setup(prec, erank, s.aamat) == 0
    rng = np.random.default_rng(1)
    x = rng.standard_normal(model.n)
    ref = np.zeros(model.n)
    Mo(ref, x)
    got = dev.prec_apply(x)
    assert relerr(got, ref) < tol_apply
    # PCG against the oracle's cg on the same operator
    h = rng.standard_normal(model.n)
    Ao = lo.MyA(s.W, model.AA, 0, model.C_lin, s.X_lin, s.S_lin_inv)
    xr, ec_r, it_r = lo.cg(Ao, h, tol=1e-8, maxIter=10000, precon=Mo)
    xg, ec_g, it_g = dev.pcg(h, 1e-8, 10000)
    assert ec_g == ec_r == 30
    assert abs(it_g - it_r) <= max(2, it_r // 10)
    assert relerr(xg, xr) < 1e-6
    res = np.zeros(model.n); Ao(res, xg)
    assert np.linalg.norm(res - h) / np.linalg.norm(h) < 2e-8


def test_pcg_trivial_exits(dev):
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    s = _iterate(model, dict(kit=0), 2)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, s.W[0], s.G[0])
    dev.prec_setup(0, 1, 1)
    x, ec, it = dev.pcg(np.zeros(model.n), 1e-6)
    assert (ec, it) == (1, 0) and not x.any()
    x, ec, it = dev.pcg(1e-9 * np.ones(model.n), 1e-6)
    assert (ec, it) == (2, 0)


def test_row_sharded_matvec_sums_to_full(dev):
    """kit=1 multi-GPU operator: the partial mat-vecs of a 3-way row sharding add up to MyA(x)."""
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    s = _iterate(model, dict(kit=0), 3)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, s.W[0], s.G[0])
    x = np.random.default_rng(3).standard_normal(model.n)
    full = dev.matvec(x)
    import ctypes as C
    from loraine_jl_amd._capi import ptr
    acc = np.zeros(model.n)
    for r in range(3):
        dev.set_shard(r, 3)
        part = np.zeros(model.n)
        dev._chk(dev.lib.lrn_matvec_partial(dev.h, ptr(x), ptr(part)), "lrn_matvec_partial")
        acc += part
    dev.set_shard(0, 1)
    assert relerr(acc, full) < 1e-13
    # the same sum inside the library: with a communicator lrn_matvec is "this rank's share + all-reduce".  Three ranks
    # emulated in this process: the host all-reduce callback adds the shares a twin context computes for ranks 1 and 2
    import loraine_jl_amd as _l
    twin = _l.Device(0)
    twin.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    twin.set_scaling(0, s.W[0], s.G[0])
    state = {}

    def allreduce(buf, op):
        assert op == 0 and buf.size == model.n
        for r in (1, 2):
            twin.set_shard(r, 3)
            part = np.zeros(model.n)
            twin._chk(twin.lib.lrn_matvec_partial(twin.h, ptr(state["x"]), ptr(part)), "lrn_matvec_partial")
            buf += part

    dev.comm_init_host(0, 3, allreduce, lambda send, recv: None)
    try:
        state["x"] = x
        assert relerr(dev.matvec(x), full) < 1e-13
    finally:
        dev.comm_destroy()
        twin.close()
    assert relerr(dev.matvec(x), full) == 0.0          # and back to the plain operator


def test_rccl_communicator_world_size_one(dev):
    """lrn_comm_unique_id / lrn_comm_init (ncclCommInitRank on the context's device) and a collective on the library's
    stream with a communicator of one rank; the hot path is unchanged by it (kit=0 assembly + solve, kit=1 PCG)."""
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    s = _iterate(model, dict(kit=0), 3)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, s.W[0], s.G[0])
    dev.prec_setup(0, 1, 1)
    h = np.random.default_rng(4).standard_normal(model.n)
    x0, ec0, it0 = dev.pcg(h, 1e-8)
    H0 = dev.schur_assemble(0, want_H=True)
    uid = dev.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    dev.comm_init(uid, 0, 1)
    try:
        v = np.arange(5, dtype=float)
        assert (dev.comm_allreduce(v.copy(), 0) == v).all()       # ncclAllReduce over one rank
        x1, ec1, it1 = dev.pcg(h, 1e-8)
        H1 = dev.schur_assemble(0, want_H=True)
    finally:
        dev.comm_destroy()
    assert (ec1, it1) == (ec0, it0) and (x1 == x0).all() and (H1 == H0).all()


@pytest.mark.parametrize("case", ["thetaG11", "lowrank"])
def test_sparse_pattern_matvec_matches_gemm_matvec(dev, case):
    """MyA (Solvers.jl:582-614) two ways: W mat(AA'x) W by two dense GEMMs, or only on the sparsity
    pattern the constraints read (option matvec_sparse); also column-sharded partial sums."""
    from loraine_jl_amd._capi import ptr
    rng = np.random.default_rng(11)
    if case == "thetaG11":
        model = lo.model_from_sdpa(os.path.join(GOLD, "thetaG11.dat-s"))
    else:
        from loraine_jl_amd.synthetic import LowRankProblem
        model = LowRankProblem(300, 500, 3, seed=5).model()
    m = int(model.msizes[0])
    Gm = rng.standard_normal((m, m)) / np.sqrt(m) + np.diag(np.exp(rng.uniform(-2, 2, m)))
    W = Gm @ Gm.T
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, W, Gm)
    x = rng.standard_normal(model.n)
    try:
        dev.set_option("matvec_sparse", 1)
        dense = dev.matvec(x)
        dev.set_option("matvec_sparse", 2)
        sparse = dev.matvec(x)
        acc = np.zeros(model.n)
        for r in range(3):
            dev.set_shard(r, 3)
            part = np.zeros(model.n)
            dev._chk(dev.lib.lrn_matvec_partial(dev.h, ptr(x), ptr(part)), "lrn_matvec_partial")
            acc += part
    finally:
        dev.set_shard(0, 1)
        dev.set_option("matvec_sparse", 0)
    Mx = model.AA[0].T @ x
    Mx = Mx.reshape(m, m, order="F"); Mx = 0.5 * (Mx + Mx.T)
    ref = model.AA[0] @ (W @ Mx @ W).reshape(-1, order="F")
    assert relerr(dense, ref) < 1e-12
    assert relerr(sparse, ref) < 1e-12
    assert relerr(acc, ref) < 1e-12


@pytest.mark.parametrize("name,erank", [("tru3", 1), ("vib3", 2)])
def test_halpha_apply_with_linear_rows(dev, name, erank):
    """MyM (Solvers.jl:866-904) when AAAATtau is not diagonal (nlin > 0, :743-745): the device keeps its
    dense Cholesky factor; the oracle solves with the sparse matrix like the reference."""
    model = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"))
    s = _iterate(model, dict(kit=1, preconditioner=1, erank=erank), 5)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, C_lin=model.C_lin)
    for i in range(model.nlmi):
        info, out = dev.prepare_w(i, s.X[i], s.S[i])
        assert info == 0
        s.W[i], s.G[i] = out["W"], out["G"]
    dev.set_lin(s.X_lin, s.S_lin_inv)
    s.preconditioner, s.erank = 1, erank
    ha = lo.Halpha(1)
    lo.Prec_for_CG_tilS_prep(s, ha)
    Mo = lo.MyM(model.AA, ha.AAAATtau, ha.Umat, ha.Z, ha.cholS)
    dev.set_option("prec_eig", 1)
    try:
        assert dev.prec_setup(1, erank, s.aamat) == 0
    finally:
        dev.set_option("prec_eig", 0)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(model.n)
    ref = np.zeros(model.n); Mo(ref, x)
    assert relerr(dev.prec_apply(x), ref) < 1e-9
    h = rng.standard_normal(model.n)
    Ao = lo.MyA(s.W, model.AA, model.nlin, model.C_lin, s.X_lin, s.S_lin_inv)
    xr, ec_r, it_r = lo.cg(Ao, h, tol=1e-8, maxIter=10000, precon=Mo)
    xg, ec_g, it_g = dev.pcg(h, 1e-8, 10000)
    assert ec_g == ec_r == 30 and abs(it_g - it_r) <= max(2, it_r // 10)
    assert relerr(xg, xr) < 1e-6


def test_partial_matvec_two_blocks_with_linear_rows(dev):
    """kit=1 multi-GPU operator on vib3 (two LMI blocks, 72 linear rows): the partial mat-vecs of a 4-way
    sharding add up to MyA(x); the C_lin term is contributed by rank 0 only."""
    from loraine_jl_amd._capi import ptr
    model = lo.model_from_sdpa(os.path.join(GOLD, "vib3.dat-s"))
    s = _iterate(model, dict(kit=0), 4)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, C_lin=model.C_lin)
    for i in range(model.nlmi):
        dev.set_scaling(i, s.W[i], s.G[i])
    dev.set_lin(s.X_lin, s.S_lin_inv)
    x = np.random.default_rng(5).standard_normal(model.n)
    full = dev.matvec(x)
    ref = np.zeros(model.n)
    lo.MyA(s.W, model.AA, model.nlin, model.C_lin, s.X_lin, s.S_lin_inv)(ref, x)
    assert relerr(full, ref) < 1e-12
    acc = np.zeros(model.n)
    try:
        for r in range(4):
            dev.set_shard(r, 4)
            part = np.zeros(model.n)
            dev._chk(dev.lib.lrn_matvec_partial(dev.h, ptr(x), ptr(part)), "lrn_matvec_partial")
            acc += part
    finally:
        dev.set_shard(0, 1)
    assert relerr(acc, ref) < 1e-12
