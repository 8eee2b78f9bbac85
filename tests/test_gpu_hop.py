"""GPU parity: the CG operator MyA (reference src/Solvers.jl:582-614) applied through the ASSEMBLED Schur matrix of
src/makeBBBB.jl:67-218 (csrc/hop.hip: one pass over the lower triangle per application) against the matrix-free forms
(pattern-restricted, two dense products) and the oracle; lrn_pcg through it; the sharded share; the cost model's switch
inside a solve."""
import os

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


def _scaling(m, rng):
    Gm = rng.standard_normal((m, m)) / np.sqrt(m) + np.diag(np.exp(rng.uniform(-2, 2, m)))
    return Gm @ Gm.T, Gm


def _lowrank(msz, nvar, seed):
    from loraine_jl_amd.synthetic import LowRankProblem
    return LowRankProblem(msz, nvar, 3, seed=seed).model()


# n = 2401 odd (8-byte loads), 500 even (16-byte loads), 9000 / 9001: the 512-row tiles of n >= 8192
@pytest.mark.parametrize("case", ["thetaG11", "lowrank500", "lowrank9000", "lowrank9001"])
def test_operator_by_H_by_pattern_and_by_gemms_agree(dev, case):
    rng = np.random.default_rng(17)
    if case == "thetaG11":
        model = lo.model_from_sdpa(os.path.join(GOLD, "thetaG11.dat-s"))
    else:
        model = _lowrank(300, int(case[7:]), 5)
    m = int(model.msizes[0])
    W, Gm = _scaling(m, rng)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, W, Gm)
    xs = [rng.standard_normal(model.n), np.eye(model.n)[:, model.n - 1].copy(), np.ones(model.n)]
    try:
        for x in xs:
            dev.set_option("matvec_h", 1)
            dev.set_option("matvec_sparse", 1)
            dense = dev.matvec(x)
            dev.set_option("matvec_sparse", 2)
            sparse = dev.matvec(x)
            dev.set_option("matvec_h", 2)
            n0 = dev.count("hop_matvec")
            byh = dev.matvec(x)
            assert dev.count("hop_matvec") == n0 + 1
            Mx = model.AA[0].T @ x
            Mx = Mx.reshape(m, m, order="F"); Mx = 0.5 * (Mx + Mx.T)
            ref = model.AA[0] @ (W @ Mx @ W).reshape(-1, order="F")
            for got in (dense, sparse, byh):
                assert relerr(got, ref) < 1e-12
            assert relerr(byh, sparse) < 1e-12 and relerr(byh, dense) < 1e-12
        assert dev.count("hop_assemble") >= 1
        # a new scaling: the matrix is assembled again, not re-used
        W2, G2 = _scaling(m, rng)
        dev.set_scaling(0, W2, G2)
        a0 = dev.count("hop_assemble")
        y2 = dev.matvec(xs[0])
        assert dev.count("hop_assemble") == a0 + 1
        dev.set_option("matvec_h", 1)
        assert relerr(y2, dev.matvec(xs[0])) < 1e-12
    finally:
        dev.set_option("matvec_h", 0)
        dev.set_option("matvec_sparse", 0)


@pytest.mark.parametrize("name", ["control1", "tru3"])
def test_operator_by_H_with_several_blocks_and_linear_rows(dev, name):
    """nlmi > 1 (H in natural order, no permutation) and the C_lin term of Solvers.jl:609 inside the assembled matrix."""
    model = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"))
    s = lo.MySolver(model, dict(kit=0, verb=0, maxit=4))
    lo.solve(s)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, C_lin=model.C_lin if model.nlin else None)
    for i in range(model.nlmi):
        dev.set_scaling(i, s.W[i], s.G[i])
    if model.nlin:
        dev.set_lin(s.X_lin, s.S_lin_inv)
    x = np.random.default_rng(0).standard_normal(model.n)
    yo = np.zeros(model.n)
    lo.MyA(s.W, model.AA, model.nlin, model.C_lin, s.X_lin, s.S_lin_inv)(yo, x)
    try:
        dev.set_option("matvec_h", 1)
        free = dev.matvec(x)
        dev.set_option("matvec_h", 2)
        byh = dev.matvec(x)
    finally:
        dev.set_option("matvec_h", 0)
    assert relerr(free, yo) < 1e-12 and relerr(byh, yo) < 1e-12


def test_pcg_through_H_keeps_the_iteration_counts_of_the_golden_iterate(dev):
    """C3 (thetaG11, H_alpha, erank 1): cg exit code, iteration count and solution with the operator applied through the
    assembled matrix = those of the matrix-free operator = the oracle's (tests/golden/iterate_thetaG11.npz)."""
    from loraine_jl_amd.model import model_from_sdpa
    g = np.load(os.path.join(GOLD, "iterate_thetaG11.npz"))
    model = model_from_sdpa(os.path.join(GOLD, "thetaG11.dat-s"), datarank=0)
    m = int(model.msizes[0])

    def unpack(lower_f32):
        M = np.zeros((m, m))
        M[np.tril_indices(m)] = lower_f32.astype(np.float64)
        return M + np.tril(M, -1).T

    X, S = unpack(g["X_lower_f32"]), unpack(g["S_lower_f32"])
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, _ = dev.prepare_w(0, X, S)
    assert info == 0
    assert dev.prec_setup(1, 1, 1) == 0
    h = g["h"]
    res = {}
    try:
        for mode in (1, 2):
            dev.set_option("matvec_h", mode)
            res[mode] = [dev.pcg(h, float(tol)) for tol in g["cg_tols"]]
    finally:
        dev.set_option("matvec_h", 0)
    assert dev.count("hop_matvec") > 0
    for (x1, e1, i1), (x2, e2, i2), ec, it, tol in zip(res[1], res[2], g["cg_exit"], g["cg_iters"], g["cg_tols"]):
        assert (e1, i1) == (e2, i2) == (int(ec), int(it))
        assert relerr(x2, x1) < 0.1 * float(tol) + 1e-11


def test_sharded_share_of_the_H_operator_sums_to_the_operator(dev):
    """One process per GPU: every rank multiplies the Schur column blocks it assembled, one all-reduce of the nvar-vector.
    Three ranks emulated in this process through the host transport: the all-reduce callback adds the shares a twin
    context computes for ranks 1 and 2."""
    import loraine_jl_amd as _l
    model = _lowrank(120, 1100, 9)
    m = int(model.msizes[0])
    W, Gm = _scaling(m, np.random.default_rng(3))
    x = np.random.default_rng(4).standard_normal(model.n)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_scaling(0, W, Gm)
    dev.set_option("matvec_h", 1)
    full = dev.matvec(x)
    twin = _l.Device(0)
    twin.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    twin.set_scaling(0, W, Gm)
    twin.set_option("matvec_h", 2)
    shares = {}

    def twin_share(r):
        # rank r of 3 on the twin context: its own communicator whose all-reduce adds nothing -> the bare share
        def ar(buf, op):
            if op == 0 and buf.size == model.n:
                shares[r] = buf.copy()
        twin.comm_init_host(r, 3, ar, lambda send, recv: None)
        try:
            twin.set_scaling(0, W, Gm)          # (new scaling version: the share is assembled for this rank)
            twin.matvec(x)
        finally:
            twin.comm_destroy()

    def allreduce(buf, op):
        if op == 0 and buf.size == model.n:
            for r in (1, 2):
                buf += shares[r]

    try:
        for r in (1, 2):
            twin_share(r)
        dev.set_option("matvec_h", 2)
        dev.comm_init_host(0, 3, allreduce, lambda send, recv: None)
        try:
            got = dev.matvec(x)
            assert dev.count("hop_matvec") >= 1
        finally:
            dev.comm_destroy()
    finally:
        dev.set_option("matvec_h", 0)
        twin.close()
    assert relerr(got, full) < 1e-12


def test_solve_switches_to_H_by_the_cost_model_and_agrees(dev):
    """A kit=1 solve of a low-rank problem: the cost model moves the operator to the assembled matrix once the CG
    iterations of an IP iteration pay for the assembly; the optimum is that of the matrix-free run."""
    from loraine_jl_amd import resident
    from loraine_jl_amd.synthetic import LowRankProblem
    P = LowRankProblem(400, 2000, 3, seed=21)
    model = P.model()
    out = {}
    for mode in (1, 0):
        dev.set_option("matvec_h", mode)
        dev.reset_timing()
        try:
            s, ha = resident.load(model, dict(kit=1, preconditioner=2, erank=3, verb=0, eDIMACS=1e-6, tol_cg_min=1e-8), device=dev)
            s.solve(ha)
        finally:
            dev.set_option("matvec_h", 0)
        out[mode] = (s.status, float(model.b @ np.ravel(s.y)), dev.count("hop_assemble"), dev.count("hop_matvec"), s.cg_iter_tot)
    assert out[1][0] == 1 and out[0][0] == 1
    assert out[1][2] == 0 and out[1][3] == 0
    assert out[0][2] > 0 and out[0][3] > 0          # (at this size the assembly pays only in the last iterations)
    assert abs(out[0][1] - out[1][1]) < 1e-6 * (1 + abs(out[1][1]))
    assert abs(out[0][1] - P.optimum) < 1e-5 * (1 + abs(P.optimum))


def test_halpha_as_one_dense_matrix_equals_the_smw_apply(dev):
    """H_alpha (MyM, reference src/Solvers.jl:866-904) formed once per scaling as the dense symmetric matrix
    D^-1/2 (I - ts (S + I)^-1 ts') D^-1/2 and applied by one pass over its lower triangle (lrn_pcg takes it when the CG
    iterations pay for the products; option prec_dense = 2 forces it): same M^-1 x as the SMW form, same cg exit codes,
    iteration counts and solutions on the golden thetaG11 iterate -- alone and together with the operator through H."""
    from loraine_jl_amd.model import model_from_sdpa
    g = np.load(os.path.join(GOLD, "iterate_thetaG11.npz"))
    model = model_from_sdpa(os.path.join(GOLD, "thetaG11.dat-s"), datarank=0)
    m = int(model.msizes[0])

    def unpack(lower_f32):
        M = np.zeros((m, m))
        M[np.tril_indices(m)] = lower_f32.astype(np.float64)
        return M + np.tril(M, -1).T

    X, S = unpack(g["X_lower_f32"]), unpack(g["S_lower_f32"])
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, _ = dev.prepare_w(0, X, S)
    assert info == 0
    res = {}
    try:
        for mode, mh in ((1, 1), (2, 1), (2, 2)):
            dev.set_option("prec_dense", mode)
            dev.set_option("matvec_h", mh)
            assert dev.prec_setup(1, 1, 1) == 0
            n0 = dev.count("prec_dense_apply")
            mx = dev.prec_apply(g["x"])
            assert (dev.count("prec_dense_apply") > n0) == (mode == 2)
            res[(mode, mh)] = (mx, [dev.pcg(g["h"], float(tol)) for tol in g["cg_tols"]])
    finally:
        dev.set_option("prec_dense", 0)
        dev.set_option("matvec_h", 0)
    smw = res[(1, 1)]
    assert relerr(smw[0], g["MyM_x"]) < 1e-10
    for key in ((2, 1), (2, 2)):
        assert relerr(res[key][0], smw[0]) < 1e-11
        assert relerr(res[key][0], g["MyM_x"]) < 1e-10
        for (x1, e1, i1), (x2, e2, i2), ec, it, tol in zip(smw[1], res[key][1], g["cg_exit"], g["cg_iters"], g["cg_tols"]):
            assert (e1, i1) == (e2, i2) == (int(ec), int(it))
            assert relerr(x2, x1) < 0.1 * float(tol) + 1e-11


def test_right_hand_sides_through_the_pattern_of_the_constraints(dev):
    """makeRHS and the corrector right-hand side (reference src/makeBBBB.jl:221-228, src/predictor_corrector.jl:186) need
    W M W only at the entries the constraints read; with every constraint sparse (C5) the second n^3 product is replaced by
    dots on the pattern (by default from msz 1500 on; forced here): same trajectory."""
    from loraine_jl_amd import resident
    from loraine_jl_amd.synthetic import LowRankProblem
    P = LowRankProblem(120, 300, 3, seed=4)
    model = P.model()
    out = {}
    for forced in (0, 1):
        dev.set_option("wmw_pattern_min", 8 if forced else 1500)
        try:
            s, ha = resident.load(model, dict(kit=1, preconditioner=2, erank=3, verb=0, eDIMACS=1e-7, tol_cg_min=1e-10, tol_cg=1e-10), device=dev)
            s.solve(ha)
        finally:
            dev.set_option("wmw_pattern_min", 1500)
        assert s.status == 1
        out[forced] = [t["primal_obj"] for t in s.trace]
    assert len(out[0]) == len(out[1])
    for a, b in zip(out[0], out[1]):
        assert a == pytest.approx(b, rel=1e-8, abs=1e-10)
    assert abs(float(model.b @ np.ravel(s.y)) - P.optimum) < 1e-6 * (1 + abs(P.optimum))
