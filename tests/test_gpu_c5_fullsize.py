"""GPU, BASELINE.json's fifth configuration at FULL size: synthetic low-rank-solution SDP, matrix side 10000, 20000
sparse constraints, kit=1 PCG with the H_beta preconditioner, erank 4 (builder-defined generator with a
planted optimum, loraine.jl_amd/synthetic.py::LowRankProblem).  The CPU oracle cannot run at this size (one dense
`eigen` of side 1e4 per iteration alone), so the hot path of one real iterate is checked through properties that do not
depend on the size -- the NT identity W S W = X (src/prepare_W.jl:28-94), the pattern-restricted operator against the
two dense products (src/Solvers.jl:595-604), sampled entries of MyA(x) against the definition <A_k, W mat(AA'x) W>
(:582-614), the residual of the preconditioned CG solve (src/predictor_corrector.jl:134) -- and the whole solve must
reach the planted optimum b'y*."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MSZ, NVAR, RANK = 10000, 20000, 4


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def c5():
    import torch
    import loraine_jl_amd
    from loraine_jl_amd import resident
    from loraine_jl_amd.synthetic import LowRankProblem
    free, _ = torch.cuda.mem_get_info(0)
    if free < 120e9:
        pytest.skip("needs ~100 GB of free device memory")
    P = LowRankProblem(MSZ, NVAR, RANK)
    model = P.model()
    dev = loraine_jl_amd.Device(0)
    opts = dict(kit=1, preconditioner=2, erank=RANK, verb=0, eDIMACS=1e-5)
    solver, ha = resident.load(model, opts, device=dev)
    yield P, model, dev, solver, ha
    dev.close()


def test_c5_hot_path_of_a_real_iterate(c5):
    P, model, dev, solver, ha = c5
    from loraine_jl_amd._capi import ptr
    solver.setup_solver()
    solver.initial_point()
    for _ in range(3):                                   # three interior-point iterations in
        solver.myIPstep(ha)
        solver.tol_cg = max(solver.tol_cg * solver.tol_cg_up, solver.tol_cg_min)
        solver.check_convergence()
    assert solver.status == 0
    solver.find_mu()
    solver.prepare_W()                                   # the scaling of iteration 4
    rng = np.random.default_rng(5)
    W, eigen_free = dev.dbg_get_block(0, "W")
    X, S = dev.ip_get_iterate(0)
    # NT identity on random vectors: W S W v = X v
    V = rng.standard_normal((MSZ, 4))
    assert relerr(W @ (S @ (W @ V)), X @ V) < 1e-9
    assert np.array_equal(W, W.T)
    # the operator two ways: on the sparsity pattern of mat(AA'x) (what runs at this size) and by two dense products
    x = rng.standard_normal(NVAR)
    dev.set_option("matvec_sparse", 2)
    y_pat = dev.matvec(x)
    dev.set_option("matvec_sparse", 1)
    y_gemm = dev.matvec(x)
    dev.set_option("matvec_sparse", 0)
    assert relerr(y_pat, y_gemm) < 1e-12
    # sampled entries against the definition: (MyA x)_k = <A_k, W M W>, M = mat(AA'x) = sum_j x_j A_j  (AA = vec(A_k) rows)
    AA = model.AA[0]
    M = (AA.T @ x).reshape(MSZ, MSZ, order="F")
    M = 0.5 * (M + M.T)
    import scipy.sparse as sp
    Ms = sp.csr_matrix(M)
    for k in rng.integers(0, NVAR, 12):
        ii = P.idx[k]
        Zkk = (W[ii, :] @ Ms) @ W[:, ii]                 # 3 x 3 block of W M W on the support of A_k
        ref = float(np.sum(P.blocks[k] * Zkk))
        assert y_pat[k] == pytest.approx(ref, rel=1e-10, abs=1e-12 * np.abs(y_pat).max())
    # preconditioned CG to 1e-8: the residual of what lrn_pcg returns, measured with the operator itself
    dev.prec_setup(2, RANK, int(solver.aamat))
    h = rng.standard_normal(NVAR)
    sol, code, its = dev.pcg(h, 1e-8, 10000)
    assert code == 30 and 0 < its < 10000
    assert relerr(dev.matvec(sol), h) < 5e-8


def test_c5_solve_reaches_the_planted_optimum(c5):
    P, model, dev, solver, ha = c5
    solver.solve(ha)
    by = float(model.b @ np.ravel(solver.y))
    assert solver.status == 1
    assert by == pytest.approx(P.optimum, rel=1e-6, abs=1e-8)
    assert solver.trace[-1]["dimacs"] < 1e-5
