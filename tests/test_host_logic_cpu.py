"""Host-side logic that needs no GPU: the model builder against the oracle's, the optimizer surface
(reference src/MOI_wrapper.jl:86-103 raw attributes, :252-265 status mapping), the sharding rules and
the synthetic low-rank generator."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import loraine_oracle as lo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,datarank", [("theta1", 0), ("control1", 0), ("tru3", 0), ("vib3", 0), ("maxG11", -1)])
def test_model_builder_matches_oracle(name, datarank):
    """product `model.py` and the oracle read SDPA files independently: same AA, C, C_lin, d_lin, b,
    sigmaA, qA, B (src/model.jl:120-229, src/MOI_wrapper.jl:142-232)."""
    from loraine_jl_amd.model import model_from_sdpa
    path = os.path.join(GOLD, f"{name}.dat-s")
    a = model_from_sdpa(path, datarank=datarank)
    b = lo.model_from_sdpa(path, datarank=datarank)
    assert a.n == b.n and a.nlmi == b.nlmi and a.nlin == b.nlin
    assert a.msizes.tolist() == b.msizes.tolist()
    assert np.array_equal(a.b, b.b) and np.array_equal(a.d_lin, b.d_lin)
    assert abs(a.C_lin - b.C_lin).sum() == 0
    assert np.array_equal(a.sigmaA, b.sigmaA) and np.array_equal(a.qA, b.qA) and np.array_equal(a.nzA, b.nzA)
    for i in range(a.nlmi):
        assert abs(a.AA[i] - b.AA[i]).sum() == 0
        assert abs(a.C[i] - b.C[i]).sum() == 0
    if datarank == -1:
        for i in range(a.nlmi):
            assert abs(a.B[i] - b.B[i]).max() < 1e-14


def test_optimizer_surface_without_gpu():
    from loraine_jl_amd import solvers
    from loraine_jl_amd.optimizer import OPTIMIZE_NOT_CALLED, NO_SOLUTION, Optimizer, UnsupportedAttribute
    o = Optimizer()
    assert o.solver_name() == "Loraine"
    # the 15 raw attributes of the reference (src/Solvers.jl:169-185)
    assert sorted(solvers.DEFAULT_OPTIONS) == sorted(
        ["kit", "tol_cg", "tol_cg_up", "tol_cg_min", "eDIMACS", "preconditioner", "erank", "aamat", "fig_ev",
         "verb", "datarank", "initpoint", "timing", "maxit", "datasparsity"])
    for k in ("kit", "tol_cg", "eDIMACS", "preconditioner", "erank", "aamat", "datarank", "initpoint", "maxit"):
        assert o.supports(k)
    o.set_attribute("kit", 1)
    assert o.get_attribute("kit") == 1
    with pytest.raises(UnsupportedAttribute):
        o.set_attribute("not_an_option", 0)
    with pytest.raises(UnsupportedAttribute):
        o.get_attribute("not_an_option")
    assert o.termination_status() == OPTIMIZE_NOT_CALLED
    assert o.primal_status() == NO_SOLUTION and o.dual_status() == NO_SOLUTION and o.result_count() == 0
    with pytest.raises(RuntimeError):
        o.optimize()


def test_shard_block_rule():
    from loraine_jl_amd import sharding as sh
    assert sh.auto_bs(4000, 1) == 128
    assert (sh.auto_bs(4000, 2), sh.auto_bs(4000, 4), sh.auto_bs(4000, 8)) == (1024, 512, 256)
    assert sh.auto_bs(100, 8) == 128 and sh.auto_bs(0, 4) == 128
    for nvar, world in [(4000, 8), (4000, 3), (1000, 4), (129, 2)]:
        bs = sh.auto_bs(nvar, world)
        cols = [sh.owned_columns(nvar, r, world, bs) for r in range(world)]
        assert sorted(c for cs in cols for c in cs) == list(range(nvar))          # a partition
        # lower-triangle work (column c carries nvar - c entries) is balanced to within one block
        work = [sum(nvar - c for c in cs) for cs in cols]
        assert max(work) - min(work) <= 2 * bs * nvar
        # pack / unpack with this width is a round trip
        H = np.random.default_rng(nvar).standard_normal((nvar, nvar)) if nvar <= 1000 else None
        if H is not None:
            buf = np.concatenate([sh.pack_shard(H, r, world, bs) for r in range(world)])
            assert np.array_equal(sh.unpack_all(buf, nvar, world, bs), H)


def test_lowrank_generator_is_consistent():
    """C5 generator: planted pair is feasible, complementary and optimal (weak duality closes)."""
    from loraine_jl_amd.synthetic import LowRankProblem
    P = LowRankProblem(60, 90, 3, seed=11)
    Xs = (P.Q * P.lam) @ P.Q.T
    Ss = np.eye(60) - P.Q @ P.Q.T
    assert np.linalg.norm(Xs @ Ss) < 1e-12 and np.trace(Xs) == pytest.approx(np.sqrt(60))
    AA = P.AA()
    assert np.allclose(AA @ Xs.reshape(-1, order="F"), P.b)                         # <M_k, X*> = b_k
    Cd = P.C_dense()
    M = (AA.T @ P.ystar).reshape(60, 60, order="F")
    assert np.allclose(Cd - M, Ss, atol=1e-12)                                       # S* = C - sum y*_k M_k
    assert P.optimum == pytest.approx(float(np.vdot(Cd, Xs)))                        # b'y* = <C, X*>
    for k in (0, 17, 89):                                                            # symmetric, traceless, 9 nnz
        Mk = P.constraint(k).toarray()
        assert np.array_equal(Mk, Mk.T) and abs(np.trace(Mk)) < 1e-14 and np.count_nonzero(Mk) == 9
    m = P.model()
    assert m.n == 90 and m.nlmi == 1 and m.nlin == 0 and int(m.qA[0, 0]) == 90


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no torchrun environment) must start N ranks itself, before anything
    touches a GPU, and exit with their code; a WORLD_SIZE that contradicts --gpus is an error."""
    import subprocess
    import sys
    import bench
    calls = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.setitem(sys.modules, "torch", None)          # the launcher must not need torch: import would fail
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                  # the ranks' exit code is the launcher's
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher with another world size: refuse
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_bench_dense_instance_is_the_oracles_model():
    """bench.py feeds the oracle a dense instance built without the oracle's (slow, per-entry) model builder: AA, sigmaA
    and qA must be exactly what `make_model` produces (src/model.jl:120-229), and the bounded constraint loop
    (`ii_stop`) must compute the first columns of the full Schur matrix."""
    import bench
    msz, nvar = 40, 30
    rows, AA, sigmaA, qA, rng = bench._dense_instance(msz, nvar, 3)
    A = [[sp.csc_matrix((msz, msz))] + [sp.csc_matrix(rows[k].reshape(msz, msz)) for k in range(nvar)]]
    model = lo.make_model(A, np.zeros(nvar), 0.0, None, None)
    assert abs(model.AA[0] - AA).max() == 0
    assert np.array_equal(model.sigmaA, sigmaA) and np.array_equal(model.qA, qA)
    W, _ = bench.make_scaling(msz, 4)
    H = lo.makeBBBBs(nvar, 1, model.A, model.AA, [W], model.qA, model.sigmaA)
    Hs = lo.makeBBBBsi(0, bench._DenseRows(rows, msz), AA, W, nvar, qA, sigmaA, ii_stop=7)
    assert np.array_equal(Hs[:, :7], H[:, :7])
    assert np.count_nonzero(Hs[7:, 7:]) == 0                       # nothing beyond the sampled constraints
