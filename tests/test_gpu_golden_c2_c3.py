"""BASELINE configs C2 (maxG11, kit=0, datarank=-1) and C3 (thetaG11, kit=1, H_alpha, erank=1) against the committed
fixtures of oracle/make_golden.py: the hot path on a REAL iterate of each solve (inputs stored as float32 lower
triangles, outputs computed by the oracle in float64 from exactly those inputs) and the per-iteration trace of the whole
solves.  These are the halves of the path no test of the reference runs (src/makeBBBB.jl:1-20,
src/predictor_corrector.jl:119-139,225-238, src/Solvers.jl:572-904); the fixtures move them from synthetic-W checks to
the north-star tolerance at the iterates the solver actually visits."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def _unpack(lower_f32, m):
    M = np.zeros((m, m))
    M[np.tril_indices(m)] = lower_f32.astype(np.float64)
    return M + np.tril(M, -1).T


def _mat(v, m):
    M = np.asarray(v).reshape(m, m, order="F")
    return (M + M.T) / 2.0


@pytest.fixture()
def dev():
    import loraine_jl_amd
    d = loraine_jl_amd.Device(0)
    yield d
    d.close()


def _iterate(name, datarank):
    from loraine_jl_amd.model import model_from_sdpa
    g = np.load(os.path.join(GOLD, f"iterate_{name}.npz"))
    model = model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"), datarank=datarank)
    m = int(model.msizes[0])
    X, S = _unpack(g["X_lower_f32"], m), _unpack(g["S_lower_f32"], m)
    Rd = model.C[0].toarray() - S - _mat(model.AA[0].T @ g["y"], m)
    return g, model, m, X, S, Rd


def test_c2_maxG11_rank_one_hot_path_on_golden_iterate(dev):
    g, model, m, X, S, Rd = _iterate("maxG11", -1)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, B=model.B)
    info, out = dev.prepare_w(0, X, S)                                   # src/prepare_W.jl:28-94
    assert info == 0
    V = g["probes"]
    assert relerr(np.diag(out["W"]), g["W_diag"]) < 1e-10 and relerr(out["W"] @ V, g["W_probe"]) < 1e-10
    assert np.allclose(np.sort(out["D"]), g["D_sorted"], rtol=1e-9, atol=0)
    for mode in (-1, 0):                                                 # makeBBBB_rank1 (:1-20) and makeBBBBs on the same data
        H = dev.schur_assemble(mode, want_H=True)
        assert relerr(np.diag(H), g["H_diag"]) < 1e-10
        assert relerr(H @ V, g["H_probe"]) < 1e-10
        assert abs(np.linalg.norm(H) - float(g["H_fro"])) < 1e-10 * float(g["H_fro"])
        i, j = g["H_sample_i"], g["H_sample_j"]
        assert np.max(np.abs(H[i, j] - g["H_sample"])) < 1e-10 * np.max(np.abs(g["H_sample"]))
    h = dev.make_rhs(g["Rp"], [Rd + S])                                  # src/makeBBBB.jl:221-228
    assert relerr(h, g["h"]) < 1e-10
    assert dev.schur_factor() == 0
    assert relerr(dev.schur_solve(g["h"]), g["dely"]) < 1e-8             # cond(H) ~ 1e4 at this iterate


def _compare_trace(o, name, obj_rel):
    tr = json.load(open(os.path.join(GOLD, f"trace_{name}.json")))
    assert o.solver.status == tr["status"] and o.solver.iter == tr["iterations"]
    assert o.objective_value() == pytest.approx(tr["objective"], rel=obj_rel)
    assert o.dual_objective_value() == pytest.approx(tr["dual_objective"], rel=obj_rel)
    return tr


def test_c2_maxG11_whole_solve_follows_the_oracle_trace():
    """every iteration of the C2 solve: objectives within 1e-8 relative of the CPU path (north star), DIMACS 1e-4"""
    from loraine_jl_amd.optimizer import Optimizer
    o = Optimizer(resident=True)
    o.set_silent(True)
    o.set_attribute("kit", 0)
    o.set_attribute("datarank", -1)
    o.read_from_file(os.path.join(GOLD, "maxG11.dat-s"))
    o.optimize()
    tr = _compare_trace(o, "maxG11", 1e-8)
    for k, t in enumerate(o.solver.trace):
        assert t["primal_obj"] == pytest.approx(tr["primal"][k], rel=1e-8, abs=1e-9)
        assert t["dual_obj"] == pytest.approx(tr["dual"][k], rel=1e-8, abs=1e-9)
        assert t["dimacs"] == pytest.approx(tr["dimacs"][k], rel=1e-4, abs=1e-10)


@pytest.mark.parametrize("prec_eig", [1, 0])
def test_c3_thetaG11_operator_preconditioner_cg_on_golden_iterate(dev, prec_eig):
    """MyA(x), MyM(x) (H_alpha, erank 1) and cg on a real thetaG11 iterate.  prec_eig=1: eigen(W) by the Jacobi
    eigendecomposition (what the reference's `eigen` returns, to rounding); 0: the default, Lanczos extremes -- the
    preconditioner then differs from the reference's by the Lanczos residual, the operator not at all."""
    g, model, m, X, S, Rd = _iterate("thetaG11", 0)
    dev.set_option("prec_eig", prec_eig)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, out = dev.prepare_w(0, X, S)
    assert info == 0
    V = g["probes"]
    assert relerr(np.diag(out["W"]), g["W_diag"]) < 1e-10 and relerr(out["W"] @ V, g["W_probe"]) < 1e-10
    assert relerr(dev.matvec(g["x"]), g["MyA_x"]) < 1e-11               # src/Solvers.jl:582-614
    assert dev.prec_setup(1, 1, 1) == 0                                  # Prec_for_CG_tilS_prep, erank 1, aamat 1
    assert relerr(dev.prec_apply(g["x"]), g["MyM_x"]) < 1e-10           # :866-904  (measured 5e-13 / 1.4e-13)
    h = dev.make_rhs(g["Rp"], [Rd + S])
    assert relerr(h, g["h"]) < 1e-10
    for tol, xs, ec, it in zip(g["cg_tols"], g["cg_x"], g["cg_exit"], g["cg_iters"]):
        x, exit_code, iters = dev.pcg(g["h"], float(tol))               # src/predictor_corrector.jl:134
        assert exit_code == int(ec)
        assert iters == int(it)                                         # same iteration count at every tolerance
        # same number of steps of the same recurrence: the iterates differ by rounding only (measured 3e-8 at
        # tol 1e-6, 5e-14 at tol 1e-10)
        assert relerr(x, xs) < 0.1 * float(tol) + 1e-11


def test_c3_thetaG11_whole_solve_against_the_oracle_trace():
    """The C3 solve (PCG with H_alpha, erank 1, tolerance 1e-2 halved per iteration).  A TRUNCATED CG solve is not a
    contraction: at these tolerances a perturbation of the iterate grows by two orders of magnitude per IP iteration
    (tools/c3_divergence.py: 3e-12 -> 2e-10 -> 1e-8 -> 1e-4 in dely over the first three iterations; the oracle run
    against ITSELF from an initial point perturbed by 1e-13 behaves the same, profiles/r02_c3_sensitivity.txt).  So
    what two correct implementations share is: the hot-path results on identical inputs (test above, 1e-10..1e-13),
    the first iterations of the trajectory to the north-star tolerance, the iteration count, and the optimum to the
    termination tolerance eDIMACS = 1e-5."""
    from loraine_jl_amd.optimizer import Optimizer
    o = Optimizer(resident=True)
    o.set_silent(True)
    for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5).items():
        o.set_attribute(k, v)
    o.read_from_file(os.path.join(GOLD, "thetaG11.dat-s"))
    o.optimize()
    tr = json.load(open(os.path.join(GOLD, "trace_thetaG11.json")))
    assert o.solver.status == tr["status"] == 1 and o.solver.iter == tr["iterations"]
    # Before the sensitivity of the truncated solves takes over.  Iteration 0 to the north-star tolerance.  Iteration 1 sits
    # on the edge: the oracle against ITSELF (X0 scaled by 1 + 1e-13, profiles/r02_c3_sensitivity.txt) differs by 5e-10 /
    # 2e-10 there and by 2.6e-4 one iteration later; the step length alpha = r'z / p'Ap of a CG step carries eps * cond(H)
    # of rounding whatever the summation order (round 4 changed the order of the dots: 1.3e-9 / 6.0e-8 measured).
    for k, (tol_p, tol_d) in enumerate([(1e-8, 1e-8), (1e-8, 1e-6)]):
        t = o.solver.trace[k]
        assert (t["cg_pre"], t["cg_cor"]) == (tr["cg_pre"][k], tr["cg_cor"][k])
        assert t["primal_obj"] == pytest.approx(tr["primal"][k], rel=tol_p)
        assert t["dual_obj"] == pytest.approx(tr["dual"][k], rel=tol_d)
    assert o.objective_value() == pytest.approx(tr["objective"], rel=2e-6)
    assert o.objective_value() == pytest.approx(400.0, rel=2e-6)                  # SDPLIB (external)
    assert abs(o.solver.cg_iter_tot - tr["cg_total"]) <= 0.15 * tr["cg_total"]


# (trace_thetaG11_tight7.json, eDIMACS 1e-7 / tol_cg_min 1e-9, is recorded evidence only: 8e-13 relative when it was run with
# tools/c3_tight.py, but that close to the breakdown of H_alpha a CG call can take its 10000 iterations and the solve
# anything between 5 s and 4 minutes depending on rounding -- not a test)
@pytest.mark.parametrize("name,suffix,edimacs", [("thetaG11", "tight", 1e-6), ("tru3", "tight", 1e-6)])
def test_kit1_solves_agree_where_both_sit_on_the_optimum(name, suffix, edimacs):
    """VERDICT r2 item 4(ii).  A truncated-CG trajectory is not reproducible between two correct implementations (DESIGN.md
    section 2), so C3's objectives at its stock tolerance (eDIMACS 1e-5) agree to 3e-7 only.  One decade further in both
    runs sit on the optimum: `oracle/make_golden.py <name>_tight` with TIGHT_EDIMACS=1e-6 TIGHT_TOL_CG_MIN=1e-8 wrote
    tests/golden/trace_<name>_tight.json (thetaG11: C3's configuration, kit=1 H_alpha erank 1; tru3: a truss problem with
    linear rows on the same path); `trace_thetaG11_tight7.json` is the same at 1e-7 / 1e-9 (oracle 400.000000000855, GPU
    400.0000000005).  North star: objectives within 1e-8 relative.  (Two decades further the reference's own
    H_alpha formulas break down -- NaN in prepare_W.jl:71-74 / PosDefException in Solvers.jl:730 -- see tests/golden/README.md.)"""
    import json
    from loraine_jl_amd.optimizer import Optimizer
    ref = json.load(open(os.path.join(GOLD, "trace_%s_%s.json" % (name, suffix))))
    assert ref["status"] == 1 and ref["options"]["eDIMACS"] == edimacs
    o = Optimizer(resident=True)
    o.set_silent(True)
    for k, v in ref["options"].items():
        if k != "verb":
            o.set_attribute(k, v)
    o.read_from_file(os.path.join(GOLD, name + ".dat-s"))
    o.optimize()
    assert o.termination_status() == "OPTIMAL"
    assert o.objective_value() == pytest.approx(ref["objective"], rel=1e-8)
    # the dual objective is only pinned by the termination test: DIMACS err5 = gap / (1 + |p| + |d|) < eDIMACS leaves a
    # relative gap of up to 2 eDIMACS at either end (round 4, other summation order in the CG dots: 1.04e-6 on thetaG11)
    assert o.dual_objective_value() == pytest.approx(ref["dual_objective"], rel=2.5 * edimacs)
    if edimacs == 1e-6:
        assert abs(o.solver.iter - ref["iterations"]) <= 2       # (at 1e-7 / 1e-9: 23 against 19, truncated-CG paths)
