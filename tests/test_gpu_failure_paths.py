"""Failure paths of the direct solve (src/predictor_corrector.jl:55-90): a Schur matrix that is not numerically
positive definite.  The reference's `cholesky(BBBB)` throws, `regcount` goes up, `+1e-4*I` is added until the
factorisation passes, and -- because :85 stores the Cholesky object where :57-58 keeps the factor -- the two solves of
that iteration are applied twice; after five such iterations the solve stops with status 3.

STRICT mode (lrn_set_option "pivot_boost" = 0) is the literal reference behaviour: the GPU path must then give the
oracle's status and regularisation count.  The DEFAULT boosts pivots at rounding level (<= 1e-12 of their diagonal
entry) instead of failing and reports how many (`chol_boosted`): whether `cholesky` of a numerically singular PSD matrix
throws is decided by rounding noise -- LAPACK happens to get through tru9's last iterations (lambda_min(H) = -1e-3 at
|H| = 4e12), a strict GPU factorisation happens not to, and with the reference's double solve of regularised
iterations that difference ends the solve with status 3 where the CPU path reports OPTIMAL.  With the default, every
problem the reference solves is solved to the same iterates; the divergence is confined to matrices that are singular
beyond rounding, where the default carries on to an optimum and the reference gives up -- pinned here so that it
stays a documented choice."""
import os

import numpy as np
import pytest

from oracle import loraine_oracle as lo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _problem(seed, monkeypatch):
    """the generator of tools/fuzz_parity.py with FUZZ_HUGE (LMI blocks of side 100-160); seeds 604, 605 and 608 have
    variables that occur only in a linear row: H is singular from the first iteration on"""
    monkeypatch.setenv("FUZZ_HUGE", "1")
    from tools.fuzz_parity import random_problem
    return random_problem(np.random.default_rng(seed))


def _oracle(A, b, d_lin, C_lin):
    om = lo.make_model([[m.copy() for m in blk] for blk in A], b.copy(), 0.0,
                       None if d_lin is None else d_lin.copy(), None if C_lin is None else C_lin.copy())
    ref = lo.MySolver(om, dict(kit=0, verb=0))
    lo.solve(ref)
    return ref


def _gpu(A, b, d_lin, C_lin, resident, boost=None, **kw):
    import loraine_jl_amd
    from loraine_jl_amd.optimizer import Optimizer
    d = loraine_jl_amd.Device(0)
    if boost is not None:
        d.set_option("pivot_boost", boost)
    o = Optimizer(resident=resident, device=d, **kw)
    o.set_silent(True)
    o.set_attribute("kit", 0)
    o.load_model([[m.copy() for m in blk] for blk in A], b.copy(), 0.0, d_lin, C_lin, max_sense=False)
    o.optimize()
    boosted = sum(t.get("chol_boosted", 0) for t in o.solver.trace)
    d.close()
    return o, boosted


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("seed", [604, 605, 608])
def test_singular_schur_matrix_strict_mode_follows_the_reference(seed, resident, monkeypatch):
    A, b, d_lin, C_lin = _problem(seed, monkeypatch)
    ref = _oracle(A, b, d_lin, C_lin)
    assert ref.status == 3 and ref.regcount == 6                  # gives up: "too many regularizations of H" (:65-71)
    o, boosted = _gpu(A, b, d_lin, C_lin, resident, boost=0.0)
    assert boosted == 0
    assert o.solver.status == ref.status                          # both give up ...
    assert o.solver.regcount == ref.regcount                      # ... after the sixth failed factorisation
    # Whether `cholesky` of a numerically SINGULAR matrix throws in a given iteration is decided by rounding noise
    # (the oracle itself stops after 10 iterations on one host and 8 on another for seed 604), so the iteration
    # counts may differ by the lucky factorisations; the iterations before the first failure on either side walk the
    # same trajectory, and every regularised iteration needs exactly one +1e-4*I.
    assert abs(o.solver.iter - ref.iter) <= 4
    first = min(next((i for i, t in enumerate(tr) if t["regcount"] > 0), len(tr)) for tr in (o.solver.trace, ref.trace))
    for tg, tr in zip(o.solver.trace[:first], ref.trace[:first]):
        assert tg["primal_obj"] == pytest.approx(tr["primal_obj"], rel=1e-6, abs=1e-8)
    assert all(t["reg_adds"] <= 1 for t in o.solver.trace) and all(t["reg_adds"] <= 1 for t in ref.trace)


@pytest.mark.parametrize("seed", [604, 605, 608])
def test_default_pivot_boost_solves_what_the_reference_gives_up_on(seed, monkeypatch):
    A, b, d_lin, C_lin = _problem(seed, monkeypatch)
    o, boosted = _gpu(A, b, d_lin, C_lin, True)
    assert boosted > 0                                            # pivots at rounding level replaced, no failure reported
    assert o.solver.regcount == 0 and o.solver.status == 1        # ... and the solve reaches an optimum the reference never sees


def test_exact_regularised_solve_is_an_opt_in_divergence(monkeypatch):
    """(H + d I)^-1 h instead of the reference's H_reg^-1 (H_reg^-1 h): another trajectory from the first regularised
    iteration on (not asserted to be better -- only that the switch exists and the default is the reference's)."""
    A, b, d_lin, C_lin = _problem(604, monkeypatch)
    ref = _oracle(A, b, d_lin, C_lin)
    o, _ = _gpu(A, b, d_lin, C_lin, True, boost=0.0, exact_regularised_solve=True)
    assert o.solver.exact_regularised_solve and not o.solver.chol_is_object
    k = next(i for i, t in enumerate(ref.trace) if t["regcount"] > 0)
    assert len(o.solver.trace) > k
    assert abs(o.solver.trace[k]["primal_obj"] - ref.trace[k]["primal_obj"]) > 1e-6 * abs(ref.trace[k]["primal_obj"])


def test_tru9_default_matches_the_oracle():
    """tru9 late in the solve: lambda_min(H) sinks to rounding level (-1e-3 at |H| = 4e12).  The default must go the
    oracle's way: 28 iterations, no regularisation (the oracle's run -- status 1, 28 iterations, regcount 0 -- is
    recorded in tests/golden/README.md; 220 s on 8 cores, too slow to repeat here)."""
    from loraine_jl_amd.optimizer import Optimizer
    o = Optimizer(resident=True)
    o.set_silent(True)
    o.set_attribute("kit", 0)
    o.read_from_file(os.path.join(GOLD, "tru9.dat-s"))
    o.optimize()
    assert o.solver.status == 1 and abs(o.solver.iter - 28) <= 1
    assert o.solver.regcount == 0
    assert o.objective_value() == pytest.approx(0.0597530923, rel=2e-7)


@pytest.mark.parametrize("resident", [True, False])
def test_unbounded_rank_one_problem_ends_in_a_reference_status(resident, monkeypatch):
    """Fuzz seed 3024 of the rank-one generator is unbounded: the objective runs to -1e18 and the iterates lose their
    finite entries.  The reference then leaves through try_cholesky's give-up (status 4, prepare_W.jl:17-21) or the
    iteration limit -- never through an exception of the host language (round 2: the host driver raised ValueError
    from SciPy's eigvalsh on a NaN matrix)."""
    import tools.fuzz_parity as F
    monkeypatch.setattr(F, "RANK1", True)
    A, b, d_lin, C_lin = F.random_problem(np.random.default_rng(3024))
    import loraine_jl_amd
    from loraine_jl_amd.optimizer import Optimizer
    d = loraine_jl_amd.Device(0)
    o = Optimizer(resident=resident, device=d)
    o.set_silent(True)
    o.set_attribute("kit", 0)
    o.set_attribute("datarank", -1)
    o.load_model([[m.copy() for m in blk] for blk in A], b.copy(), 0.0, d_lin, C_lin, max_sense=False)
    o.optimize()
    d.close()
    assert o.solver.status in (3, 4)
    om = lo.make_model([[m.copy() for m in blk] for blk in A], b.copy(), 0.0, d_lin, C_lin, datarank=-1)
    ref = lo.MySolver(om, dict(kit=0, datarank=-1, verb=0))
    lo.solve(ref)
    assert ref.status in (3, 4)
