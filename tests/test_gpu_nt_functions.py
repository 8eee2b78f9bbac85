"""GPU parity at FUNCTION level of the NT scaling and the step on the device-resident path, both routes:
`lrn_ip_prepare_w` (nt_mode 1: the eigen-free route of csrc/prepw.hip::prepare_w_ns, nt_mode 0: the reference's SVD
route) against the oracle's `prepare_W` (reference src/prepare_W.jl:28-94), then `lrn_ip_find_step` / `lrn_ip_update` /
`lrn_ip_rhs_corr` against the oracle's `find_step` and corrector right-hand side (src/predictor_corrector.jl:248-326,186)
on ONE iterate with identical inputs (the oracle's dely, step lengths and sigma are fed to the device, so every function
is compared by itself).  Plus the two fallbacks of the eigen-free route, forced: `ns_maxit` (Newton-Schulz gives up ->
SVD route for that iteration) and `lyap_maxit` (the Lyapunov CG gives up in the MIDDLE of an iteration)."""
import copy
import os

import numpy as np
import pytest
import scipy.sparse as sp

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
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def _oracle_step(model, X, S, y, opts):
    """One IP iteration of the oracle from (X, S, y), with the state after every stage."""
    ha = lo.Halpha(int(opts.get("kit", 0)))
    s = lo.MySolver(model, dict(opts, verb=0))
    lo.setup_solver(s, ha)
    lo.initial_point(s)
    s.X, s.S, s.y = [X.copy()], [S.copy()], y.copy()
    s.iter += 1
    s.cg_iter_pre = s.cg_iter_cor = 0
    lo.find_mu(s)
    lo.prepare_W(s)
    assert s.status == 0
    st = dict(mu=s.mu, tau=s.tau, W=s.W[0].copy(), Si=s.Si[0].copy(), G=s.G[0].copy(), D=s.D[0].copy())
    lo.predictor(s, ha)
    st["pred"] = dict(dely=np.ravel(s.dely).copy(), delX=s.delX[0].copy(), delS=s.delS[0].copy(), alpha=float(s.alpha[0]),
                      beta=float(s.beta[0]), Xn=s.Xn[0].copy(), Sn=s.Sn[0].copy(), RNT=s.RNT[0].copy(), Rp=s.Rp.copy(),
                      trXnSn=float(np.sum(s.Xn[0] * s.Sn[0])))
    lo.sigma_update(s)
    st["sigma"] = s.sigma
    lo.corrector(s, ha)
    st["corr"] = dict(h=s.h_corr.copy(), dely=np.ravel(s.dely).copy(), delX=s.delX[0].copy(), delS=s.delS[0].copy(),
                      alpha=float(s.alpha[0]), beta=float(s.beta[0]), X=s.X[0].copy(), S=s.S[0].copy())
    return st


def _device_step(dev, model, X, S, y, st, nt_mode):
    """The same iteration through the C ABI, function by function; returns the relative errors."""
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    dev.set_option("nt_mode", nt_mode)
    dev.reset_timing()
    dev.ip_set_c(0, model.C[0].toarray())
    dev.ip_set_iterate(0, X, S)
    assert dev.ip_prepare_w(0) == 0
    e = {}
    blk = lambda name: dev.dbg_get_block(0, name)[0]
    Wd, flag = dev.dbg_get_block(0, "W")
    assert flag == (1 if nt_mode == 1 else 0)                    # the route that was asked for is the one that ran
    e["W"] = relerr(Wd, st["W"])
    e["Si"] = relerr(blk("Si"), st["Si"])
    e["WSW=X"] = relerr(Wd @ S @ Wd, X)
    p, q = st["pred"], st["corr"]
    dev.ip_residual_d(y)
    a, b = dev.ip_find_step(True, 0.0, st["tau"], p["dely"])
    e["pred.delS"] = relerr(blk("delS"), p["delS"])
    e["pred.delX"] = relerr(blk("delX"), p["delX"])
    e["pred.alpha"] = abs(a[0] - p["alpha"]) / abs(p["alpha"])
    e["pred.beta"] = abs(b[0] - p["beta"]) / abs(p["beta"])
    tr = dev.ip_update(True, [p["alpha"]], [p["beta"]])
    e["pred.trXnSn"] = abs(tr[0] - p["trXnSn"]) / max(abs(p["trXnSn"]), 1e-300)
    grg = st["G"] @ p["RNT"] @ st["G"].T                          # the form the corrector consumes: G RNT G' (:186, :257)
    if nt_mode == 1:
        got = blk("Qm")
    else:
        Gd = blk("G")
        got = Gd @ blk("RNT") @ Gd.T
    e["G.RNT.G'"] = relerr(got, grg)
    sm = st["sigma"] * st["mu"]
    e["corr.rhs"] = relerr(dev.ip_rhs_corr(sm), q["h"] - p["Rp"])
    a, b = dev.ip_find_step(False, sm, st["tau"], q["dely"])
    e["corr.delS"] = relerr(blk("delS"), q["delS"])
    e["corr.delX"] = relerr(blk("delX"), q["delX"])
    e["corr.alpha"] = abs(a[0] - q["alpha"]) / abs(q["alpha"])
    e["corr.beta"] = abs(b[0] - q["beta"]) / abs(q["beta"])
    dev.ip_update(False, [q["alpha"]], [q["beta"]])
    Xd, Sd = dev.ip_get_iterate(0)
    e["X+"] = relerr(Xd, q["X"])
    e["S+"] = relerr(Sd, q["S"])
    return e


def _check(e, tol_scaling, tol_step, label):
    print(label, {k: "%.1e" % v for k, v in e.items()})
    for k in ("W", "Si", "WSW=X"):
        assert e[k] < tol_scaling, (label, k, e[k])
    for k, v in e.items():
        if k not in ("W", "Si", "WSW=X"):
            assert v < tol_step, (label, k, v)


def _unpack(lower_f32, m):
    M = np.zeros((m, m))
    M[np.tril_indices(m)] = lower_f32.astype(np.float64)
    return M + np.tril(M, -1).T


@pytest.mark.parametrize("nt_mode", [1, 0])
@pytest.mark.parametrize("name,opts,iters", [("theta1", dict(kit=0, initpoint=1), 6),
                                             ("maxG11", dict(kit=0, datarank=-1), None),
                                             ("thetaG11", dict(kit=1, preconditioner=1, erank=1), None)])
def test_one_iteration_function_by_function_on_committed_iterates(dev, name, opts, iters, nt_mode):
    """theta1 after six oracle iterations; the committed golden iterates of maxG11 (C2) and thetaG11 (C3)."""
    model = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"), datarank=int(opts.get("datarank", 0)))
    m = int(model.msizes[0])
    if iters is not None:
        s = lo.MySolver(model, dict(opts, verb=0, maxit=iters))
        lo.solve(s)
        X, S, y = s.X[0], s.S[0], np.ravel(s.y)
    else:
        g = np.load(os.path.join(GOLD, f"iterate_{name}.npz"))
        X, S, y = _unpack(g["X_lower_f32"], m), _unpack(g["S_lower_f32"], m), np.ravel(g["y"])
    st = _oracle_step(model, X, S, y, opts)
    try:
        e = _device_step(dev, model, X, S, y, st, nt_mode)
    finally:
        dev.set_option("nt_mode", 1)
    _check(e, 1e-10, 1e-9, f"{name} nt_mode={nt_mode}")


def _near_central_path(m, cond, seed):
    """X with the given condition number, S = mu X^-1 perturbed (eigenvalues of XS within [0.5, 2] mu, eigenvectors rotated
    by an angle that keeps cond(L_X' S L_X) moderate): what iterates of an IP solve look like."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    lam = np.logspace(0, np.log10(cond), m) / np.sqrt(cond)
    E = rng.standard_normal((m, m))
    R = np.linalg.qr(np.eye(m) + (0.3 / np.sqrt(cond)) * (E - E.T))[0]
    X = (Q * lam) @ Q.T
    Q2 = Q @ R
    S = (Q2 * (rng.uniform(0.5, 2.0, m) / lam)) @ Q2.T
    return (X + X.T) / 2, (S + S.T) / 2


def _random_model(m, nvar, seed):
    rng = np.random.default_rng(seed)
    A = [sp.csc_matrix((m, m))]
    for _ in range(nvar):
        M = sp.random(m, m, density=0.08, random_state=rng, data_rvs=rng.standard_normal).toarray()
        A.append(sp.csc_matrix((M + M.T) / 2))
    C0 = rng.standard_normal((m, m))
    A[0] = sp.csc_matrix(-(C0 + C0.T) / 2)
    return lo.make_model([A], rng.standard_normal(nvar), 0.0, None, None)


@pytest.mark.parametrize("nt_mode", [1, 0, 2])
@pytest.mark.parametrize("cond", [1e6, 1e9, 1e12])
def test_one_iteration_function_by_function_on_ill_conditioned_iterates(dev, cond, nt_mode):
    """Synthetic X, S with cond(X) = cond(S) from 1e6 to 1e12.  The oracle's own SVD route loses cond * eps here (both device
    routes sit at the same distance from it): tolerance of the scaling 1e-16 cond (1e-10 at least), of the step 4 x that."""
    m, nvar = 96, 40
    model = _random_model(m, nvar, 3)
    X, S = _near_central_path(m, cond, 11)
    y = 0.1 * np.random.default_rng(5).standard_normal(nvar)
    st = _oracle_step(model, X, S, y, dict(kit=0))
    # nt_mode 2 (test only): the eigen-free route with the Newton-Schulz scale and schedule taken from a Lanczos run on K
    # (round 4; by default from msz 1500 on) at this size
    if nt_mode == 2:
        dev.set_option("ns_lanczos_min", 8)
    try:
        e = _device_step(dev, model, X, S, y, st, 1 if nt_mode == 2 else nt_mode)
        if nt_mode == 2:
            assert dev.count("ns_lanczos_scaled") > 0
    finally:
        dev.set_option("nt_mode", 1)
        dev.set_option("ns_lanczos_min", 1500)
    tol = max(1e-10, 1e-16 * cond)          # measured 2e-11 / 1.3e-8 / 1.6e-5 at cond 1e6 / 1e9 / 1e12, the same on both routes
    _check(e, tol, max(1e-9, 4 * tol), f"cond={cond:g} nt_mode={nt_mode}")


@pytest.mark.parametrize("nt_mode", [1, 0])
def test_one_iteration_function_by_function_at_the_size_of_the_128_tile_kernels(dev, nt_mode):
    """msz 1536: from 1500 on the products with the Cholesky factors (L_X' dS L_X, L_X TX L_X', L_X Z L_X', L_S^-T L_S^-1)
    run on the 128-tile kernel over the triangular K ranges only (csrc/prepw.hip::pgemm_nt, round 4), and P = Z Y of the
    Newton-Schulz iteration as lower tiles + mirror."""
    m, nvar = 1536, 6
    rng = np.random.default_rng(2)
    A = [sp.csc_matrix((m, m))]
    for _ in range(nvar):
        M = sp.random(m, m, density=0.002, random_state=rng, data_rvs=rng.standard_normal)
        A.append(sp.csc_matrix((M + M.T) / 2))
    C0 = sp.random(m, m, density=0.01, random_state=rng, data_rvs=rng.standard_normal)
    A[0] = sp.csc_matrix(-(C0 + C0.T) / 2)
    model = lo.make_model([A], rng.standard_normal(nvar), 0.0, None, None)
    X, S = _near_central_path(m, 1e6, 4)
    y = 0.1 * rng.standard_normal(nvar)
    st = _oracle_step(model, X, S, y, dict(kit=0))
    try:
        e = _device_step(dev, model, X, S, y, st, nt_mode)
    finally:
        dev.set_option("nt_mode", 1)
    _check(e, 2e-10, 2e-9, f"msz 1536 nt_mode={nt_mode}")


def _run(path, device, **opts):
    from loraine_jl_amd.optimizer import Optimizer
    o = Optimizer(resident=True, device=device)
    o.set_silent(True)
    for k, v in opts.items():
        o.set_attribute(k, v)
    o.read_from_file(path)
    o.optimize()
    return o


@pytest.mark.parametrize("name,opts", [("theta1", dict(kit=0, eDIMACS=1e-6, initpoint=1)), ("control1", dict(kit=0, eDIMACS=1e-6))])
@pytest.mark.parametrize("knob,value,counter", [("ns_maxit", 4, "ns_fallback"), ("lyap_maxit", 3, "lyap_fallback")])
def test_forced_fallbacks_keep_the_oracle_trajectory(dev, name, opts, knob, value, counter):
    """`ns_maxit = 4`: Newton-Schulz cannot converge -> the SVD route runs for the iteration (prepw.hip `ns_fallback`).
    `lyap_maxit = 3`: the Lyapunov CG of the second-order term gives up -> the SVD quantities are formed in the middle of
    the iteration, after the predictor's directions came from the other route (ipstep.hip `lyap_fallback`).  Both must
    leave the per-iteration trace within 1e-8 of the oracle's."""
    path = os.path.join(GOLD, f"{name}.dat-s")
    ref = lo.MySolver(lo.model_from_sdpa(path), dict(opts, verb=0))
    lo.solve(ref)
    default = {"ns_maxit": 40, "lyap_maxit": 300}[knob]
    dev.set_option(knob, value)
    dev.reset_timing()
    try:
        o = _run(path, dev, **opts)
        fallbacks = dev.count(counter)
    finally:
        dev.set_option(knob, default)
    assert fallbacks > 0
    assert o.termination_status() == "OPTIMAL"
    assert o.solver.iter == ref.iter
    for tg, tr in zip(o.solver.trace, ref.trace):
        assert tg["primal_obj"] == pytest.approx(tr["primal_obj"], rel=1e-8, abs=1e-9)
        if tr["dimacs"] > 1e-5:
            assert tg["dimacs"] == pytest.approx(tr["dimacs"], rel=1e-4)
    assert o.objective_value() == pytest.approx(lo.objective_value(ref), rel=1e-8, abs=1e-10)


@pytest.mark.parametrize("width", [1e-9, 1e-8, 1e-7])
def test_jacobi_early_stop_with_clustered_singular_values(dev, width):
    """`jacobi_early = 3e-8` ends the SVD after a sweep whose rotated pairs were all that close to orthogonal.
    Inside a cluster of singular values of relative width ~ the threshold every cosine is below it while the rotation
    angles are O(1): the NT identities must still hold (prepare_W.jl:60-74; SVD route, the only one that runs Jacobi)."""
    m = 120
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    # three clusters of 30 nearly equal eigenvalues of X S plus a spread tail
    d = np.concatenate([1.0 + width * rng.uniform(-1, 1, 30), 3.0 + 3.0 * width * rng.uniform(-1, 1, 30),
                        0.2 + 0.2 * width * rng.uniform(-1, 1, 30), np.logspace(-2, 2, 30)])
    lam = np.logspace(-2, 2, m)
    X = (Q * lam) @ Q.T
    S = (Q * (d / lam)) @ Q.T                        # X S = Q diag(d) Q': singular values of L_S' L_X = sqrt(d), clustered
    X, S = (X + X.T) / 2, (S + S.T) / 2
    A = [[sp.csc_matrix((m, m)), sp.identity(m, format="csc")]]
    model = lo.make_model(A, np.ones(1), 0.0, None, None)
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    res = {}
    for early in (3e-8, 0.0):
        dev.set_option("jacobi_early", early)
        try:
            info, out = dev.prepare_w(0, X, S)
        finally:
            dev.set_option("jacobi_early", 3e-8)
        assert info == 0
        W, G, Gi, D = out["W"], out["G"], out["Gi"], out["D"]
        res[early] = (relerr(W @ S @ W, X), relerr(G.T @ S @ G, np.diag(D)), relerr(Gi @ X @ Gi.T, np.diag(D)),
                      relerr(G @ G.T, W), relerr(np.sort(D), np.sort(np.sqrt(d))))
    print("width", width, {k: ["%.1e" % v for v in r] for k, r in res.items()})
    for r in res.values():
        assert max(r) < 1e-10


@pytest.mark.parametrize("name,opts", [("theta1", dict(kit=0, eDIMACS=1e-6, initpoint=1)), ("control1", dict(kit=0, eDIMACS=1e-6)),
                                       ("maxG11", dict(kit=0, datarank=-1))])
def test_both_forms_of_the_lyapunov_equation_give_the_same_solve(dev, name, opts):
    """Round 4: the second-order term of the corrector from  (Yh/s + s Zh) R + R (Yh/s + s Zh) = C/s + s Zh C Zh  (default)
    instead of  Yh R + R Yh = C  (`lyap_form = 0`): the same R, so the same trajectory (objectives of every iteration to
    1e-9), in a fraction of the CG steps."""
    path = os.path.join(GOLD, f"{name}.dat-s")
    runs = {}
    for form in (0, 1):
        dev.set_option("lyap_form", form)
        try:
            o = _run(path, dev, **opts)
        finally:
            dev.set_option("lyap_form", 1)
        assert o.termination_status() == "OPTIMAL"
        runs[form] = (o.solver.iter, [t["primal_obj"] for t in o.solver.trace], sum(t["lyap_steps"] for t in o.solver.trace))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert a == pytest.approx(b, rel=1e-9, abs=1e-10)
    print(name, "Lyapunov CG steps per solve:", runs[0][2], "->", runs[1][2])
    assert runs[1][2] < runs[0][2]
