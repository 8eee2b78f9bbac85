"""GPU, BASELINE.json's metric configuration at FULL size (synthetic dense SDP, matrix side 2000, 4000 constraints,
128 GB of constraint data generated on the device): the oracle cannot run here, so the assembly is checked through
size-independent properties -- sampled entries against tr(A_i W A_j W) on the host, exact homogeneity under a
power-of-two rescaling of W, symmetry / positive definiteness, the solve residual, and agreement of the three
assembly formulations (Cholesky path, T_k through the factor, T_k = W A_k W)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MSZ, NVAR, SEED = 2000, 4000, 20250614


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def big():
    import torch
    import loraine_jl_amd
    free, _ = torch.cuda.mem_get_info(0)
    if free < 250e9:
        pytest.skip("needs an otherwise empty 288 GB device")
    d = loraine_jl_amd.Device(0)
    d.synthetic_dense_model(MSZ, NVAR, SEED)
    rng = np.random.default_rng(SEED + 1)
    G = rng.standard_normal((MSZ, MSZ)) / np.sqrt(MSZ) + np.eye(MSZ)
    W = G @ G.T
    d.set_scaling(0, W, G)
    d.reset_timing()
    H = d.schur_assemble(0, want_H=True)
    assert d.count("schur_chol") == 1
    yield d, W, G, H
    d.close()


def test_fullsize_sampled_entries_match_the_definition(big):
    d, W, G, H = big
    idx = [0, 1, 777, 2048, 3999]
    A = [d.get_constraint(0, k) for k in idx]
    T = [W @ a @ W for a in A]
    for p, i in enumerate(idx):
        for q, j in enumerate(idx):
            ref = float(np.sum(A[q] * T[p]))                     # tr(A_j W A_i W)
            assert H[j, i] == pytest.approx(ref, rel=1e-11, abs=1e-9 * abs(H[i, i]))


def test_fullsize_symmetric_positive_definite_and_solves(big):
    d, W, G, H = big
    assert np.array_equal(H, H.T)
    assert d.schur_factor() == 0
    h = np.random.default_rng(3).standard_normal(NVAR)
    x = d.schur_solve(h)
    assert relerr(H @ x, h) < 1e-9


def test_fullsize_homogeneity_is_exact_for_a_power_of_two(big):
    d, W, G, H = big
    d.set_scaling(0, 4.0 * W, 2.0 * G)                           # chol(4W) = 2L exactly => H scales by 16 exactly
    try:
        H4 = d.schur_assemble(0, want_H=True)
    finally:
        d.set_scaling(0, W, G)
    assert np.array_equal(H4, 16.0 * H)


def test_fullsize_three_formulations_agree(big):
    d, W, G, H = big
    try:
        for opt, key in ((2, "schur_via_l"), (0, None)):
            d.set_option("schur_chol", opt)
            d.reset_timing()
            Hx = d.schur_assemble(0, want_H=True)
            assert d.count("schur_chol") == 0
            if key:
                assert d.count(key) > 0
            assert relerr(Hx, H) < 1e-13
    finally:
        d.set_option("schur_chol", -1)
