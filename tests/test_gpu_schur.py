"""GPU parity: Schur-complement assembly / factor / solve through the C ABI against the
CPU oracle on the same inputs (reference src/makeBBBB.jl, predictor_corrector.jl:53-90)."""
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
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _spd(m, seed, cond=1e3):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    lam = np.logspace(0, np.log10(cond), m)
    G = Q * np.sqrt(lam)[None, :]
    return G @ G.T, G


def _herm_lower(H):
    return np.tril(H) + np.tril(H, -1).T


def _brute_H(Amat, W):
    """H_ij = <A_i, W A_j W> by plain matmuls (full-tensor independent reference)."""
    T = np.stack([W @ a @ W for a in Amat])
    n = Amat.shape[0]
    return Amat.reshape(n, -1) @ T.reshape(n, -1).T


def _upload(dev, model):
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes,
                     B=model.B if len(model.B) else None, C_lin=model.C_lin if model.nlin else None)


@pytest.mark.parametrize("name,kappa", [("theta1", 8), ("theta1", 0), ("theta1", 1000), ("control1", 8),
                                        ("tru3", 8), ("vib3", 8), ("maxG11", 8)])
def test_assemble_matches_oracle(dev, name, kappa):
    model = lo.model_from_sdpa(os.path.join(GOLD, f"{name}.dat-s"), kappa=kappa)
    Ws, Gs = [], []
    for i, m in enumerate(model.msizes):
        W, G = _spd(int(m), 10 + i)
        Ws.append(W); Gs.append(G)
    _upload(dev, model)
    for i in range(model.nlmi):
        dev.set_scaling(i, Ws[i], Gs[i])
    rng = np.random.default_rng(0)
    Href = lo.makeBBBBs(model.n, model.nlmi, model.A, model.AA, Ws, model.qA, model.sigmaA)
    if model.nlin:
        xl = rng.uniform(0.5, 2.0, model.nlin); sl = rng.uniform(0.5, 2.0, model.nlin)
        dev.set_lin(xl, 1.0 / sl)
        Href = Href + (model.C_lin @ sp.diags(xl / sl) @ model.C_lin.T).toarray()
    Href = _herm_lower(Href)
    H = dev.schur_assemble(0, want_H=True)
    assert relerr(H, Href) < 1e-13
    assert np.array_equal(H, H.T)
    assert dev.schur_factor() == 0
    h = rng.standard_normal(model.n)
    x = dev.schur_solve(h)
    assert relerr(Href @ x, h) < 1e-8


def test_rank1_matches_oracle_and_general(dev):
    model = lo.model_from_sdpa(os.path.join(GOLD, "maxG11.dat-s"), datarank=-1)
    W, G = _spd(800, 3)
    _upload(dev, model)
    dev.set_scaling(0, W, G)
    H1 = dev.schur_assemble(-1, want_H=True)
    Href = _herm_lower(lo.makeBBBB_rank1(model.n, model.nlmi, model.B, [G]))
    assert relerr(H1, Href) < 1e-12
    H0 = dev.schur_assemble(0, want_H=True)          # general path on the same rank-one data
    assert relerr(H0, Href) < 1e-12


def _dense_model(msz, nvar, seed, density=1.0):
    rng = np.random.default_rng(seed)
    A = [[sp.csc_matrix((msz, msz))]]
    for k in range(nvar):
        R = rng.standard_normal((msz, msz))
        if density < 1.0:
            R = R * (rng.random((msz, msz)) < density)
        A[0].append(sp.csc_matrix((R + R.T) / 2))
    return lo.make_model(A, rng.standard_normal(nvar), 0.0, None, None)


@pytest.mark.parametrize("chol", [0, 1, 2])
@pytest.mark.parametrize("msz,nvar", [(96, 40), (300, 130), (257, 300), (16, 5), (130, 20), (333, 37)])
def test_dense_mfma_path_matches_oracle(dev, msz, nvar, chol):
    """The C4-shaped path: every constraint dense -> GEMM1/GEMM2/GEMM3; chol=0: T_k = W A_k W,
    chol=1: the Cholesky path (L' A_k L on triangular K ranges, packed lower tiles, weighted slabs),
    chol=2: T_k = L (L' A_k L) L' (four triangular products, mirrored store) feeding the W path's GEMM3."""
    model = _dense_model(msz, nvar, msz + nvar)
    W, G = _spd(msz, 5)
    dev.set_option("dense_threshold", 1)             # force the MFMA path regardless of the cost model
    dev.set_option("schur_chol", chol)
    try:
        _upload(dev, model)
        dev.set_scaling(0, W, G)
        dev.reset_timing()
        H = dev.schur_assemble(0, want_H=True)
        assert dev.count("schur_chol") == (1 if chol == 1 else 0)
        assert (dev.count("schur_via_l") > 0) == (chol == 2)
    finally:
        dev.set_option("dense_threshold", -1)
        dev.set_option("schur_chol", -1)
    Amat = np.stack([model.A[0][k + 1].toarray() for k in range(nvar)])
    Href = _brute_H(Amat, W)
    assert relerr(H, Href) < 1e-13
    assert dev.schur_factor() == 0
    h = np.random.default_rng(1).standard_normal(nvar)
    x = dev.schur_solve(h)
    assert relerr(Href @ x, h) < 1e-9


def test_mixed_dense_sparse_owners(dev):
    """A few dense constraints + many sparse ones in one block: dense x dense (GEMM3),
    dense x sparse (gather) and sparse x sparse (pair kernels) meet in one matrix."""
    msz, nd, ns = 200, 9, 60
    rng = np.random.default_rng(11)
    A = [[sp.csc_matrix((msz, msz))]]
    for k in range(nd):
        R = rng.standard_normal((msz, msz)); A[0].append(sp.csc_matrix((R + R.T) / 2))
    for k in range(ns):
        nn = int(rng.integers(1, 12))
        r = rng.integers(0, msz, nn); c = rng.integers(0, msz, nn); v = rng.standard_normal(nn)
        Mx = sp.coo_matrix((v, (r, c)), shape=(msz, msz)).toarray()
        A[0].append(sp.csc_matrix(Mx + Mx.T))
    perm = rng.permutation(nd + ns)
    A[0] = [A[0][0]] + [A[0][1 + p] for p in perm]
    model = lo.make_model(A, rng.standard_normal(nd + ns), 0.0, None, None, kappa=30)
    W, G = _spd(msz, 9)
    dev.set_option("dense_threshold", 1000)
    Hs = []
    try:
        _upload(dev, model)
        dev.set_scaling(0, W, G)
        for chol in (0, 2):                          # T_k = W A_k W, and T_k = L (L' A_k L) L'
            dev.set_option("schur_chol", chol)
            dev.reset_timing()
            Hs.append(dev.schur_assemble(0, want_H=True))
            assert dev.count("schur_chol") == 0 and (dev.count("schur_via_l") > 0) == (chol == 2)
    finally:
        dev.set_option("dense_threshold", -1)
        dev.set_option("schur_chol", -1)
    Amat = np.stack([model.A[0][k + 1].toarray() for k in range(nd + ns)])
    Href = _brute_H(Amat, W)
    H = Hs[0]
    assert relerr(H, Href) < 1e-13
    assert relerr(Hs[1], Href) < 1e-13
    Horacle = _herm_lower(lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA))
    assert relerr(Horacle, Href) < 1e-13


def test_synthetic_dense_generator_and_assembly(dev):
    msz, nvar = 130, 48
    dev.synthetic_dense_model(msz, nvar, 20250614)
    A = np.stack([dev.get_constraint(0, k) for k in range(nvar)])
    assert np.array_equal(A, A.transpose(0, 2, 1))
    assert abs(A.mean()) < 0.05 and 0.6 < A.std() < 0.9          # var 1 on diag, 1/2 off-diag
    dev2_A = dev.get_constraint(0, 7)
    assert np.array_equal(dev2_A, A[7])                           # deterministic
    W, G = _spd(msz, 2)
    dev.set_scaling(0, W, G)
    H = dev.schur_assemble(0, want_H=True)
    Href = _brute_H(A, W)
    assert relerr(H, Href) < 1e-13


def test_shard_export_import_roundtrip(dev):
    """world=2 column ownership on one GPU: two shard assemblies glued by the exchange
    buffers equal the unsharded matrix (the RCCL all-gather is replaced by a concat)."""
    import torch
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    W, G = _spd(50, 4)
    _upload(dev, model)
    dev.set_scaling(0, W, G)
    Hfull = dev.schur_assemble(0, want_H=True)
    dev.set_option("shard_bs", 16)
    parts = []
    for r in range(2):
        dev.set_shard(r, 2)
        dev.schur_assemble(0)
        buf = torch.zeros(dev.shard_doubles(), dtype=torch.float64, device="cuda")
        dev.schur_export_shard(buf)
        parts.append(buf)
    allbuf = torch.cat(parts)
    dev.schur_import_all(allbuf)              # geometry of the 2-rank exchange
    H2 = dev.schur_get()
    dev.set_shard(0, 1)
    dev.set_option("shard_bs", 0)             # back to auto
    # the C library's exchange layout is the one loraine.jl_amd/sharding.py specifies
    from loraine_jl_amd import sharding
    sig = model.sigmaA[:, 0]
    Hpos = np.tril(Hfull[np.ix_(sig, sig)])
    for r in range(2):
        mine = np.zeros_like(Hpos)
        cols = sharding.owned_columns(model.n, r, 2, 16)
        mine[:, cols] = Hpos[:, cols]
        assert relerr(np.tril(sharding.unpack_all(np.concatenate([p.cpu().numpy() for p in parts]), model.n, 2, 16))[:, cols],
                      mine[:, cols]) < 1e-15
    assert relerr(H2, Hfull) < 1e-15


def test_rank1_sharded_columns(dev):
    """datarank=-1 with world=3 column ownership: the union of the shards is the full matrix."""
    import torch
    model = lo.model_from_sdpa(os.path.join(GOLD, "maxG11.dat-s"), datarank=-1)
    W, G = _spd(800, 3)
    _upload(dev, model)
    dev.set_scaling(0, W, G)
    Hfull = dev.schur_assemble(-1, want_H=True)
    parts = []
    for r in range(3):
        dev.set_shard(r, 3)
        from loraine_jl_amd import sharding as _sh
        assert dev.shard_bs() == _sh.auto_bs(model.n, 3)       # the C rule and its Python specification agree
        dev.schur_assemble(-1)
        buf = torch.zeros(dev.shard_doubles(), dtype=torch.float64, device="cuda")
        dev.schur_export_shard(buf)
        parts.append(buf)
    dev.schur_import_all(torch.cat(parts))
    H2 = dev.schur_get()
    dev.set_shard(0, 1)
    assert relerr(H2, Hfull) < 1e-15


def test_against_committed_golden_iterate(dev):
    """C-ABI results on the committed golden iterate of theta1 (oracle/make_golden.py): NT scaling,
    Schur matrix, right-hand side and the Cholesky solve, without running the oracle."""
    g = np.load(os.path.join(GOLD, "iterates_theta1.npz"))
    import loraine_jl_amd  # noqa: F401
    from loraine_jl_amd.model import model_from_sdpa
    model = model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    info, out = dev.prepare_w(0, g["X"], g["S"])
    assert info == 0
    assert relerr(out["W"], g["W"]) < 1e-11
    assert np.allclose(np.sort(out["D"]), g["D"], rtol=1e-10)
    H = dev.schur_assemble(0, want_H=True)
    assert relerr(np.tril(H), g["H_lower"]) < 1e-10
    h = dev.make_rhs(g["Rp"], [g["Rd"] + g["S"]])
    assert relerr(h, g["h"]) < 1e-10
    assert dev.schur_factor() == 0
    assert relerr(dev.schur_solve(g["h"]), g["dely"]) < 1e-8


def test_two_dense_blocks_second_larger(dev):
    """Regression (found by tools/fuzz_parity.py): the MFMA-path workspaces were sized for the first LMI
    block; a larger second block with dense constraints wrote past them."""
    import scipy.sparse as sp
    rng = np.random.default_rng(42)
    nvar, sizes = 30, [17, 29]
    A = []
    for m in sizes:
        blk = [sp.csc_matrix((m, m))]
        for k in range(nvar):
            R = rng.standard_normal((m, m)) * (rng.random((m, m)) < (1.0 if k % 3 else 0.3))
            blk.append(sp.csc_matrix(R + R.T))
        A.append(blk)
    model = lo.make_model(A, np.ones(nvar), 0.0, None, None)
    dev.set_option("dense_threshold", 50)         # every constraint with >= 50 entries takes the MFMA path
    try:
        dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    finally:
        dev.set_option("dense_threshold", -1)
    W = []
    for i, m in enumerate(sizes):
        w, g = _spd(m, 10 + i)
        W.append(w)
        dev.set_scaling(i, w, g)
    H = dev.schur_assemble(0, want_H=True)
    Href = lo.makeBBBBs(model.n, model.nlmi, model.A, model.AA, W, model.qA, model.sigmaA)
    Href = np.tril(Href) + np.tril(Href, -1).T
    assert relerr(H, Href) < 1e-13


def test_chol_path_equals_w_path_midsize(dev):
    """Cholesky path vs the T_k = W A_k W path on a problem large enough for the direct-to-LDS kernels
    (msz 1000: 8 tiles a side, K ranges 1000 ... 104), W with condition 1e6, auto selection."""
    msz, nvar = 1000, 200
    dev.synthetic_dense_model(msz, nvar, 7)
    W, G = _spd(msz, 8, cond=1e6)
    dev.set_scaling(0, W, G)
    dev.reset_timing()
    H1 = dev.schur_assemble(0, want_H=True)
    assert dev.count("schur_chol") == 1              # auto: all constraints dense, one rank
    try:
        dev.set_option("schur_chol", 0)
        dev.reset_timing()
        H0 = dev.schur_assemble(0, want_H=True)
        assert dev.count("schur_chol") == 0 and dev.count("schur_via_l") == 0
        dev.set_option("schur_chol", 2)
        dev.reset_timing()
        H2 = dev.schur_assemble(0, want_H=True)
        assert dev.count("schur_chol") == 0 and dev.count("schur_via_l") > 0
    finally:
        dev.set_option("schur_chol", -1)
    assert relerr(H1, H0) < 1e-13
    assert relerr(H2, H0) < 1e-13
    assert np.array_equal(H1, H1.T)
    A = np.stack([dev.get_constraint(0, k) for k in range(0, nvar, 50)])
    Href = _brute_H(A, W)
    assert relerr(H1[::50, ::50], Href) < 1e-13


def test_chol_path_falls_back_when_w_is_singular(dev):
    msz, nvar = 150, 12
    model = _dense_model(msz, nvar, 77)
    rng = np.random.default_rng(5)
    G = rng.standard_normal((msz, msz - 20))         # 20 eigenvalues at -1e-6: the Cholesky of W breaks down
    W = G @ G.T - 1e-6 * np.eye(msz)
    dev.set_option("dense_threshold", 1)
    dev.set_option("schur_chol", 1)
    try:
        _upload(dev, model)
        dev.set_scaling(0, W, np.zeros((msz, msz)))
        dev.reset_timing()
        H = dev.schur_assemble(0, want_H=True)
        assert dev.count("wchol_fail") == 1 and dev.count("schur_chol") == 0 and dev.count("schur_via_l") == 0
    finally:
        dev.set_option("dense_threshold", -1)
        dev.set_option("schur_chol", -1)
    Amat = np.stack([model.A[0][k + 1].toarray() for k in range(nvar)])
    assert relerr(H, _brute_H(Amat, W)) < 1e-12


def test_chol_path_two_blocks_and_switching(dev):
    """Two dense LMI blocks (sides 40 and 150) through the Cholesky path, then the same context switches to
    the W path and back: the shared T workspace changes layout between assemblies."""
    rng = np.random.default_rng(43)
    nvar, sizes = 20, [40, 150]
    A = []
    for m in sizes:
        blk = [sp.csc_matrix((m, m))]
        for k in range(nvar):
            R = rng.standard_normal((m, m))
            blk.append(sp.csc_matrix(R + R.T))
        A.append(blk)
    model = lo.make_model(A, np.ones(nvar), 0.0, None, None)
    dev.set_option("dense_threshold", 1)
    try:
        dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
    finally:
        dev.set_option("dense_threshold", -1)
    W = []
    for i, m in enumerate(sizes):
        w, g = _spd(m, 20 + i)
        W.append(w)
        dev.set_scaling(i, w, g)
    Href = lo.makeBBBBs(model.n, model.nlmi, model.A, model.AA, W, model.qA, model.sigmaA)
    Href = np.tril(Href) + np.tril(Href, -1).T
    try:
        for chol in (1, 0, 2, 1, 1):
            dev.set_option("schur_chol", chol)
            dev.reset_timing()
            H = dev.schur_assemble(0, want_H=True)
            assert dev.count("schur_chol") == (2 if chol == 1 else 0)
            assert relerr(H, Href) < 1e-13
    finally:
        dev.set_option("schur_chol", -1)


@pytest.mark.parametrize("msz,nvar,world", [(260, 300, 2), (260, 300, 3), (260, 300, 5), (1000, 160, 3), (1000, 160, 8)])
def test_chol_path_column_split_partial_sums(dev, msz, nvar, world):
    """world > 1 on the Cholesky path: the ranks split the columns of the matrix variable in 16-column units (all
    three GEMMs shard, tile grids anchored at the range start) and hold partial sums of the whole Schur matrix;
    their sum (the all-reduce) is the one-rank matrix.  Also more ranks than units (idle ranks contribute zero), and
    the dealing equals its Python specification."""
    import torch
    from loraine_jl_amd import sharding
    dev.synthetic_dense_model(msz, nvar, 11)
    W, G = _spd(msz, 12, cond=1e4)
    dev.set_scaling(0, W, G)
    Hfull = dev.schur_assemble(0, want_H=True)
    assert not dev.schur_is_partial_sum()
    total = torch.zeros(nvar * nvar, dtype=torch.float64, device="cuda")
    buf = torch.zeros(nvar * nvar, dtype=torch.float64, device="cuda")
    shares = 0.0
    dev.set_option("schur_chol", 1)             # (auto keeps the Schur column blocks when ranks outnumber the tiles)
    try:
        for r in range(world):
            dev.set_shard(r, world)
            dev.reset_timing()
            dev.schur_assemble(0)
            assert dev.count("schur_chol") == 1 and dev.schur_is_partial_sum()
            cols = sharding.column_range(msz, nvar, r, world)
            assert (dev.timing("gemm3_share") > 0) == (cols is not None)
            shares += dev.timing("gemm3_share")
            dev.schur_export_full(buf)
            torch.cuda.synchronize()
            total += buf
        torch.cuda.synchronize()
        dev.schur_import_full(total)
        assert not dev.schur_is_partial_sum()
        H2 = dev.schur_get()
    finally:
        dev.set_shard(0, 1)
        dev.set_option("schur_chol", -1)
    assert shares == pytest.approx(1.0, abs=1e-12)       # the ranks' K ranges tile the packed index exactly
    assert relerr(H2, Hfull) < 1e-13
    assert dev.schur_factor() == 0


def test_via_l_path_sharded_three_ranks(dev):
    """world = 3 with the T_k form forced: the ranks run the W path with T_k = L (L' A_k L) L' for their own Schur
    columns; the shards glued by the exchange layout equal the one-rank matrix (which took the Cholesky path)."""
    import torch
    msz, nvar = 300, 420
    dev.synthetic_dense_model(msz, nvar, 13)
    W, G = _spd(msz, 14, cond=1e5)
    dev.set_scaling(0, W, G)
    dev.reset_timing()
    Hfull = dev.schur_assemble(0, want_H=True)
    assert dev.count("schur_chol") == 1
    parts = []
    dev.set_option("schur_chol", 2)              # (auto would take the column split of the Cholesky path)
    try:
        for r in range(3):
            dev.set_shard(r, 3)
            dev.reset_timing()
            dev.schur_assemble(0)
            assert dev.count("schur_chol") == 0 and dev.count("schur_via_l") > 0 and not dev.schur_is_partial_sum()
            buf = torch.zeros(dev.shard_doubles(), dtype=torch.float64, device="cuda")
            dev.schur_export_shard(buf)
            parts.append(buf)
        dev.schur_import_all(torch.cat(parts))
        H2 = dev.schur_get()
    finally:
        dev.set_shard(0, 1)
        dev.set_option("schur_chol", -1)
    assert relerr(H2, Hfull) < 1e-13
    A = np.stack([dev.get_constraint(0, k) for k in range(0, nvar, 60)])
    assert relerr(H2[::60, ::60], _brute_H(A, W)) < 1e-13


@pytest.mark.parametrize("ksplit,stagger", [(1, 0), (3, 0), (7, 5), (64, 1000)])
def test_chol_path_split_and_stagger_knobs(dev, ksplit, stagger):
    """GEMM3' under non-default split-K factors and K-walk staggers (wrap-around inside a split, more stagger
    than chunks, a single split): same matrix as the default to round-off."""
    msz, nvar = 270, 150
    dev.synthetic_dense_model(msz, nvar, 21)
    W, G = _spd(msz, 22)
    dev.set_scaling(0, W, G)
    dev.reset_timing()
    H0 = dev.schur_assemble(0, want_H=True)
    assert dev.count("schur_chol") == 1
    try:
        dev.set_option("gemm3_ksplit", ksplit)
        dev.set_option("gemm3_stagger", stagger)
        H1 = dev.schur_assemble(0, want_H=True)
    finally:
        dev.set_option("gemm3_ksplit", 0)
        dev.set_option("gemm3_stagger", 0)
    assert relerr(H1, H0) < 1e-14
    A = np.stack([dev.get_constraint(0, k) for k in range(0, nvar, 30)])
    assert relerr(H1[::30, ::30], _brute_H(A, W)) < 1e-13


def test_auto_mode_keeps_column_blocks_when_ranks_outnumber_tiles(dev):
    """msz 260 = 3 column tiles: with 5 ranks the automatic choice is the Schur column-block sharding (T_k through
    the factor), with 2 ranks the column split of the matrix variable."""
    dev.synthetic_dense_model(260, 40, 3)
    W, G = _spd(260, 4)
    dev.set_scaling(0, W, G)
    try:
        dev.set_shard(0, 5)                      # (rank 0 owns the only 128-wide Schur column block of nvar = 40)
        dev.reset_timing()
        dev.schur_assemble(0)
        assert not dev.schur_is_partial_sum() and dev.count("schur_via_l") > 0
        dev.set_shard(1, 2)
        dev.reset_timing()
        dev.schur_assemble(0)
        assert dev.schur_is_partial_sum() and dev.count("schur_chol") == 1
    finally:
        dev.set_shard(0, 1)


@pytest.mark.parametrize("msz,nvar", [(300, 150), (1000, 300), (520, 260)])
def test_block_masks_do_not_change_the_schur_matrix(dev, msz, nvar):
    """Round 2: GEMM1'/2'/3' skip the 16x16 blocks that are zero, beyond the edges or never read (interleaved block
    ownership + per-wave masks, GEMM3' as regular and short launches).  Skipped work contributes exact zeros or unread
    entries: with option gemm_no_skip = 1 (every block of every tile computed, one GEMM3' launch) the lower triangle must
    come out the same -- bit for bit where no summation order changed, 1e-14 otherwise -- and equal the definition."""
    dev.synthetic_dense_model(msz, nvar, 31)
    W, G = _spd(msz, 32, cond=1e5)
    dev.set_scaling(0, W, G)
    dev.set_option("schur_chol", 1)
    try:
        dev.reset_timing()
        H1 = np.tril(dev.schur_assemble(0, want_H=True))
        assert dev.count("schur_chol") == 1
        dev.set_option("gemm_no_skip", 1)
        H0 = np.tril(dev.schur_assemble(0, want_H=True))
    finally:
        dev.set_option("gemm_no_skip", 0)
        dev.set_option("schur_chol", -1)
    assert relerr(H1, H0) < 1e-14
    A = np.stack([dev.get_constraint(0, k) for k in range(min(nvar, 12))])
    Href = _brute_H(A, W)
    assert relerr(H1[:12, :12], np.tril(Href)) < 1e-13


def test_pair_kernel_lane_widths_agree(dev):
    """pair_wave_kernel<16> (four Schur entries per wavefront, the default for short products) and <64> (one): the same
    terms in another association -- equal to 1e-14 -- on theta1 (103 two-entry constraints and one of 50 entries, all
    through the pair path with kappa = 1000)."""
    import loraine_jl_amd  # noqa: F401
    model = lo.model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"), kappa=1000)
    _upload(dev, model)
    W, G = _spd(50, 9)
    dev.set_scaling(0, W, G)
    Hs = {}
    try:
        for lanes in (64, 16, 8, 4):
            dev.set_option("pair_lanes", lanes)
            Hs[lanes] = np.tril(dev.schur_assemble(0, want_H=True))
    finally:
        dev.set_option("pair_lanes", 0)
    Href = lo.makeBBBBs(model.n, 1, model.A, model.AA, [W], model.qA, model.sigmaA)
    for lanes, H in Hs.items():
        assert relerr(H, Hs[64]) < 1e-14
        assert relerr(H, np.tril(Href)) < 1e-13


def test_reference_timer_names_are_aliases(dev):
    """lrn_get_timing answers to the reference's TimerOutputs section names (src/makeBBBB.jl:86-98, :30)."""
    dev.synthetic_dense_model(300, 64, 5)
    W, G = _spd(300, 6)
    dev.set_scaling(0, W, G)
    dev.set_option("profile", 1)
    dev.reset_timing()
    dev.schur_assemble(0)
    assert dev.timing("BBBBone1") == dev.timing("gemm1") > 0
    assert dev.timing("BBBBone2") == dev.timing("gemm2") > 0
    assert dev.timing("BBBBone3") == pytest.approx(dev.timing("gemm3") + dev.timing("reduce3"))
    assert dev.timing("BBBBone4") == 0.0
    assert dev.timing("BBBBs") == dev.timing("assemble") >= dev.timing("BBBBone")


@pytest.mark.parametrize("nvar", [320, 960, 1000])
def test_gemm3_tile_and_schedule_variants_agree(dev, nvar):
    """GEMM3' with the 128 and the 160 workgroup tile, as one launch (regular tiles of every split first, short tiles
    last) and as two.  The schedule does not touch the arithmetic (bit-identical); the tile size changes the tile
    count and with it the split-K factor, i.e. how the K range is cut into slabs (1e-14)."""
    msz = 200
    dev.synthetic_dense_model(msz, nvar, 41)
    W, G = _spd(msz, 42)
    dev.set_scaling(0, W, G)
    dev.set_option("schur_chol", 1)
    Hs = []
    try:
        for tile, sched in ((128, 1), (160, 1), (128, 0), (160, 0)):
            dev.set_option("gemm3_tile", tile)
            dev.set_option("gemm3_sched", sched)
            Hs.append(np.tril(dev.schur_assemble(0, want_H=True)))
    finally:
        dev.set_option("gemm3_tile", 0)
        dev.set_option("gemm3_sched", 1)
        dev.set_option("schur_chol", -1)
    assert np.array_equal(Hs[2], Hs[0]) and np.array_equal(Hs[3], Hs[1])
    assert relerr(Hs[1], Hs[0]) < 1e-14
    A = np.stack([dev.get_constraint(0, k) for k in range(8)])
    assert relerr(Hs[0][:8, :8], np.tril(_brute_H(A, W))) < 1e-13


@pytest.mark.parametrize("msz", [200, 300, 457])
def test_gemm12_pattern_loops_equal_the_branch_per_block_kernel(dev, msz):
    """GEMM1' / GEMM2' of the factor path: the K-steps with skipped blocks (triangular heads, edge tiles of msz % 128 != 0,
    packed diagonal tiles) as straight-line loops per block pattern (round 3) against the round-2 body with a branch per
    block (option gemm_dyn_masks).  A block outside the exact set multiplies stored zeros: bit-identical."""
    nvar = 70
    dev.synthetic_dense_model(msz, nvar, 45)
    W, G = _spd(msz, 46)
    dev.set_scaling(0, W, G)
    dev.set_option("schur_chol", 1)
    Hs = []
    try:
        for dyn in (0, 1):
            dev.set_option("gemm_dyn_masks", dyn)
            Hs.append(np.tril(dev.schur_assemble(0, want_H=True)))
    finally:
        dev.set_option("gemm_dyn_masks", 0)
        dev.set_option("schur_chol", -1)
    assert np.array_equal(Hs[0], Hs[1])
    A = np.stack([dev.get_constraint(0, k) for k in range(6)])
    assert relerr(Hs[0][:6, :6], np.tril(_brute_H(A, W))) < 1e-13


@pytest.mark.parametrize("msz", [200, 300, 457, 1000])
def test_gemm1_leaves_out_the_blocks_gemm2_never_reads(dev, msz):
    """Round 4: in the diagonal tiles of P_k = A_k L the 16x16 blocks above the block diagonal are read by GEMM2' only
    against the stored zeros of L' -- GEMM1' does not compute them and stores zeros (option gemm1_diag, default 1; pattern
    loops and the branch-per-block body).  Bit-identical to computing them, twice in a row (the zeros are re-stored: the
    workspace holds the full products of the gemm1_diag = 0 run in between), and equal to the definition."""
    nvar = 70
    dev.synthetic_dense_model(msz, nvar, 47)
    W, G = _spd(msz, 48)
    dev.set_scaling(0, W, G)
    dev.set_option("schur_chol", 1)
    Hs = []
    try:
        for diag, dyn in ((1, 0), (0, 0), (1, 0), (1, 1)):
            dev.set_option("gemm1_diag", diag)
            dev.set_option("gemm_dyn_masks", dyn)
            Hs.append(np.tril(dev.schur_assemble(0, want_H=True)))
    finally:
        dev.set_option("gemm1_diag", 1)
        dev.set_option("gemm_dyn_masks", 0)
        dev.set_option("schur_chol", -1)
    for H in Hs[1:]:
        assert np.array_equal(Hs[0], H)
    A = np.stack([dev.get_constraint(0, k) for k in range(6)])
    assert relerr(Hs[0][:6, :6], np.tril(_brute_H(A, W))) < 1e-13


@pytest.mark.parametrize("nvar", [416, 1050, 1056])
def test_gemm3_last_tile_row_of_height_160(dev, nvar):
    """nvar % 128 in (0, 32]: the last 128 + nvar % 128 rows of H are tiled by 128 x 160 and one 160 x 160 tile on a second
    stream beside the 128 x 128 tiles of the leading part (gemm_f64.hip, tile_class 4; C4: 4000 = 30 x 128 + 160) instead
    of a row of edge tiles.  Same split-K ranges, same order of the K walk: bit-identical to the edge-tile schedule."""
    msz = 200
    dev.synthetic_dense_model(msz, nvar, 43)
    W, G = _spd(msz, 44)
    dev.set_scaling(0, W, G)
    dev.set_option("schur_chol", 1)
    dev.set_option("gemm3_tile", 128)
    Hs = []
    try:
        for strip in (1, 0, 1):
            dev.set_option("gemm3_strip", strip)
            Hs.append(np.tril(dev.schur_assemble(0, want_H=True)))
    finally:
        dev.set_option("gemm3_strip", 1)
        dev.set_option("gemm3_tile", 0)
        dev.set_option("schur_chol", -1)
    assert np.array_equal(Hs[0], Hs[1]) and np.array_equal(Hs[2], Hs[1])
    rng = np.random.default_rng(5)
    idx = np.unique(np.r_[rng.integers(0, nvar, 6), nvar - 1, nvar - 160, nvar - 161])
    A = np.stack([dev.get_constraint(0, int(k)) for k in idx])
    Hfull = Hs[0] + np.tril(Hs[0], -1).T
    assert relerr(Hfull[np.ix_(idx, idx)], _brute_H(A, W)) < 1e-13


@pytest.mark.parametrize("msz,nvar,mixed", [(256, 30, False), (258, 21, False), (300, 23, True), (272, 9, False)])
def test_dense_passes_over_the_column_tails(dev, msz, nvar, mixed):
    """The passes over dense constraint data read only the column tails rows >= j of the symmetric matrices (round 4):
    AA*vec(Z) with the weights Z + Z' (makeRHS, src/makeBBBB.jl:221-228; Z deliberately NOT symmetric: <A_k, Z> =
    <A_k, sym Z> must come out), and mat(AA'x) as the mirrored lower triangle (src/Solvers.jl:595; with sparse constraints
    beside the dense ones when `mixed`).  msz 258 / 300 / 272: tails that do not end on a chunk boundary."""
    rng = np.random.default_rng(msz + nvar)
    model = _dense_model(msz, nvar, 7, density=1.0)
    if mixed:                    # a few sparse constraints behind the dense ones
        A = [list(model.A[0])]
        for _ in range(6):
            i, j = rng.integers(0, msz, 2)
            M = sp.lil_matrix((msz, msz)); M[i, j] = M[j, i] = rng.standard_normal(); M[i, i] = 1.0
            A[0].append(sp.csc_matrix(M))
        model = lo.make_model(A, rng.standard_normal(len(A[0]) - 1), 0.0, None, None)
    n = model.n
    W, G = _spd(msz, 9)
    dev.set_option("dense_threshold", 1000)
    try:
        _upload(dev, model)
        dev.set_scaling(0, W, G)
        AAd = model.AA[0].toarray()
        # makeRHS with a non-symmetric Rd + S: h = Rp + AA vec(W (Rd+S) W)
        RdS = rng.standard_normal((msz, msz))
        Rp = rng.standard_normal(n)
        h = dev.make_rhs(Rp, [RdS])
        ref = Rp + AAd @ (W @ RdS @ W).reshape(-1, order="F")
        assert relerr(h, ref) < 1e-12
        # MyA through the dense route: mat(AA'x), two products, AA vec(.)
        dev.set_option("matvec_h", 1)
        x = rng.standard_normal(n)
        Mx = (AAd.T @ x).reshape(msz, msz, order="F"); Mx = 0.5 * (Mx + Mx.T)
        assert relerr(dev.matvec(x), AAd @ (W @ Mx @ W).reshape(-1, order="F")) < 1e-12
    finally:
        dev.set_option("dense_threshold", -1)
        dev.set_option("matvec_h", 0)
