"""GPU parity tests of the HIP building blocks (FP64 MFMA GEMM, blocked Cholesky, triangular
solves) through the C ABI.  Tolerances are FP64 round-off scaled by the contraction length."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import loraine_jl_amd
    d = loraine_jl_amd.Device(0)
    yield d
    d.close()


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_mfma_f64_lane_map(dev):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((16, 4))
    B = rng.standard_normal((4, 16))          # asymmetric on purpose
    D = dev.dbg_mfma_probe(A, B)
    ref = A @ B
    if not np.allclose(D, ref, rtol=1e-14, atol=1e-14):
        # diagnose: which row permutation did the hardware use?
        perm = [int(np.argmin(np.abs(ref - D[r]).sum(axis=1))) for r in range(16)]
        raise AssertionError(f"f64 MFMA C/D lane map mismatch; observed row source per output row: {perm}")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 77), (50, 50, 50), (801, 801, 801),
                                   (1, 300, 19), (300, 1, 19), (257, 513, 1000)])
@pytest.mark.parametrize("tA,tB", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_all_layouts(dev, M, N, K, tA, tB):
    rng = np.random.default_rng(M * 7 + N * 3 + K + 2 * tA + tB)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    C0 = rng.standard_normal((M, N))
    C = dev.dbg_gemm(A, B, tA, tB, alpha=0.75, beta=-0.5, Cin=C0)
    ref = 0.75 * (A.T if tA else A) @ (B.T if tB else B) - 0.5 * C0
    assert relerr(C, ref) < 1e-14 * max(8, np.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 256), (2000, 2176, 403), (2050, 2300, 270)])
def test_gemm_direct_to_lds_path(dev, M, N, K):
    """NT layout, even leading dimensions, >= 256 tiles: served by gemm_f64_lds_kernel
    (buffer_load ... lds staging; edges and the K tail come from descriptor range checks)."""
    from loraine_jl_amd import _capi
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K))
    C0 = rng.standard_normal((M, N))
    C = dev.dbg_gemm(A, B, False, True, alpha=1.25, beta=0.5, Cin=C0)
    assert relerr(C, 1.25 * A @ B.T + 0.5 * C0) < 1e-14 * max(8, np.sqrt(K))
    # triangular + x2 epilogue on the same path
    if M == N:
        C = dev.dbg_gemm(A, B, False, True, flags=_capi.GEMM_TRI_LOWER | _capi.GEMM_OFFDIAG_X2)
        ref = A @ B.T
        tiles = np.arange(M) // 128
        low = tiles[:, None] > tiles[None, :]; dia = tiles[:, None] == tiles[None, :]
        assert relerr(C[dia], ref[dia]) < 1e-13 and relerr(C[low], 2 * ref[low]) < 1e-13
        assert not C[tiles[:, None] < tiles[None, :]].any()


@pytest.mark.parametrize("M,N,K", [(2001, 2049, 301), (2177, 2001, 1001)])
def test_gemm_direct_to_lds_path_odd_leading_dimensions(dev, M, N, K):
    """Round 4: rows that are only 8-byte aligned (odd leading dimensions, lda = M, ldb = N) go through the same 16-byte
    LDS DMA -- measured correct on gfx950 (tools/probe_unaligned_dma.py): the last pair of an odd row reads one element of
    the next row (a row of C that is never stored) or, in the last row, past the descriptor's range (zero)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K))
    C = dev.dbg_gemm(A, B, False, True, alpha=1.25)
    assert relerr(C, 1.25 * A @ B.T) < 1e-14 * max(8, np.sqrt(K))


@pytest.mark.parametrize("n", [2200, 3001])
def test_gemm_plain_products_that_do_not_fill_whole_rounds_of_128_tiles(dev, n):
    """Round 4: a plain NT product of 256 .. 1023 128-tiles whose last round of workgroups would run half empty (msz 2200:
    324 tiles, 3001: 576) goes to the 64-tile DMA kernel when the round model prices it cheaper (gemm_f64.hip)."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); B = rng.standard_normal((n, n))
    C = dev.dbg_gemm(A, B, False, True, alpha=0.5)
    assert relerr(C, 0.5 * A @ B.T) < 1e-14 * np.sqrt(n)


@pytest.mark.parametrize("n", [801, 800, 640, 1111])
def test_gemm_mid_size_slabs_kernel(dev, n):
    """Round 4: a plain NT product whose 64-tiles do not fill the chip (msz 400 .. 1400) runs as split-K slabs on
    gemm_f64_mid_kernel (three LDS stages filled by 16-byte DMA from rows of any 8-byte alignment, XOR-swizzled images,
    1-D item grid) + reduce_slabs: against NumPy, odd and even sides."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); B = rng.standard_normal((n, n))
    C = dev.dbg_gemm(A, B, False, True, alpha=-0.5)
    assert relerr(C, -0.5 * A @ B.T) < 1e-14 * np.sqrt(n)


@pytest.mark.parametrize("M,N,K", [(10, 500, 100), (257, 513, 1000), (64, 64, 64), (65, 63, 70), (1, 300, 64), (300, 1, 257),
                                   (129, 1100, 333)])
def test_gemm_mid_kernel_ragged_shapes(dev, M, N, K):
    """gemm_f64_mid_kernel also serves unsplit plain NT products (beta = 0, K >= 64) of any shape that takes the 64 tile:
    rows / columns beyond M / N and k-rows beyond K come back as zeros from the buffer descriptor's range check."""
    rng = np.random.default_rng(M * 5 + N * 3 + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K))
    C = dev.dbg_gemm(A, B, False, True, alpha=2.0)
    assert relerr(C, 2.0 * A @ B.T) < 1e-14 * max(8, np.sqrt(K))


def test_gemm_splitk_matches(dev):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((5000, 300)); B = rng.standard_normal((5000, 260))
    C = dev.dbg_gemm(A, B, True, False, ksplit=7)
    assert relerr(C, A.T @ B) < 1e-13


def test_gemm_tri_flags_and_square(dev):
    from loraine_jl_amd import _capi
    rng = np.random.default_rng(6)
    n, K = 700, 130
    A = rng.standard_normal((n, K))
    C0 = rng.standard_normal((n, n))
    C = dev.dbg_gemm(A, A, False, True, alpha=1.0, beta=1.0, Cin=C0,
                     flags=_capi.GEMM_TRI_LOWER | _capi.GEMM_SQUARE)
    ref = (A @ A.T) ** 2 + C0
    il = np.tril_indices(n)
    assert relerr(C[il], ref[il]) < 1e-13
    # strictly-upper tiles (64- or 128-granular) are untouched
    assert np.array_equal(C[0, n - 1], C0[0, n - 1])


@pytest.mark.parametrize("m", [300, 320])     # 320: K-contiguous direct-to-LDS kernel (msz % 16 == 0)
def test_gemm_packed_symmetric_dot(dev, m):
    """GEMM2-style lower/x2 storage + GEMM3-style K-segment skipping reproduce full <A_i,T_j>."""
    from loraine_jl_amd import _capi
    rng = np.random.default_rng(7)
    nv = 150                              # constraints; msz = m (3 tiles of 128)
    Wm = rng.standard_normal((m, m)); Wm = Wm @ Wm.T / m
    As = rng.standard_normal((nv, m, m)); As = (As + As.transpose(0, 2, 1)) / 2
    Tfull = np.stack([Wm @ a @ Wm for a in As])
    # stored T: lower 128-tiles, strictly-lower x2, upper tiles zero
    T_st = np.zeros_like(Tfull)
    for j in range(nv):
        P = As[j] @ Wm
        Tj = dev.dbg_gemm(Wm, P, flags=_capi.GEMM_TRI_LOWER | _capi.GEMM_OFFDIAG_X2)
        T_st[j] = Tj
    tiles = np.arange(m) // 128
    low = tiles[:, None] > tiles[None, :]
    dia = tiles[:, None] == tiles[None, :]
    assert relerr(T_st[:, dia], Tfull[:, dia]) < 1e-13
    assert relerr(T_st[:, low], 2 * Tfull[:, low]) < 1e-13
    assert np.all(T_st[:, tiles[:, None] < tiles[None, :]] == 0.0)
    Amat = np.asfortranarray(As.transpose(0, 2, 1).reshape(nv, m * m).T)     # (m^2 x nv), col-major vec
    Tmat = np.asfortranarray(T_st.transpose(0, 2, 1).reshape(nv, m * m).T)
    H = dev.dbg_gemm(Amat, Tmat, True, False, flags=_capi.GEMM_TRI_LOWER | _capi.GEMM_KSEG_TRI, ksplit=3)
    Href = As.reshape(nv, -1) @ Tfull.reshape(nv, -1).T
    t2 = np.arange(nv) // 128
    mask = t2[:, None] >= t2[None, :]
    assert relerr(H[mask], Href[mask]) < 1e-13


@pytest.mark.parametrize("n", [1, 7, 64, 65, 104, 500, 1300, 2113])
def test_potrf_potrs(dev, n):
    rng = np.random.default_rng(n)
    Mx = rng.standard_normal((n, n + 3))
    A = Mx @ Mx.T + n * 1e-3 * np.eye(n)
    L, info = dev.dbg_potrf(A)
    assert info == 0
    Lref = np.linalg.cholesky(A)
    assert relerr(L, Lref) < 1e-11
    b = rng.standard_normal(n)
    x, info = dev.dbg_potrs(A, b)
    assert info == 0
    assert relerr(A @ x, b) < 1e-10


@pytest.mark.parametrize("n", [513, 801, 1024, 2049, 2112, 2305, 3000, 4000])
def test_potrs_superblock_path(dev, n):
    """n > 512: the triangular solves run in 256-row super-blocks (one launch each, wave-level substitution through the
    64 x 64 diagonal blocks): sizes with a full, a 64-aligned and a ragged last super-block, against LAPACK."""
    import scipy.linalg as sla
    rng = np.random.default_rng(n)
    Mx = rng.standard_normal((n, n + 3))
    A = Mx @ Mx.T + n * 1e-3 * np.eye(n)
    b = rng.standard_normal(n)
    x, info = dev.dbg_potrs(A, b)
    assert info == 0
    xref = sla.cho_solve(sla.cho_factor(A, lower=True), b)
    assert relerr(x, xref) < 1e-9
    assert relerr(A @ x, b) < 1e-10


@pytest.mark.parametrize("n", [9216, 9500, 10040, 9217])
def test_potrf_two_level_blocking(dev, n):
    """n >= 9000: super-blocks of 1024 columns -- inside one the steps update only its own remaining columns, the rest of
    the trailing matrix gets the sixteen panels at once (K = 1024 GEMM on the lower tiles), the next super-block starts with
    a diagonal-block and a panel launch of its own (round 4, chol.hip).  9216 = 9 x 1024; 9500 and 10040: ragged last
    super-block / last 64-block; a failing pivot in the second super-block is reported at its column."""
    rng = np.random.default_rng(n)
    Mx = rng.standard_normal((n, 64))
    d = 1.0 + rng.random(n)
    A = Mx @ Mx.T + np.diag(d) * 8.0                     # SPD, cond ~ 1e2
    L, info = dev.dbg_potrf(A)
    assert info == 0
    L = np.tril(L)
    assert np.abs(L @ L.T - A).max() <= 1e-13 * np.abs(A).max() * 64
    Lref = np.linalg.cholesky(A)
    assert relerr(L, Lref) < 1e-12
    bad = 1024 + 70                                       # second panel of the second super-block
    B = A.copy()
    B[bad, bad] = -1.0
    _, info = dev.dbg_potrf(B)
    assert info == bad + 1


def test_potrf_reports_not_pd(dev):
    rng = np.random.default_rng(3)
    n = 200
    Mx = rng.standard_normal((n, n))
    A = Mx @ Mx.T
    A[150, 150] = -1.0
    _, info = dev.dbg_potrf(A)
    assert info == 151
    A2 = np.full((5, 5), np.nan)
    _, info = dev.dbg_potrf(A2)
    assert info == 1


@pytest.mark.parametrize("col", [0, 63, 64, 70, 199, 450])
def test_potrf_failing_column_in_every_kind_of_block(dev, col):
    """The first non-positive pivot is reported wherever it falls: in the first diagonal block (a launch of its own), at
    the first column of a later block or inside it (factored by the tile (0, 0) workgroup of the trailing update, whose
    straight-line pivot chain hands a panel with a bad pivot to the careful loop), in the ragged last block; and an exactly
    singular matrix fails at the dependent column."""
    rng = np.random.default_rng(11 + col)
    n = 451
    Mx = rng.standard_normal((n, n + 4))
    A = Mx @ Mx.T + 0.5 * np.eye(n)
    L0 = np.linalg.cholesky(A)
    # make the pivot of column `col` negative: subtract more than the Schur complement's diagonal entry
    A2 = A.copy()
    A2[col, col] -= 1.01 * L0[col, col] ** 2
    _, info = dev.dbg_potrf(A2)
    assert info == col + 1
    # exact rank deficiency: column `col` (> 0) a copy of column 0 -- the pivot is zero up to rounding, never accepted
    # as a large positive number
    if col > 0:
        B = Mx[:, :n].copy()
        B[col, :] = B[0, :]
        A3 = B @ B.T
        L, info = dev.dbg_potrf(A3)
        assert info == col + 1 or (info == 0 and abs(np.tril(L)[col, col]) < 1e-6 * np.abs(np.diag(np.tril(L))).max())


@pytest.mark.parametrize("n,nrhs,trans", [(50, 50, False), (300, 300, True), (801, 17, False), (130, 200, True)])
def test_trsm(dev, n, nrhs, trans):
    rng = np.random.default_rng(n + nrhs)
    Mx = rng.standard_normal((n, n + 5))
    A = Mx @ Mx.T + 0.1 * np.eye(n)
    B = rng.standard_normal((n, nrhs))
    X, info = dev.dbg_trsm(A, B, trans)
    assert info == 0
    L = np.linalg.cholesky(A)
    ref = np.linalg.solve(L.T if trans else L, B)
    assert relerr(X, ref) < 1e-10


def test_probes(dev):
    tf = dev.mfma_f64_peak()
    gb = dev.hbm_copy_peak(1 << 30)
    print(f"FP64 MFMA issue-rate probe: {tf:.1f} TFLOP/s ; HBM copy: {gb:.0f} GB/s")
    assert tf > 10 and gb > 500


@pytest.mark.parametrize("n", [1500, 3000])
def test_two_factorisations_side_by_side_on_two_streams(dev, n):
    """cholesky(X) and cholesky(S) of the NT scaling run on two streams: their workgroups share the slots of the chip, so
    the replicas of a diagonal block (potrf_step_kernel) can start after the leader has stored the factor over the tile --
    they must read their copy of it, not the matrix.  Same bits as one after the other, and LAPACK's factor."""
    import scipy.sparse as sp
    rng = np.random.default_rng(n)
    def spd():
        Mx = rng.standard_normal((n, n + 7)) / np.sqrt(n)
        return Mx @ Mx.T + 0.05 * np.eye(n)
    X, S = spd(), spd()
    nvar = 4
    AA = sp.csc_matrix((np.ones(nvar), (np.arange(nvar), np.arange(nvar) * (n + 1))), shape=(nvar, n * n))
    dev.upload_model([AA], np.arange(nvar, dtype=np.int64).reshape(-1, 1), np.zeros((2, 1), dtype=np.int64), [n])
    dev.ip_set_c(0, np.eye(n))
    got = {}
    try:
        for streams in (1, 0, 1):
            dev.set_option("prepw_streams", streams)
            dev.ip_set_iterate(0, X, S)
            assert dev.ip_prepare_w(0) == 0
            got.setdefault(streams, []).append((np.tril(dev.dbg_get_block(0, "LX")[0]), np.tril(dev.dbg_get_block(0, "LS")[0])))
    finally:
        dev.set_option("prepw_streams", 1)
    (lx1, ls1), (lx1b, ls1b) = got[1]
    lx0, ls0 = got[0][0]
    assert np.array_equal(lx1, lx0) and np.array_equal(ls1, ls0) and np.array_equal(lx1b, lx0) and np.array_equal(ls1b, ls0)
    assert relerr(lx0, np.linalg.cholesky(X)) < 1e-11 and relerr(ls0, np.linalg.cholesky(S)) < 1e-11
