"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy/SciPy restatement of the per-iteration linear-algebra hot path of
kocvara/Loraine.jl v0.2.5 *and* of the interior-point loop that drives it, written
from the algorithm as the reference states it.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product (`loraine.jl_amd/`) never does and fails loudly when the HIP
library is missing.

Every function cites the reference file:line it follows (paths relative to the
reference checkout, e.g. src/prepare_W.jl:28-94).

Pinning status
--------------
* Julia is not installed in the build container, so the reference cannot be executed and
  no golden vectors can be captured from it.
* kit=0 general path: PINNED by the reference's own known answers (theta1 -> 23,
  ex_corr, ex_dist, ex_maxcut, LP k.jl; tests/test_oracle_kat.py).
* kit=1 (cg / MyA / H_alpha / H_beta) and datarank=-1: the reference's tests never run
  them -> **parity unpinned** by the reference; pinned here only by algebraic identities
  (MyA(x) == H x, SMW identity, rank-1 == general on rank-1 data) and by agreement with
  the pinned kit=0 path on the same inputs.
* `cg` restates the published algorithm of ConjugateGradients.jl 0.1 (third-party, source
  absent from the reference checkout); iteration counts are **parity unpinned**.
* `fsvd` (FameSVD.jl 0.1) is replaced by LAPACK SVD; NT scaling is invariant to the order
  and sign of singular triples.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# Options -- src/Solvers.jl:169-185 (code defaults win over README defaults)
# --------------------------------------------------------------------------------------
DEFAULT_OPTIONS = {
    "kit": 0,
    "tol_cg": 1.0e-2,
    "tol_cg_up": 0.5,
    "tol_cg_min": 1.0e-7,
    "eDIMACS": 1.0e-7,
    "preconditioner": 1,
    "erank": 1,
    "aamat": 1,
    "fig_ev": 0,
    "verb": 1,
    "datarank": 0,
    "initpoint": 0,
    "timing": 1,
    "maxit": 100,
    "datasparsity": 8,
}


# --------------------------------------------------------------------------------------
# Model -- src/model.jl:34-87
# --------------------------------------------------------------------------------------
@dataclass
class MyModel:
    A: list            # A[ilmi][k] scipy csc (msz x msz), k = 0..nvar ; A[ilmi][0] = F0
    AA: list           # AA[ilmi] csr (nvar x msz^2), row j = -vec(A[ilmi][j+1])
    B: list            # B[ilmi] csr (nvar x msz) or [] when datarank != -1
    C: list            # C[ilmi] = -A[ilmi][0]
    nzA: np.ndarray    # (nvar, nlmi)
    sigmaA: np.ndarray  # (nvar, nlmi) 0-based permutation, decreasing nnz (stable)
    qA: np.ndarray     # (2, nlmi)
    b: np.ndarray
    b_const: float
    d_lin: np.ndarray
    C_lin: sp.csr_matrix  # (nvar x nlin)
    n: int
    msizes: np.ndarray
    nlin: int
    nlmi: int


def read_sdpa(path: str):
    """Minimal SDPA sparse-format reader (format used by examples/data/*.dat-s).

    Returns dict(nvar, blocks (signed sizes), c, entries) with entries a list of
    (matno, blkno, i, j, val), all 1-based like the file.
    """
    with open(path) as fh:
        lines = []
        for ln in fh:
            s = ln.strip()
            if not s or s[0] in '*"':
                continue
            lines.append(s)

    def nums(s):
        for ch in "{}(),":
            s = s.replace(ch, " ")
        return s.split()

    nvar = int(nums(lines[0])[0])
    nblocks = int(nums(lines[1])[0])
    blocks = [int(float(t)) for t in nums(lines[2])[:nblocks]]
    pos = 3
    cvals = []
    while len(cvals) < nvar:
        cvals += [float(t) for t in nums(lines[pos])]
        pos += 1
    c = np.array(cvals[:nvar], dtype=np.float64)
    entries = []
    for ln in lines[pos:]:
        t = nums(ln)
        if len(t) < 5:
            continue
        entries.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])))
    return dict(nvar=nvar, blocks=blocks, c=c, entries=entries)


def prep_sparse(nz_col: np.ndarray, kappa: int):
    """src/model.jl:153-174 -- nnz-sorted order (stable, decreasing) and dense/sparse split."""
    n = nz_col.shape[0]
    sigma = np.argsort(-nz_col, kind="stable")
    sisi = nz_col[sigma]
    q = n
    for j in range(n):
        if sisi[j] <= kappa:
            q = j
            break
    return sigma, q


def prep_B(Ai: list, n: int):
    """src/model.jl:176-197 -- rank-one factor extraction A_k = b_k b_k^T."""
    m = Ai[0].shape[0]
    rows, cols, vals = [], [], []
    for k in range(n):
        Ak = Ai[k + 1].tocsc()
        ii = Ak.indices
        if ii.size == 0:
            continue
        _, first = np.unique(ii, return_index=True)
        bidx = ii[np.sort(first)]
        tmp = Ak[bidx, :][:, bidx].toarray()
        w, v = np.linalg.eigh((tmp + tmp.T) / 2.0)
        bbb = np.sign(v[:, -1]) * np.sqrt(np.diag(tmp).astype(complex)).real
        tmp2 = np.outer(bbb, bbb)
        err = np.linalg.norm(tmp - tmp2)
        if not err <= 5.0e-6:
            raise ValueError(
                f"Obtained an error of `{err} > 5e-6` when converting matrix into rank `1`, "
                "use `datarank = 0` to disable the rank-1 conversion.")
        rows += [k] * len(bidx)
        cols += list(bidx)
        vals += list(bbb)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, m))


def prepare_A(A: list, datarank: int, kappa: int, n: int = None):
    """src/model.jl:120-150 (+ prep_AA! :199-229)."""
    nlmi = len(A)
    if n is None:
        n = len(A[0]) - 1 if nlmi else 0
    AA, B, C = [], [], []
    nzA = np.zeros((n, nlmi), dtype=np.int64)
    sigmaA = np.zeros((n, nlmi), dtype=np.int64)
    qA = np.zeros((2, nlmi), dtype=np.int64)
    for i in range(nlmi):
        C.append((-A[i][0]).tocsc())
        m = A[i][0].shape[0]
        rows, cols, vals = [], [], []
        for j in range(n):
            Aj = A[i][j + 1].tocoo()
            rows.append(np.full(Aj.nnz, j, dtype=np.int64))
            cols.append(Aj.col.astype(np.int64) * m + Aj.row.astype(np.int64))  # column-major vec
            vals.append(-Aj.data)
            nzA[j, i] = Aj.nnz
        rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
        vals = np.concatenate(vals) if vals else np.zeros(0)
        AA.append(sp.csr_matrix((vals, (rows, cols)), shape=(n, m * m)))
        if datarank == -1:
            B.append(prep_B(A[i], n))
        sigmaA[:, i], q = prep_sparse(nzA[:, i], kappa)
        qA[0, i] = qA[1, i] = q
    return AA, B, C, nzA, sigmaA, qA


def model_from_sdpa(path: str, datarank: int = 0, kappa: int = 8) -> MyModel:
    """SDPA file -> MyModel with the sign conventions of src/MOI_wrapper.jl:142-232.

    SDPA: min c'x s.t. sum_k F_k x_k - F_0 >= 0.  A[lmi][0] = F_0, A[lmi][k] = F_k (both
    triangles), b = -c (Min sense), diagonal (negative-size) blocks become rows of C_lin:
    C_lin = -coef^T (nvar x nlin), d_lin = -F_0[ii]  (MOI_wrapper.jl:145-149,217).
    """
    d = read_sdpa(path)
    nvar = d["nvar"]
    blocks = d["blocks"]
    psd_blocks = [bi for bi, s in enumerate(blocks) if s > 0]
    lin_blocks = [bi for bi, s in enumerate(blocks) if s < 0]
    lmi_of_block = {bi: k for k, bi in enumerate(psd_blocks)}
    lin_off = {}
    off = 0
    for bi in lin_blocks:
        lin_off[bi] = off
        off += -blocks[bi]
    nlin = off
    coo = [[([], [], []) for _ in range(nvar + 1)] for _ in psd_blocks]
    lin_r, lin_c, lin_v = [], [], []
    d_lin = np.zeros(nlin)
    for (mat, blk, i, j, v) in d["entries"]:
        if v == 0.0:
            continue
        bi = blk - 1
        if blocks[bi] > 0:
            I, J, V = coo[lmi_of_block[bi]][mat]
            I.append(i - 1); J.append(j - 1); V.append(v)
            if i != j:
                I.append(j - 1); J.append(i - 1); V.append(v)
        else:
            assert i == j, "off-diagonal entry in a diagonal block"
            r = lin_off[bi] + i - 1
            if mat == 0:
                d_lin[r] += -v            # constants = -F_0
            else:
                lin_r.append(mat - 1); lin_c.append(r); lin_v.append(-v)   # C_lin = -coef^T
    A = []
    for k, bi in enumerate(psd_blocks):
        m = blocks[bi]
        A.append([sp.csc_matrix((V, (I, J)), shape=(m, m)) for (I, J, V) in coo[k]])
    C_lin = sp.csr_matrix((lin_v, (lin_r, lin_c)), shape=(nvar, nlin))
    b = -d["c"]
    return make_model(A, b, 0.0, d_lin, C_lin, datarank, kappa)


def make_model(A, b, b_const, d_lin, C_lin, datarank=0, kappa=8) -> MyModel:
    nlmi = len(A)
    n = len(b)
    msizes = np.array([A[i][0].shape[0] for i in range(nlmi)], dtype=np.int64)
    for i in range(nlmi):
        for k in range(n + 1):
            A[i][k] = sp.csc_matrix(A[i][k])
            A[i][k].eliminate_zeros()
    AA, B, C, nzA, sigmaA, qA = prepare_A(A, datarank, kappa, n)
    nlin = 0 if C_lin is None else C_lin.shape[1]
    if C_lin is None:
        C_lin = sp.csr_matrix((n, 0))
        d_lin = np.zeros(0)
    return MyModel(A, AA, B, C, nzA, sigmaA, qA, np.asarray(b, float), float(b_const),
                   np.asarray(d_lin, float), sp.csr_matrix(C_lin), n, msizes, nlin, nlmi)


# --------------------------------------------------------------------------------------
# helpers -- src/kron_etc.jl
# --------------------------------------------------------------------------------------
def vec(M):
    return np.asarray(M).reshape(-1, order="F")


def mat(v):
    """src/kron_etc.jl:13-18 -- reshape + symmetrise."""
    n = math.isqrt(v.size)
    At = np.asarray(v).reshape(n, n, order="F")
    return (At + At.T) / 2.0


def my_kron(A, B, Cm):
    """src/kron_etc.jl:4-11 -- vec(B*C*A')."""
    return vec(B @ (Cm @ A.T))


def btrace(nlmi, X, S):
    """src/kron_etc.jl:21-28."""
    t = 0.0
    for i in range(nlmi):
        Xi = X[i].toarray() if sp.issparse(X[i]) else X[i]
        Si = S[i].toarray() if sp.issparse(S[i]) else S[i]
        t += float(np.sum(Xi * Si))
    return t


# --------------------------------------------------------------------------------------
# NT scaling -- src/prepare_W.jl
# --------------------------------------------------------------------------------------
def _chol_lower(M):
    return np.linalg.cholesky(M)


def try_cholesky(solver, Xl, i, name):
    """src/prepare_W.jl:5-26 -- Cholesky with 1e-5*I regularisation loop."""
    try:
        return _chol_lower(Xl[i])
    except np.linalg.LinAlgError:
        icount = 0
        while True:
            try:
                _chol_lower(Xl[i])
                break
            except np.linalg.LinAlgError:
                pass
            Xl[i] = Xl[i] + 1e-5 * np.eye(Xl[i].shape[0])
            icount += 1
            if icount > 1000:
                solver.status = 4
                return np.eye(Xl[i].shape[0])
        return _chol_lower(Xl[i])


def prepare_W(solver):
    """src/prepare_W.jl:28-94."""
    for i in range(solver.model.nlmi):
        LX = try_cholesky(solver, solver.X, i, "X")
        LS = try_cholesky(solver, solver.S, i, "S")
        CC = LS.T @ LX                                   # :39
        _, Dtmp, Vt = np.linalg.svd(CC)                  # :42 (fsvd -> LAPACK)
        V = Vt.T
        solver.D[i] = Dtmp.copy()                        # :50
        Di2 = 1.0 / np.sqrt(Dtmp)                        # :52
        G = (LX @ V) * Di2[None, :]                      # :60
        solver.G[i] = G
        solver.Gi[i] = np.linalg.inv(G)                  # :63
        solver.W[i] = G @ G.T                            # :64
        m = LS.shape[0]
        Linv = sla.solve_triangular(LS, np.eye(m), lower=True)
        solver.Si[i] = sla.solve_triangular(LS.T, Linv, lower=False)   # :68
        DD = G.T @ solver.S[i] @ G                       # :71
        DD = (DD + DD.T) / 2.0
        solver.DDsi[i] = 1.0 / np.sqrt(np.diag(DD))      # :74
    if solver.model.nlin > 0:
        solver.Si_lin = 1.0 / solver.S_lin               # :86
    else:
        solver.Si_lin = np.zeros(0)
    return solver.D, solver.G, solver.Gi, solver.W, solver.Si, solver.DDsi, solver.Si_lin


# --------------------------------------------------------------------------------------
# Schur complement assembly -- src/makeBBBB.jl
# --------------------------------------------------------------------------------------
def makeBBBB_rank1(n, nlmi, B, G):
    """src/makeBBBB.jl:1-20 -- H = sum_lmi ((B G)(B G)^T).^2."""
    BBBB = np.zeros((n, n))
    for ilmi in range(nlmi):
        BB = (B[ilmi] @ G[ilmi]).T          # msz x n   :7
        tmp = BB.T @ BB                     # :10
        BBBB = BBBB + tmp ** 2              # :12-14
    return BBBB


def _dot(A, Bm, W):
    """src/makeBBBB.jl:39-64 -- literal <A*W, W*B> for symmetric CSC A, B."""
    A = A.tocsc(); Bm = Bm.tocsc()
    result = 0.0
    for i in range(A.shape[1]):
        a0, a1 = A.indptr[i], A.indptr[i + 1]
        if a0 == a1:
            continue
        for j in range(Bm.shape[1]):
            b0, b1 = Bm.indptr[j], Bm.indptr[j + 1]
            if b0 == b1:
                continue
            AW = 0.0
            for k in range(a0, a1):
                AW += A.data[k] * W[A.indices[k], j]
            WB = 0.0
            for k in range(b0, b1):
                WB += W[i, Bm.indices[k]] * Bm.data[k]
            result += AW * WB
    return result


def makeBBBBsi_literal(ilmi, Ailmi, AAilmi, Wilmi, n, qA, sigmaA):
    """src/makeBBBB.jl:67-218, loop-for-loop (small inputs only; used to validate the
    vectorised restatement below)."""
    BBBB = np.zeros((n, n))
    for ii in range(n):
        i = sigmaA[ii, ilmi]
        Ai = Ailmi[i + 1]
        if Ai.nnz == 0:
            continue
        if ii < qA[0, ilmi]:                                 # branch 1  :81-104
            tmp1 = Wilmi @ Ai.toarray()                      # :88
            tmp = tmp1 @ Wilmi                               # :92
            tmp2 = AAilmi @ vec(tmp)                         # :95
            indi = sigmaA[ii:, ilmi]
            BBBB[indi, i] = -tmp2[indi]                      # :100
            BBBB[i, indi] = -tmp2[indi]                      # :101
        else:                                                # branch 3  :139-213
            if Ai.nnz > 1:
                for jj in range(ii, n):
                    j = sigmaA[jj, ilmi]
                    Aj = Ailmi[j + 1]
                    if Aj.nnz == 0:
                        continue
                    ttt = _dot(Ai, Aj, Wilmi)                # :161
                    if i >= j:
                        BBBB[i, j] = ttt
                    else:
                        BBBB[j, i] = ttt
            else:
                Ai_c = Ai.tocoo()
                r = int(Ai_c.row[0]); vi = float(Ai_c.data[0])
                for jj in range(ii, n):
                    j = sigmaA[jj, ilmi]
                    Aj = Ailmi[j + 1]
                    if Aj.nnz == 0:
                        continue
                    Aj_c = Aj.tocoo()
                    rj = int(Aj_c.row[0]); vj = float(Aj_c.data[0])
                    ttt = vi * Wilmi[r, rj] * Wilmi[r, rj] * vj   # :201
                    if i >= j:
                        BBBB[i, j] = ttt
                    else:
                        BBBB[j, i] = ttt
    return BBBB


def makeBBBBsi(ilmi, Ailmi, AAilmi, Wilmi, n, qA, sigmaA, ii_stop=None):
    """src/makeBBBB.jl:67-218 -- same branch structure and write pattern as the reference,
    with the inner j-loops vectorised (row i against all j at once).  `ii_stop` (bench.py's bounded CPU sample
    only): leave the constraint loop :77 after that many positions."""
    m = Wilmi.shape[0]
    BBBB = np.zeros((n, n))
    nnz_per = np.asarray(AAilmi.getnnz(axis=1)).ravel()
    pat = {}

    def pattern():
        # union pattern P of vec-positions touched by any constraint (sparse branches only: built on first use)
        if not pat:
            AAc = AAilmi.tocsr()
            P = np.unique(AAc.indices)
            pat.update(P=P, Pp=P % m, Pq=P // m, AAP=AAc[:, P].tocsr())         # AAP: n x |P|
        return pat["P"], pat["Pp"], pat["Pq"], pat["AAP"]

    nnz_all = int(nnz_per.sum())
    for ii in range(n if ii_stop is None else min(n, ii_stop)):
        i = sigmaA[ii, ilmi]
        if nnz_per[i] == 0:
            continue
        Ai = Ailmi[i + 1].tocoo()
        if ii < qA[0, ilmi]:
            # (the union pattern is at least nnz_all / n wide: dense data never builds it)
            if Ai.nnz * 4 > m * m or nnz_all * 8 > n * m * m or pattern()[0].size * 8 > m * m:
                tmp = (Wilmi @ Ailmi[i + 1].toarray()) @ Wilmi
                tmp2 = AAilmi @ vec(tmp)
            else:
                # T_i on the union pattern only: T[p,q] = sum_(r,c) a_rc W[p,r] W[c,q]
                P, Pp, Pq, AAP = pattern()
                tP = np.zeros(P.size)
                for r, c, a in zip(Ai.row, Ai.col, Ai.data):
                    tP += a * Wilmi[Pp, r] * Wilmi[c, Pq]
                tmp2 = AAP @ tP
            indi = sigmaA[ii:, ilmi]
            BBBB[indi, i] = -tmp2[indi]
            BBBB[i, indi] = -tmp2[indi]
        else:
            P, Pp, Pq, AAP = pattern()
            tP = np.zeros(P.size)
            for r, c, a in zip(Ai.row, Ai.col, Ai.data):
                tP += a * Wilmi[Pp, r] * Wilmi[c, Pq]
            # AA holds -A; two minus signs cancel for <A_i, W A_j W> written from A entries
            row = -(AAP @ tP)
            js = sigmaA[ii:, ilmi]
            js = js[nnz_per[js] > 0]
            lo = np.minimum(i, js)
            hi = np.maximum(i, js)
            BBBB[hi, lo] = row[js]
    return BBBB


def makeBBBBs(n, nlmi, A, AA, W, qA, sigmaA, literal=False):
    """src/makeBBBB.jl:24-36."""
    BBBB = np.zeros((n, n))
    f = makeBBBBsi_literal if literal else makeBBBBsi
    for ilmi in range(nlmi):
        BBBB += f(ilmi, A[ilmi], AA[ilmi], W[ilmi], n, qA, sigmaA)
    return BBBB


def makeRHS(nlmi, AA, W, S, Rp, Rd):
    """src/makeBBBB.jl:221-228."""
    h = Rp.copy()
    for i in range(nlmi):
        h = h + AA[i] @ vec(W[i] @ (Rd[i] + S[i]) @ W[i])
    return h


# --------------------------------------------------------------------------------------
# CG operators and preconditioners -- src/Solvers.jl:570-904
# --------------------------------------------------------------------------------------
class MyA:
    """src/Solvers.jl:572-614 -- Ax = sum AA vec(W mat(AA' x) W) [+ C_lin(..)]."""

    def __init__(self, W, AA, nlin, C_lin, X_lin, S_lin_inv):
        self.W, self.AA, self.nlin, self.C_lin = W, AA, nlin, C_lin
        self.X_lin, self.S_lin_inv = X_lin, S_lin_inv

    def __call__(self, Ax, x):
        m = self.AA[0].shape[0]
        ax1 = np.zeros(m)
        for ilmi in range(len(self.AA)):
            ax = self.AA[ilmi].T @ x                          # :595
            waxw = self.W[ilmi] @ mat(ax) @ self.W[ilmi]      # :600-601
            ax1 += self.AA[ilmi] @ vec(waxw)                  # :604
        if self.nlin > 0:
            ax1 += self.C_lin @ ((self.X_lin * self.S_lin_inv) * (self.C_lin.T @ x))   # :609
        Ax[:] = ax1


class MyM_no:
    """src/Solvers.jl:616-622."""

    def __call__(self, Mx, x):
        Mx[:] = x


class Halpha:
    """src/Solvers.jl:149-162."""

    def __init__(self, kit):
        self.kit = kit
        self.Umat = []
        self.Z = []
        self.cholS = None
        self.AAAATtau = None


def _tau(lambda_s, aamat):
    """src/Solvers.jl:646-650 / :715-719."""
    if aamat == 0:
        return 1.0 * np.min(lambda_s)
    return (np.min(lambda_s) + np.mean(lambda_s)) / 2.0 - 1.0e-14


def Prec_for_CG_beta(solver, halpha):
    """src/Solvers.jl:624-663 -- H_beta: diagonal d = sum tau^2 (+ diag(C_lin D C_lin'))."""
    nvar = solver.model.n
    d = np.zeros(nvar)
    for ilmi in range(solver.model.nlmi):
        n = solver.W[ilmi].shape[0]
        k = solver.erank
        lam = np.linalg.eigvalsh(solver.W[ilmi])              # :642 (ascending)
        lambda_s = lam[: n - k]
        ttau = _tau(lambda_s, solver.aamat)
        if solver.aamat < 3:
            d += ttau ** 2
    if solver.model.nlmi > 0 and solver.model.nlin > 0:
        Cl = solver.model.C_lin
        xs = solver.X_lin * solver.S_lin_inv
        d += np.asarray(Cl.multiply(Cl).dot(xs)).ravel()      # :660
    halpha.AAAATtau = d


class MyM_beta:
    """src/Solvers.jl:665-672."""

    def __init__(self, AA, AAAATtau):
        self.d = AAAATtau

    def __call__(self, Mx, x):
        Mx[:] = x / self.d


def Prec_for_CG_tilS_prep(solver, halpha):
    """src/Solvers.jl:674-809 (+ prec_alpha_S! :819-864) -- H_alpha setup.

    The k>1 "slow" formula materialises kron(Umat, Z) in the reference (:759); here the
    same S is formed as k products AU_a * Z (identical algebra)."""
    model = solver.model
    nlmi, nvar, k = model.nlmi, model.n, solver.erank
    halpha.Z = []
    halpha.Umat = [None] * nlmi
    dvec = np.zeros(nvar)
    for ilmi in range(nlmi):
        n = solver.W[ilmi].shape[0]
        lam, vect = np.linalg.eigh(solver.W[ilmi])            # :706
        vect_l = vect[:, n - k:]
        lambda_l = lam[n - k:]
        vect_s = vect[:, : n - k]
        lambda_s = lam[: n - k]
        ttau = _tau(lambda_s, solver.aamat)
        Umat = vect_l * np.sqrt(lambda_l - ttau)[None, :]     # :721-722
        halpha.Umat[ilmi] = Umat
        VV = np.hstack([vect_s, vect_l])
        W0 = (VV * np.concatenate([lambda_s, np.full(k, ttau)])[None, :]) @ VV.T   # :725
        W0 = (W0 + W0.T) / 2.0
        Z = np.linalg.cholesky(2.0 * W0 + Umat @ Umat.T)      # :730
        halpha.Z.append(Z)
        if solver.aamat < 3:
            dvec += ttau ** 2                                 # :739
    Dlin = None
    if model.nlin > 0:
        xs = solver.X_lin * solver.S_lin_inv
        Dlin = (model.C_lin @ sp.diags(xs) @ model.C_lin.T).tocsc()   # :744
        AAAATtau = (sp.diags(dvec) + Dlin).tocsc()
    else:
        AAAATtau = sp.diags(dvec).tocsc()
    halpha.AAAATtau = AAAATtau
    # t = [AU_a Z]_{lmi,a}
    cols = []
    for ilmi in range(nlmi):
        n = solver.W[ilmi].shape[0]
        AAc = model.AA[ilmi].tocoo()
        q = AAc.col // n
        p = AAc.col % n
        for a in range(k):
            AU = sp.csr_matrix((AAc.data * halpha.Umat[ilmi][q, a], (AAc.row, p)), shape=(nvar, n))
            cols.append(AU @ halpha.Z[ilmi])
    t = np.hstack(cols)
    if k > 1 or model.nlin > 0:
        import scipy.sparse.linalg as spla
        if model.nlin > 0:
            rhs = spla.spsolve(AAAATtau, t) if t.shape[1] > 1 else spla.spsolve(AAAATtau, t[:, 0])[:, None]
            S = t.T @ np.asarray(rhs).reshape(t.shape)
        else:
            S = t.T @ (t / dvec[:, None])                     # :767
    else:
        ts = t / np.sqrt(dvec)[:, None]                       # :770,:833
        S = ts.T @ ts                                         # :859
    S = (S + S.T) / 2.0 + np.eye(S.shape[0])                  # :804
    halpha.cholS = np.linalg.cholesky(S)                      # :805


class MyM:
    """src/Solvers.jl:811-817,866-904 -- SMW apply of H_alpha^{-1}."""

    def __init__(self, AA, AAAATtau, Umat, Z, cholS):
        self.AA, self.AAAATtau, self.Umat, self.Z, self.cholS = AA, AAAATtau, Umat, Z, cholS
        import scipy.sparse.linalg as spla
        dd = AAAATtau.diagonal()
        if (AAAATtau - sp.diags(dd)).nnz == 0:
            self._solve = lambda v: v / dd
        else:
            lu = spla.splu(AAAATtau.tocsc())
            self._solve = lu.solve

    def __call__(self, Mx, x):
        nvar = x.shape[0]
        nlmi = len(self.AA)
        AAAAinvx = self._solve(x)                              # :874
        y33 = []
        for ilmi in range(nlmi):
            y22 = self.AA[ilmi].T @ AAAAinvx                   # :878
            y33.append(vec(self.Z[ilmi].T @ mat(y22) @ self.Umat[ilmi]))   # :879
        y33 = np.concatenate(y33) if y33 else np.zeros(0)
        y33 = sla.cho_solve((self.cholS, True), y33)           # :883
        yy2 = np.zeros(nvar)
        ii = 0
        for ilmi in range(nlmi):
            n, k = self.Umat[ilmi].shape
            yy = np.zeros(n * n)
            for a in range(k):
                xx = self.Z[ilmi] @ y33[ii:ii + n]             # :892
                yy += np.kron(self.Umat[ilmi][:, a], xx)       # :893
                ii += n
            yy2 += self.AA[ilmi] @ yy                          # :896
        yyy2 = self._solve(yy2)                                # :900
        Mx[:] = AAAAinvx - yyy2                                # :902


def cg(A, b, tol=1e-6, maxIter=100, precon=None):
    """Restatement of ConjugateGradients.jl 0.1 `cg` (third-party; source absent from the
    reference checkout -- call sites src/predictor_corrector.jl:134,235).  x0 = 0, relative
    residual ||r||/||r0|| <= tol, exits: ||b||==0 -> (1,0); ||r0||<=tol -> (2,0);
    alpha<0 or Inf -> (-13,it); converged -> (30,it); maxIter -> (-2,maxIter)."""
    n = b.shape[0]
    x = np.zeros(n)
    if precon is None:
        precon = MyM_no()
    if np.linalg.norm(b) == 0.0:
        return x, 1, 0
    r = np.zeros(n); z = np.zeros(n); Ap = np.zeros(n)
    A(r, x)
    r = b - r
    residual_0 = np.linalg.norm(r)
    if residual_0 <= tol:
        return x, 2, 0
    precon(z, r)
    p = z.copy()
    for it in range(1, maxIter + 1):
        A(Ap, p)
        gamma = float(r @ z)
        pAp = float(p @ Ap)
        alpha = gamma / pAp if pAp != 0.0 else math.inf
        if alpha == math.inf or alpha < 0:
            return x, -13, it
        x += alpha * p
        r -= alpha * Ap
        residual = np.linalg.norm(r) / residual_0
        if residual <= tol:
            return x, 30, it
        precon(z, r)
        beta = float(z @ r) / gamma
        p = z + beta * p
    return x, -2, maxIter


# --------------------------------------------------------------------------------------
# Solver state + IP loop -- src/Solvers.jl:18-147,304-568; initial_point.jl;
# predictor_corrector.jl
# --------------------------------------------------------------------------------------
class MySolver:
    def __init__(self, model: MyModel, options: Optional[dict] = None):
        o = dict(DEFAULT_OPTIONS)
        if options:
            for k_, v_ in options.items():
                if k_ not in DEFAULT_OPTIONS:
                    raise KeyError(f"unsupported option {k_}")     # MOI_wrapper.jl:86-103
                o[k_] = v_
        self.kit = int(o["kit"]); self.tol_cg = float(o["tol_cg"])
        self.tol_cg_up = float(o["tol_cg_up"]); self.tol_cg_min = float(o["tol_cg_min"])
        self.eDIMACS = float(o["eDIMACS"]); self.preconditioner = int(o["preconditioner"])
        self.erank = int(o["erank"]); self.aamat = int(o["aamat"]); self.fig_ev = int(o["fig_ev"])
        self.verb = int(o["verb"]); self.datarank = int(o["datarank"])
        self.initpoint = int(o["initpoint"]); self.timing = int(o["timing"])
        self.maxit = int(o["maxit"]); self.datasparsity = int(o["datasparsity"])
        self.model = model
        # range checks, src/Solvers.jl:263-291
        if self.kit < 0 or self.kit > 1:
            self.kit = 0
        if self.tol_cg < self.tol_cg_min and self.kit == 1:
            self.tol_cg = self.tol_cg_min
        if self.tol_cg_min > self.eDIMACS and self.kit == 1:
            self.tol_cg_min = self.eDIMACS
        if self.kit == 1 and (self.preconditioner < 0 or self.preconditioner > 4):
            self.preconditioner = 1
        if self.erank < 0:
            self.erank = 1
        if self.datarank < -1:
            self.datarank = 0
        if self.initpoint < 0 or self.initpoint > 1:
            self.initpoint = 1
        self.cg_iter_tot = 0
        self.status = 0
        self.trace = []       # per-iteration record for parity tests
        self.hooks = {}       # test hooks (e.g. capture H)


def setup_solver(solver, halpha):
    """src/Solvers.jl:363-446."""
    m = solver.model
    z = lambda: [np.zeros((int(s), int(s))) for s in m.msizes]
    solver.X = z(); solver.S = z(); solver.delX = z(); solver.delS = z()
    solver.D = [np.zeros(int(s)) for s in m.msizes]
    solver.G = z(); solver.Gi = z(); solver.W = z(); solver.Si = z()
    solver.DDsi = [np.zeros(int(s)) for s in m.msizes]
    solver.Rd = z(); solver.Rc = z(); solver.Xn = z(); solver.Sn = z(); solver.RNT = z()
    solver.alpha = np.zeros(m.nlmi); solver.beta = np.zeros(m.nlmi)
    solver.regcount = 0
    if solver.kit == 1:
        if m.nlmi == 0:
            solver.kit = 0
        elif m.nlmi > 0 and solver.erank >= int(np.max(m.msizes)) - 1:
            solver.kit = 0
    if len(m.B) > 0:
        for ilmi in range(m.nlmi):
            if m.B[ilmi].nnz == 0:
                solver.datarank = 0


def initial_point(solver):
    """src/initial_point.jl:1-81."""
    m = solver.model
    n = len(m.b)
    solver.y = np.zeros(n)
    b2 = 1.0 + np.abs(m.b)
    f = 0.0
    for i in range(m.nlmi):
        msz = float(m.msizes[i])
        if solver.initpoint == 0:
            Eps = 1.0
        else:
            f = np.linalg.norm(b2) / (1.0 + sp.linalg.norm(m.AA[i]))
            Eps = math.sqrt(msz) * max(1.0, math.sqrt(msz) * f)
        solver.X[i] = Eps * np.eye(int(msz))
        if solver.initpoint == 0:
            Eta = float(m.n)
        else:
            mf = max(f, sp.linalg.norm(m.C[i]))
            mf = (1.0 + mf) / math.sqrt(msz)
            Eta = math.sqrt(msz) * max(1.0, mf)
        solver.S[i] = Eta * np.eye(int(msz))
    dd = m.nlin
    if m.nlin > 0:
        Cl = m.C_lin.tocsr()
        rown = np.sqrt(np.asarray(Cl.multiply(Cl).sum(axis=1)).ravel())
        if solver.initpoint == 0:
            Epss = 1.0
        else:
            p = b2 / (1.0 + rown)
            Epss = max(1.0, float(np.max(p)))
        solver.X_lin = Epss * np.ones(dd)
        if solver.initpoint == 0:
            Etaa = 1.0
        else:
            mf = max(float(np.max(rown)), float(np.linalg.norm(m.d_lin)))
            mf = mf / math.sqrt(dd)
            Etaa = max(1.0, mf)
        solver.S_lin = Etaa * np.ones(dd)
        solver.S_lin_inv = 1.0 / solver.S_lin
    else:
        solver.X_lin = np.zeros(0); solver.S_lin = np.zeros(0); solver.S_lin_inv = np.zeros(0)
    solver.delX_lin = np.zeros(dd); solver.delS_lin = np.zeros(dd)
    solver.Xn_lin = np.zeros(dd); solver.Sn_lin = np.zeros(dd); solver.RNT_lin = np.zeros(dd)
    solver.Rd_lin = np.zeros(dd)
    solver.sigma = 3.0
    solver.tau = 0.95
    solver.expon = 3.0
    solver.DIMACS_error = 1.0
    solver.iter = 0
    solver.status = 0


def find_mu(solver):
    """src/Solvers.jl:480-494."""
    m = solver.model
    trXS = 0.0
    for i in range(m.nlmi):
        trXS += float(np.sum(solver.X[i] * solver.S[i]))
    mu = trXS
    if m.nlin > 0:
        mu += float(solver.X_lin @ solver.S_lin)
    solver.mu = mu / (float(np.sum(m.msizes)) + m.nlin)
    return solver.mu


def _lin_schur(solver):
    m = solver.model
    xs = solver.X_lin * solver.S_lin_inv
    return (m.C_lin @ sp.diags(xs) @ m.C_lin.T).toarray()


def predictor(solver, halpha):
    """src/predictor_corrector.jl:5-146."""
    m = solver.model
    solver.predict = True
    Rp = m.b.copy()
    for i in range(m.nlmi):
        Rp = Rp - m.AA[i] @ vec(solver.X[i])
        solver.Rd[i] = m.C[i].toarray() - solver.S[i] - mat(m.AA[i].T @ solver.y)
    if m.nlin > 0:
        Rp = Rp - m.C_lin @ solver.X_lin
        solver.Rd_lin = m.d_lin - solver.S_lin - m.C_lin.T @ solver.y
    solver.Rp = Rp

    if solver.kit == 0:
        t0 = time.perf_counter()
        if m.nlmi > 0:
            if solver.datarank == -1:
                BBBB = makeBBBB_rank1(m.n, m.nlmi, m.B, solver.G)
            else:
                BBBB = makeBBBBs(m.n, m.nlmi, m.A, m.AA, solver.W, m.qA, m.sigmaA)
        else:
            BBBB = np.zeros((m.n, m.n))
        if m.nlin > 0:
            BBBB = BBBB + _lin_schur(solver)
        solver.t_assembly = time.perf_counter() - t0
        if "H" in solver.hooks:
            solver.hooks["H"](solver, BBBB)

    if m.nlmi > 0:
        h = makeRHS(m.nlmi, m.AA, solver.W, solver.S, solver.Rp, solver.Rd)
    else:
        h = solver.Rp.copy()
    if m.nlin > 0:
        h = h + m.C_lin @ ((solver.X_lin * solver.Si_lin) * solver.Rd_lin + solver.X_lin)

    if solver.kit == 0:
        t0 = time.perf_counter()
        Hl = np.tril(BBBB)                       # Hermitian(BBBB,:L)  :39
        Hs = Hl + np.tril(Hl, -1).T
        solver.chol_is_object = False
        try:
            L = np.linalg.cholesky(Hs)           # :57
        except np.linalg.LinAlgError:
            solver.regcount += 1                 # :59-85
            if solver.regcount > 5:
                solver.cholBBBB = np.eye(m.n)
                solver.status = 3
                return
            icount = 0
            while True:
                try:
                    L = np.linalg.cholesky(Hs)
                    solver.reg_adds = icount
                    break
                except np.linalg.LinAlgError:
                    Hs = Hs + 1e-4 * np.eye(m.n)
                    icount += 1
                    if icount > 1000:
                        solver.cholBBBB = np.eye(m.n)
                        solver.status = 3
                        return
            # :85 stores the Cholesky OBJECT here, where :57-58 keeps the factor L: `cholBBBB' \ (cholBBBB \ h)`
            # (:90, :199) is then H_reg^-1 (H_reg^-1 h) for the rest of this IP iteration -- restated as is
            solver.chol_is_object = True
        solver.cholBBBB = L
        solver.dely = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)  # :90
        if solver.chol_is_object:
            solver.dely = sla.solve_triangular(L.T, sla.solve_triangular(L, solver.dely, lower=True), lower=False)
        solver.t_solve = time.perf_counter() - t0
    else:
        t0 = time.perf_counter()
        A = MyA(solver.W, m.AA, m.nlin, m.C_lin, solver.X_lin, solver.S_lin_inv)
        if solver.preconditioner == 0:
            M = MyM_no()
        elif solver.preconditioner == 1:
            Prec_for_CG_tilS_prep(solver, halpha)
            M = MyM(m.AA, halpha.AAAATtau, halpha.Umat, halpha.Z, halpha.cholS)
        else:
            Prec_for_CG_beta(solver, halpha)
            M = MyM_beta(m.AA, halpha.AAAATtau)
        solver.dely, exit_code, num_iters = cg(A, h, tol=solver.tol_cg, maxIter=10000, precon=M)
        solver.cg_iter_pre += num_iters
        solver.cg_iter_tot += num_iters
        solver.t_solve = time.perf_counter() - t0
        solver.t_assembly = 0.0
    solver.h_pred = h
    find_step(solver)


def sigma_update(solver):
    """src/predictor_corrector.jl:148-179."""
    m = solver.model
    amin = min([*solver.alpha, solver.alpha_lin])
    bmin = min([*solver.beta, solver.beta_lin])
    step_pred = min(amin, bmin)
    if solver.mu > 1e-6:
        if step_pred < 1.0 / math.sqrt(3.0):
            expon_used = 1.0
        else:
            expon_used = max(solver.expon, 3.0 * step_pred ** 2)
    else:
        expon_used = max(1.0, min(solver.expon, 3.0 * step_pred ** 2))
    if btrace(m.nlmi, solver.Xn, solver.Sn) < 0:
        solver.sigma = 0.8
    else:
        tmp1 = btrace(m.nlmi, solver.Xn, solver.Sn) if m.nlmi > 0 else 0.0
        tmp2 = float(solver.Xn_lin @ solver.Sn_lin) if m.nlin > 0 else 0.0
        tmp12 = (tmp1 + tmp2) / (float(np.sum(m.msizes)) + m.nlin)
        solver.sigma = min(1.0, (tmp12 / solver.mu) ** expon_used)
    return solver.sigma


def corrector(solver, halpha):
    """src/predictor_corrector.jl:181-246."""
    m = solver.model
    solver.predict = False
    h = solver.Rp.copy()
    for i in range(m.nlmi):
        G = solver.G[i]
        inner = G.T @ solver.Rd[i] @ G + np.diag(solver.D[i]) \
            - np.diag((solver.sigma * solver.mu) / solver.D[i]) - solver.RNT[i]
        h = h + m.AA[i] @ my_kron(G, G, inner)                 # :186
    if m.nlin > 0:
        tmp = (solver.delX_lin * solver.delS_lin) * solver.Si_lin - (solver.sigma * solver.mu) * solver.Si_lin
        h = h + m.C_lin @ ((solver.X_lin * solver.Si_lin) * solver.Rd_lin + solver.X_lin + tmp)
    t0 = time.perf_counter()
    if solver.kit == 0:
        L = solver.cholBBBB
        solver.dely = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)  # :199
        if getattr(solver, "chol_is_object", False):
            solver.dely = sla.solve_triangular(L.T, sla.solve_triangular(L, solver.dely, lower=True), lower=False)
    else:
        A = MyA(solver.W, m.AA, m.nlin, m.C_lin, solver.X_lin, solver.S_lin_inv)
        if solver.preconditioner == 0:
            M = MyM_no()
        elif solver.preconditioner == 1:
            M = MyM(m.AA, halpha.AAAATtau, halpha.Umat, halpha.Z, halpha.cholS)
        else:
            M = MyM_beta(m.AA, halpha.AAAATtau)
        solver.dely, exit_code, num_iters = cg(A, h, tol=solver.tol_cg, maxIter=10000, precon=M)
        solver.cg_iter_cor += num_iters
        solver.cg_iter_tot += num_iters
    solver.t_solve += time.perf_counter() - t0
    solver.h_corr = h
    find_step(solver)


def _eigmin(M):
    return float(sla.eigvalsh(M, subset_by_index=[0, 0])[0])


def find_step(solver):
    """src/predictor_corrector.jl:248-326."""
    m = solver.model
    for i in range(m.nlmi):
        W, G, Gi = solver.W[i], solver.G[i], solver.Gi[i]
        solver.delS[i] = solver.Rd[i] - mat(m.AA[i].T @ solver.dely)          # :252
        Xi = my_kron(W, W, solver.delS[i])                                    # :253
        if solver.predict:
            solver.delX[i] = mat(-vec(solver.X[i]) - Xi)                      # :255
        else:
            solver.delX[i] = mat(vec((solver.sigma * solver.mu) * solver.Si[i] - solver.X[i]) - Xi
                                 + my_kron(G, G, solver.RNT[i]))              # :257
        delSb = G.T @ solver.delS[i] @ G                                      # :263
        delXb = Gi @ solver.delX[i] @ Gi.T                                    # :264
        dd = solver.DDsi[i]
        XXX = dd[None, :] * delXb * dd[:, None]                               # :268
        XXX = (XXX + XXX.T) / 2.0
        mimiX = _eigmin(XXX)
        solver.alpha[i] = 0.99 if mimiX > -1e-6 else min(1.0, -solver.tau / mimiX)
        XXX = dd[None, :] * delSb * dd[:, None]
        XXX = (XXX + XXX.T) / 2.0
        mimiS = _eigmin(XXX)
        solver.beta[i] = 0.99 if mimiS > -1e-6 else min(1.0, -solver.tau / mimiS)
    if m.nlin > 0:
        find_step_lin(solver)
    else:
        solver.alpha_lin = 1.0
        solver.beta_lin = 1.0
    if solver.predict:
        for i in range(m.nlmi):
            G, Gi = solver.G[i], solver.Gi[i]
            solver.Xn[i] = solver.X[i] + solver.alpha[i] * solver.delX[i]
            solver.Sn[i] = solver.S[i] + solver.beta[i] * solver.delS[i]
            deed = solver.D[i][:, None] + solver.D[i][None, :]
            solver.RNT[i] = -(Gi @ solver.delX[i] @ solver.delS[i] @ G
                              + G.T @ solver.delS[i] @ solver.delX[i] @ Gi.T) / deed      # :309
    else:
        solver.yold = solver.y
        bmin = min([*solver.beta, solver.beta_lin])
        amin = min([*solver.alpha, solver.alpha_lin])
        solver.y = solver.y + bmin * solver.dely
        for i in range(m.nlmi):
            Xn = solver.X[i] + amin * solver.delX[i]
            solver.X[i] = (Xn + Xn.T) / 2.0
            Sn = solver.S[i] + bmin * solver.delS[i]
            solver.S[i] = (Sn + Sn.T) / 2.0


def find_step_lin(solver):
    """src/predictor_corrector.jl:329-364."""
    m = solver.model
    solver.delS_lin = solver.Rd_lin - m.C_lin.T @ solver.dely
    if solver.predict:
        solver.delX_lin = -solver.X_lin - solver.X_lin * solver.Si_lin * solver.delS_lin
    else:
        solver.delX_lin = (-solver.X_lin - solver.X_lin * solver.Si_lin * solver.delS_lin
                           + (solver.sigma * solver.mu) * solver.Si_lin + solver.RNT_lin)
    mimiX = float(np.min(solver.delX_lin / solver.X_lin))
    solver.alpha_lin = 0.99 if mimiX > -1e-6 else min(1.0, -solver.tau / mimiX)
    mimiS = float(np.min(solver.delS_lin / solver.S_lin))
    solver.beta_lin = 0.99 if mimiS > -1e-6 else min(1.0, -solver.tau / mimiS)
    if solver.predict:
        solver.Xn_lin = solver.X_lin + solver.alpha_lin * solver.delX_lin
        solver.Sn_lin = solver.S_lin + solver.beta_lin * solver.delS_lin
        solver.RNT_lin = -(solver.delX_lin * solver.delS_lin) * solver.Si_lin
    else:
        amin = min([*solver.alpha, solver.alpha_lin])
        bmin = min([*solver.beta, solver.beta_lin])
        solver.X_lin = solver.X_lin + amin * solver.delX_lin
        solver.S_lin = solver.S_lin + bmin * solver.delS_lin
        solver.S_lin_inv = 1.0 / solver.S_lin


def myIPstep(solver, halpha):
    """src/Solvers.jl:448-478."""
    solver.iter += 1
    if solver.iter > solver.maxit:
        solver.status = 4
    solver.cg_iter_pre = 0
    solver.cg_iter_cor = 0
    find_mu(solver)
    t0 = time.perf_counter()
    prepare_W(solver)
    solver.t_prepw = time.perf_counter() - t0
    predictor(solver, halpha)
    if solver.status in (2, 3):
        return
    sigma_update(solver)
    corrector(solver, halpha)


def check_convergence(solver):
    """src/Solvers.jl:496-568 (norm(M,2) of a matrix is Frobenius in Julia)."""
    m = solver.model
    nb = float(np.linalg.norm(m.b))
    by = float(m.b @ solver.y)
    solver.err1 = float(np.linalg.norm(solver.Rp)) / (1.0 + nb)
    e2 = e3 = e4 = e6 = 0.0
    CX = 0.0
    for i in range(m.nlmi):
        nC = float(sp.linalg.norm(m.C[i]))
        e2 += max(0.0, -_eigmin(solver.X[i]) / (1.0 + nb))
        e3 += float(np.linalg.norm(solver.Rd[i])) / (1.0 + nC)
        e4 += max(0.0, -_eigmin(solver.S[i]) / (1.0 + nC))
        CXi = float(m.C[i].multiply(solver.X[i]).sum())
        CX += CXi
        e6 += float(np.sum(solver.S[i] * solver.X[i])) / (1.0 + abs(CXi) + abs(by))
    e5 = (CX - by) / (1.0 + abs(CX) + abs(by))
    if m.nlin > 0:
        nd = float(np.linalg.norm(m.d_lin))
        dX = float(m.d_lin @ solver.X_lin)
        e2 += max(0.0, -float(np.min(solver.X_lin)) / (1.0 + nb))
        e3 += float(np.linalg.norm(solver.Rd_lin)) / (1.0 + nd)
        e4 += max(0.0, -float(np.min(solver.S_lin)) / (1.0 + nd))
        e5 = (CX + dX - by) / (1.0 + abs(CX) + abs(by))
        e6 += float(solver.S_lin @ solver.X_lin) / (1.0 + abs(dX) + abs(by))
    solver.err2, solver.err3, solver.err4, solver.err5, solver.err6 = e2, e3, e4, e5, e6
    if m.nlmi > 0:
        DIMACS_error = solver.err1 + e2 + e3 + e4 + abs(e5) + e6
    else:
        DIMACS_error = e2 + e3 + e4 + abs(e5) + e6
    solver.DIMACS_error = DIMACS_error
    solver.primal_obj = -by + m.b_const
    solver.dual_obj = -CX - (float(m.d_lin @ solver.X_lin) if m.nlin > 0 else 0.0)
    if solver.verb > 0 and solver.status == 0:
        print("%3d %16.8e %9.2e" % (solver.iter, solver.primal_obj, DIMACS_error))
    if DIMACS_error < solver.eDIMACS:
        solver.status = 1
    if DIMACS_error > 1e55:
        solver.status = 2
    elif abs(by) > 1e55:
        solver.status = 3


def solve(solver, halpha=None):
    """src/Solvers.jl:304-361."""
    if halpha is None:
        halpha = Halpha(solver.kit)
    t1 = time.perf_counter()
    setup_solver(solver, halpha)
    initial_point(solver)
    while solver.status == 0:
        t2 = time.perf_counter()
        myIPstep(solver, halpha)
        solver.itertime = time.perf_counter() - t2
        solver.tol_cg = max(solver.tol_cg * solver.tol_cg_up, solver.tol_cg_min)
        if solver.status in (2, 3):
            break
        check_convergence(solver)
        solver.trace.append(dict(
            iter=solver.iter, primal_obj=solver.primal_obj, dual_obj=solver.dual_obj,
            dimacs=solver.DIMACS_error,
            errs=(solver.err1, solver.err2, solver.err3, solver.err4, solver.err5, solver.err6),
            mu=solver.mu, sigma=solver.sigma, cg_pre=solver.cg_iter_pre, cg_cor=solver.cg_iter_cor,
            regcount=solver.regcount, reg_adds=getattr(solver, "reg_adds", 0),
            t_prepw=getattr(solver, "t_prepw", 0.0), t_assembly=getattr(solver, "t_assembly", 0.0),
            t_solve=getattr(solver, "t_solve", 0.0), itertime=solver.itertime))
        if solver.preconditioner == 4:
            if ((solver.cg_iter_cor / 2 > solver.erank * solver.model.nlmi * math.sqrt(solver.model.n) / 20
                 and solver.iter > math.sqrt(solver.model.n) / 60) or solver.cg_iter_cor > 100):
                solver.preconditioner = 1
                solver.aamat = 2
    solver.tottime = time.perf_counter() - t1
    return solver


def objective_value(solver, max_sense=False):
    """src/MOI_wrapper.jl:315-319."""
    val = float(solver.model.b @ solver.y) - solver.model.b_const
    return val if max_sense else -val


def dual_objective_value(solver, max_sense=False):
    """src/MOI_wrapper.jl:321-327."""
    m = solver.model
    val = btrace(m.nlmi, m.C, solver.X) - m.b_const
    if m.nlin > 0:
        val += float(m.d_lin @ solver.X_lin)
    return val if max_sense else -val
