"""Static problem data in the reference's layout (`MyModel`, reference src/model.jl:34-87) and
its construction: SDPA sparse files -> MOI-equivalent sign mapping (src/MOI_wrapper.jl:142-232)
-> `_prepare_A` (src/model.jl:120-229).  Host-side, one-time; the arrays built here are what
`Device.upload_model` hands to the GPU library.

    max  b'y - b_const   s.t.  sum_j y_j A_ij <= C_i  (i = 1..nlmi),   C_lin' y <= d_lin
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import scipy.sparse as sp


@dataclass
class MyModel:
    A: List[list]           # A[i][k], k = 0..n : F_0, F_1..F_n of block i (csc, both triangles)
    AA: List[sp.csr_matrix]  # AA[i] (n x m_i^2), row j = -vec(A[i][j+1])
    B: List[sp.csr_matrix]   # rank-one factors (datarank = -1) or []
    C: List[sp.csc_matrix]   # C[i] = -A[i][0]
    nzA: np.ndarray
    sigmaA: np.ndarray       # 0-based here; the C ABI receives it 1-based
    qA: np.ndarray
    b: np.ndarray
    b_const: float
    d_lin: np.ndarray
    C_lin: sp.csr_matrix
    n: int
    msizes: np.ndarray
    nlin: int
    nlmi: int


def _rank_one_rows(blockA, n):
    """prep_B, src/model.jl:176-197: A_k = b_k b_k' on the support of A_k, or an error."""
    m = blockA[0].shape[0]
    out = sp.lil_matrix((n, m))
    for k in range(n):
        Ak = blockA[k + 1].tocsc()
        if Ak.nnz == 0:
            continue
        support = list(dict.fromkeys(Ak.indices.tolist()))
        sub = Ak[support, :][:, support].toarray()
        _, vecs = np.linalg.eigh(0.5 * (sub + sub.T))
        with np.errstate(invalid="ignore"):
            bk = np.sign(vecs[:, -1]) * np.sqrt(np.diag(sub))
        miss = np.linalg.norm(sub - np.outer(bk, bk))
        if not miss <= 5.0e-6:
            raise ValueError(f"Obtained an error of `{miss} > 5e-6` when converting matrix into rank `1`, "
                             "use `datarank = 0` to disable the rank-1 conversion.")
        out[k, support] = bk
    return out.tocsr()


def _prepare_A(A, datarank, kappa, n):
    """src/model.jl:120-150: AA (prep_AA!), B (prep_B), C, nzA, sigmaA, qA (prep_sparse!)."""
    nlmi = len(A)
    AA, B, C = [], [], []
    nzA = np.zeros((n, nlmi), dtype=np.int64)
    sigmaA = np.zeros((n, nlmi), dtype=np.int64)
    qA = np.zeros((2, nlmi), dtype=np.int64)
    for i, blk in enumerate(A):
        m = blk[0].shape[0]
        C.append(sp.csc_matrix(-blk[0]))
        rr, cc, vv = [], [], []
        for j in range(n):
            co = blk[j + 1].tocoo()
            nzA[j, i] = co.nnz
            rr.append(np.full(co.nnz, j, dtype=np.int64))
            cc.append(co.col.astype(np.int64) * m + co.row)
            vv.append(-co.data)
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        AA.append(sp.csr_matrix((cat(vv, float), (cat(rr, np.int64), cat(cc, np.int64))), shape=(n, m * m)))
        if datarank == -1:
            B.append(_rank_one_rows(blk, n))
        order = np.argsort(-nzA[:, i], kind="stable")       # sortperm(rev=true) is stable
        sigmaA[:, i] = order
        below = np.nonzero(nzA[order, i] <= kappa)[0]
        qA[:, i] = below[0] if below.size else n
    return AA, B, C, nzA, sigmaA, qA


def build_model(A, b, b_const=0.0, d_lin=None, C_lin=None, datarank=0, kappa=8) -> MyModel:
    n = len(b)
    for blk in A:
        for k in range(len(blk)):
            blk[k] = sp.csc_matrix(blk[k])
            blk[k].eliminate_zeros()
    AA, B, C, nzA, sigmaA, qA = _prepare_A(A, datarank, kappa, n)
    if C_lin is None or C_lin.shape[1] == 0:
        C_lin = sp.csr_matrix((n, 0))
        d_lin = np.zeros(0)
    msizes = np.array([blk[0].shape[0] for blk in A], dtype=np.int64)
    return MyModel(A, AA, B, C, nzA, sigmaA, qA, np.asarray(b, float), float(b_const), np.asarray(d_lin, float),
                   sp.csr_matrix(C_lin), n, msizes, int(C_lin.shape[1]), len(A))


def _tokens(line):
    for ch in "{}(),":
        line = line.replace(ch, " ")
    return line.split()


def read_sdpa(path):
    """SDPA sparse format: nvar / nblocks / block sizes / c / (mat blk i j val) lines."""
    rows = [ln.strip() for ln in open(path)]
    rows = [ln for ln in rows if ln and ln[0] not in '*"']
    nvar = int(_tokens(rows[0])[0])
    nblk = int(_tokens(rows[1])[0])
    sizes = [int(float(t)) for t in _tokens(rows[2])[:nblk]]
    c, at = [], 3
    while len(c) < nvar:
        c += [float(t) for t in _tokens(rows[at])]
        at += 1
    ent = []
    for ln in rows[at:]:
        t = _tokens(ln)
        if len(t) >= 5:
            ent.append((int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])))
    return nvar, sizes, np.array(c[:nvar]), ent


def model_from_sdpa(path, datarank=0, kappa=8) -> MyModel:
    """min c'x, sum F_k x_k - F_0 >= 0  ->  A[lmi][0] = F_0, A[lmi][k] = F_k, b = -c;
    diagonal blocks (negative size) -> rows of C_lin = -coef', d_lin = -F_0[ii]
    (src/MOI_wrapper.jl:145-149,179-217)."""
    nvar, sizes, c, ent = read_sdpa(path)
    psd = [k for k, s in enumerate(sizes) if s > 0]
    lmi_id = {blk: i for i, blk in enumerate(psd)}
    lin_base, nlin = {}, 0
    for k, s in enumerate(sizes):
        if s < 0:
            lin_base[k] = nlin
            nlin += -s
    trip = [[([], [], []) for _ in range(nvar + 1)] for _ in psd]
    lr, lc, lv = [], [], []
    d_lin = np.zeros(nlin)
    for mat, blk, i, j, v in ent:
        if v == 0.0:
            continue
        blk -= 1
        if sizes[blk] > 0:
            I, J, V = trip[lmi_id[blk]][mat]
            I.append(i - 1); J.append(j - 1); V.append(v)
            if i != j:
                I.append(j - 1); J.append(i - 1); V.append(v)
        else:
            r = lin_base[blk] + i - 1
            if mat == 0:
                d_lin[r] -= v
            else:
                lr.append(mat - 1); lc.append(r); lv.append(-v)
    A = [[sp.csc_matrix((V, (I, J)), shape=(sizes[blk], sizes[blk])) for (I, J, V) in trip[i]]
         for i, blk in enumerate(psd)]
    C_lin = sp.csr_matrix((lv, (lr, lc)), shape=(nvar, nlin))
    return build_model(A, -c, 0.0, d_lin, C_lin, datarank, kappa)
