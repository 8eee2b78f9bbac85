"""Multi-GPU layout of the Schur complement (kit=0): one process per GPU, every rank holds the
full static data, assembles the lower-triangle COLUMN BLOCKS it owns (block-cyclic in the
reference's nnz-sorted order, so the triangular work is balanced), and one RCCL all-gather
over xGMI gives every rank the whole matrix before the (replicated) Cholesky.

The exchange buffers are rank-major: rank r's buffer holds its blocks (snake block-cyclic: r, 2P-1-r, 2P+r, ...) each
`bs` columns x nvar rows, zero-padded to `blocks_per_rank` so that all ranks send the same
number of bytes (all_gather_into_tensor needs equal sizes).  The C library implements the
same layout (lrn_schur_export_shard / lrn_schur_import_all); the pure-Python pack/unpack here
are the executable specification used by the CPU (gloo) tests.
"""
import numpy as np

SHARD_BS = 128


def auto_bs(nvar, world):
    """Block width the C library picks when the option "shard_bs" is 0 (update_shard_bs in csrc/model.hip):
    two blocks per rank, rounded up to the 128-column tile."""
    if world <= 1 or nvar <= 0:
        return SHARD_BS
    per = (nvar + 2 * world - 1) // (2 * world)
    return max(SHARD_BS, ((per + 127) // 128) * 128)


def geometry(nvar, world, bs=SHARD_BS):
    nblk = (nvar + bs - 1) // bs
    bpr = (nblk + world - 1) // world
    return nblk, bpr, bpr * bs * nvar


def owner_of_block(blk, world):
    """snake (boustrophedon) block-cyclic: rounds alternate direction so that the long
    (early) columns of the lower triangle are spread evenly -- same map as shard_owner()
    in csrc/lrn_common.h."""
    r = blk % world
    return world - 1 - r if (blk // world) & 1 else r


def global_block(rank, lb, world):
    return lb * world + (world - 1 - rank if lb & 1 else rank)


def owner_of_column(col, world, bs=SHARD_BS):
    return owner_of_block(col // bs, world)


def owned_columns(nvar, rank, world, bs=SHARD_BS):
    return [c for c in range(nvar) if owner_of_column(c, world, bs) == rank]


def pack_shard(H, rank, world, bs=SHARD_BS):
    """H: (nvar x nvar) column-major-addressable array -> flat shard buffer of this rank."""
    nvar = H.shape[0]
    nblk, bpr, size = geometry(nvar, world, bs)
    buf = np.zeros(size)
    for lb in range(bpr):
        gb = global_block(rank, lb, world)
        if gb >= nblk:
            continue
        c0, c1 = gb * bs, min(nvar, (gb + 1) * bs)
        buf[lb * bs * nvar: lb * bs * nvar + (c1 - c0) * nvar] = np.asarray(H[:, c0:c1]).reshape(-1, order="F")
    return buf


def unpack_all(buf_all, nvar, world, bs=SHARD_BS):
    """Inverse of the all-gather of `pack_shard` buffers -> (nvar x nvar) matrix."""
    nblk, bpr, size = geometry(nvar, world, bs)
    H = np.zeros((nvar, nvar), order="F")
    for r in range(world):
        for lb in range(bpr):
            gb = global_block(r, lb, world)
            if gb >= nblk:
                continue
            c0, c1 = gb * bs, min(nvar, (gb + 1) * bs)
            seg = buf_all[r * size + lb * bs * nvar: r * size + lb * bs * nvar + (c1 - c0) * nvar]
            H[:, c0:c1] = np.asarray(seg).reshape(nvar, c1 - c0, order="F")
    return H


# ------------------------------------------------------------------ kit = 0, dense data: column split of the matrix variable
def _packed_off_base(c, S):
    q = c >> 4
    return 16 * (q * S - 8 * q * (q + 1)) + (c - 16 * q) * (S - 16 * (q + 1))


def column_range_cost(msz, nd, c0, c1):
    """`col_range_cost` of csrc/schur.hip: what the columns [c0, c1) of the matrix variable cost a rank on the Cholesky
    path, in ms.  The products run on the trailing blocks with the 128-tile grid anchored at c0: GEMM1' tile column j
    has (ntm - j) tiles of K = M - 128 j, GEMM2' tile (i, j), i >= j, has K = M - 128 i (whole tiles, the last tile
    column may be partly empty); GEMM3' costs exactly the packed length of the range.  Constants: least-squares fit to
    the per-rank times of the C4 instance replayed on one GPU (tools/shard_balance.py, profiles/r02_shard_balance.txt):
    a fixed equivalent of ~220 K per tile in GEMM1'/GEMM2', GEMM3' linear in the packed length."""
    S = (msz + 15) // 16 * 16
    M = float(msz - c0)
    ntm, ntn = (msz - c0 + 127) // 128, (c1 - c0 + 127) // 128
    # the last tile column may be partly empty: its waves skip the 16-column blocks beyond the range (a block per 32
    # columns and wave column), but a K-step of the masked loop has a floor: measured 0.5-0.7 of a full tile column for
    # 32 of 128 columns, 0.9 for 96
    rem = (c1 - c0) - 128 * (ntn - 1)
    last = 1.0 if rem >= 128 else min(1.0, 0.4 + 0.65 * float((rem + 31) // 32) / 4.0)
    k1 = k2 = tiles = 0.0
    for j in range(ntn):
        f = last if j == ntn - 1 else 1.0
        k1 += f * float(ntm - j) * (M - 128.0 * j)
        k2 += f * (float(ntm - j) * M - 128.0 * (0.5 * float(ntm - 1) * ntm - 0.5 * float(j - 1) * j))
        tiles += f * float(ntm - j)
    k3 = 16.0 * (c1 - c0) + float(_packed_off_base(c1, S) - _packed_off_base(c0, S))
    s = float(nd) / 4000.0
    return s * (0.0016774 * (k1 + 219.0 * tiles) + 0.0016283 * (k2 + 228.0 * tiles)) + s * s * 0.00024209 * k3


def column_range(msz, nd, rank, world):
    """Executable specification of `col_runs` (csrc/schur.hip): the columns [c0, c1) of the matrix variable a rank owns
    on the Cholesky path, or None for a rank left idle (more ranks than 16-column units).  With W = L L', column c of
    At_k = L' A_k L needs only columns >= c of L and A_k and <At_i, At_j> is a sum over columns, so a rank computes ITS
    columns of every At_k and its share of every inner product; the ranks' partial Schur matrices are added by one
    all-reduce.  Every rank gets one contiguous range with ends at multiples of 16 (the block width of the packed
    layout), chosen by a dynamic programme over the 16-column units to minimise the largest `column_range_cost`
    (ties keep the smallest cut)."""
    if world <= 1:
        return (0, msz)
    nu = (msz + 15) // 16

    def col(u):
        return min(msz, 16 * u)

    P = min(world, nu)
    INF = 1e300
    dp = [[INF] * (nu + 1) for _ in range(P + 1)]
    cut = [[0] * (nu + 1) for _ in range(P + 1)]
    dp[0][0] = 0.0
    for p in range(1, P + 1):
        for j in range(p, nu + 1):
            for i in range(p - 1, j):
                if dp[p - 1][i] >= dp[p][j]:
                    continue
                seg = column_range_cost(msz, nd, col(i), col(j))
                v = dp[p - 1][i] if dp[p - 1][i] > seg else seg
                if v < dp[p][j]:
                    dp[p][j], cut[p][j] = v, i
    lo, hi, j = [0] * P, [0] * P, nu
    for p in range(P, 0, -1):
        lo[p - 1], hi[p - 1] = cut[p][j], j
        j = cut[p][j]
    return (col(lo[rank]), col(hi[rank])) if rank < P else None


def agree_on_plan(dev, group=None, mode=0):
    """(Executable specification of `comm_agree_plan` in csrc/comm.hip, which the product path runs inside
    lrn_schur_assemble; kept for the CPU tests over gloo.)
    Every rank must enter the same collective after the assembly.  The choice between the two exchanges depends on
    free device memory, which differs between ranks: all-reduce (MIN) each rank's own `lrn_schur_plan` and pin the
    result on every rank (option "schur_plan").  Returns the agreed plan (1 all-reduce of partial sums, 0 all-gather
    of column blocks)."""
    import torch
    import torch.distributed as dist
    mine = dev.schur_plan(mode)
    on_gpu = dist.get_backend(group) != "gloo"
    t = torch.tensor([mine], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    plan = int(t.item())
    dev.set_option("schur_plan", plan)
    return plan


def check_same_exchange(dev, group=None):
    """(Specification of the status reduction in `comm_schur_exchange`, csrc/comm.hip -- there a rank whose assembly
    FAILED enters it too, and every rank then returns the error.)
    After the assembly: all ranks hold the same kind of result (partial sums or column blocks)?  One tiny
    all-reduce that every rank enters unconditionally; a disagreement (a rank whose W could not be factored, a
    plan that was not pinned) raises on every rank instead of pairing an all-reduce with an all-gather."""
    import torch
    import torch.distributed as dist
    f = 1 if dev.schur_is_partial_sum() else 0
    on_gpu = dist.get_backend(group) != "gloo"
    t = torch.tensor([f, -f], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0].item()), -int(t[1].item())
    if hi != lo:
        raise RuntimeError("ranks disagree on the Schur exchange (partial sums on some, column blocks on others): "
                           "refusing to enter mismatched collectives")
    return bool(hi)


# ------------------------------------------------------------------ kit = 1: sharded PCG
def pcg_allreduce(matvec_partial, allreduce_sum, precon, b, tol, max_iter, xp=np):
    """`cg(A, b; tol, maxIter, precon)` (ConjugateGradients.jl 0.1, call sites
    src/predictor_corrector.jl:134,235) with the operator applied as
    `allreduce_sum(matvec_partial(p))`: every rank holds all vectors and runs the same
    iteration; one all-reduce of an nvar-vector per mat-vec is the only exchange.
    Returns (x, exit_code, iterations)."""
    x = xp.zeros_like(b)
    bn = float(xp.linalg.norm(b))
    if bn == 0.0:
        return x, 1, 0
    r = b.clone() if hasattr(b, "clone") else b.copy()
    res0 = bn
    if res0 <= tol:
        return x, 2, 0
    z = precon(r)
    p = z.clone() if hasattr(z, "clone") else z.copy()
    for it in range(1, max_iter + 1):
        Ap = allreduce_sum(matvec_partial(p))
        gamma = float(r @ z)
        pAp = float(p @ Ap)
        alpha = gamma / pAp if pAp != 0.0 else float("inf")
        if alpha == float("inf") or alpha < 0 or alpha != alpha:
            return x, -13, it
        x += alpha * p
        r -= alpha * Ap
        if float(xp.linalg.norm(r)) / res0 <= tol:
            return x, 30, it
        z = precon(r)
        beta = float(z @ r) / gamma
        p = z + beta * p
    return x, -2, max_iter


class DistributedHotPath:
    """Attach to a solver (`solvers.MySolver` / `resident.ResidentSolver`) to run the interior-point loop with one
    process per GPU: every rank drives the same (replicated, deterministic) iteration and only the hot path is sharded
    (SURVEY.md 8e).  All this class does is create the library's communicator (csrc/comm.hip: RCCL, or host callbacks
    over a gloo group): from then on `lrn_schur_assemble` agrees on the path, checks every rank's outcome and exchanges
    on the library's stream (kit=0), and `lrn_pcg` all-reduces its nvar-vector (kit=1) -- the loop itself is unchanged,
    which is what a Julia host gets from `LoraineHIP.comm_init!` as well.  From matrix side 4096 on the n^3 products of
    prepare_W / find_step / the right-hand sides are computed as column blocks by the ranks and all-gathered inside the
    library (option "shard_products"); below that size, the Cholesky factorisations, the Lanczos runs and the
    preconditioner setup are replicas."""

    def __init__(self, solver, rank, world, group=None):
        self.rank, self.world, self.group = int(rank), int(world), group
        solver.dist = self
        # the C library shards Schur columns in sigma-position space, which exists for one LMI block;
        # multi-block problems assemble replicated (every rank the whole matrix, no exchange)
        self.shard_schur = self.world > 1 and (solver.kit == 1 or solver.model.nlmi == 1)
        if self.shard_schur:
            self.transport = solver.dev.comm_init_torch(self.rank, self.world, group)
        else:
            solver.dev.set_shard(0, 1)

    def allgather(self, dev):
        """(The exchange happened inside lrn_schur_assemble.)"""

    def pcg(self, dev, h, tol, max_iter=10000):
        return dev.pcg(h, tol, max_iter)
