"""NumPy prototype (CPU, uses the oracle): would a BLOCK Lanczos step shorten the launch chain of the step-length rule?

The eigmin searches of lrn_ip_find_step cost their longer chain of dependent launches, 8-10 us per Lanczos step
(DESIGN.md section 8).  A block step -- b vectors per pass over M, one launch -- needs fewer steps for the same
eigenvalue.  This script takes the matrices whose smallest eigenvalue the reference's find_step asks for
(src/predictor_corrector.jl:268-285) from the oracle's maxG11 run (last 8 of the first 6 iterations) and counts the steps
until the residual bound of the smallest Ritz pair passes the library's test (1e-10 x max(|theta|, 1e-4 scale); looks every
16 / 8 / 4 steps for b = 1 / 2 / 4).  Record: profiles/r04_block_lanczos_proto.txt."""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loraine_oracle as lo  # noqa: E402


def block(M, bs, batch, mmax, tol=1e-10, seed=0):
    n = M.shape[0]
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.random((n, bs)) - 0.5)
    Qp = np.zeros((n, bs))
    Bp = np.zeros((bs, bs))
    A, B = [], []
    th = [0.0]
    for j in range(mmax):
        W = M @ Q
        Aj = Q.T @ W
        W -= Q @ Aj + Qp @ Bp.T
        Qn, Bj = np.linalg.qr(W)
        A.append((Aj + Aj.T) / 2)
        B.append(Bj)
        Qp, Q, Bp = Q, Qn, Bj
        m = j + 1
        if m % batch == 0:
            N = m * bs
            T = np.zeros((N, N))
            for i in range(m):
                T[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] = A[i]
                if i < m - 1:
                    T[(i + 1) * bs:(i + 2) * bs, i * bs:(i + 1) * bs] = B[i]
                    T[i * bs:(i + 1) * bs, (i + 1) * bs:(i + 2) * bs] = B[i].T
            th, S = np.linalg.eigh(T)
            res = np.linalg.norm(B[-1] @ S[-bs:, 0])
            if res <= tol * max(abs(th[0]), 1e-4 * max(abs(th[0]), abs(th[-1]))):
                return m, th[0]
    return mmax, th[0]


def main():
    mats = []
    orig = lo._eigmin

    def hook(M):
        mats.append(M.copy())
        return orig(M)
    lo._eigmin = hook
    model = lo.model_from_sdpa(os.path.join(ROOT, "tests", "golden", "maxG11.dat-s"), datarank=-1)
    s = lo.MySolver(model, dict(kit=0, datarank=-1, verb=0, maxit=6))
    lo.solve(s)
    for k, M in enumerate(mats[-8:]):
        ex = sla.eigvalsh(M, subset_by_index=[0, 0])[0]
        if ex > -1e-6:
            continue          # (sign class only: the library stops these runs on its `settled` rule)
        out = [block(M, 1, 16, 600), block(M, 2, 8, 300), block(M, 4, 4, 300)]
        print("matrix %d lambda_min %.6e | steps b=1: %d  b=2: %d (%.2f)  b=4: %d (%.2f) | errors %.1e %.1e %.1e" % (
            k, ex, out[0][0], out[1][0], out[1][0] / out[0][0], out[2][0], out[2][0] / out[0][0],
            abs(out[0][1] - ex), abs(out[1][1] - ex), abs(out[2][1] - ex)))


if __name__ == "__main__":
    main()
