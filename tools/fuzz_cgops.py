"""Randomised unit-level parity of the kit=1 operators against the CPU oracle: random multi-block models
with linear rows, random positive definite (X, S) per block; compares the NT scaling identities, MyA,
makeRHS, H_beta and H_alpha (erank 1..3, Jacobi and Lanczos setups) applies, and one PCG solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import loraine_jl_amd
from oracle import loraine_oracle as lo

def relerr(a, b): return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))

def spd(rng, m, cond):
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    lam = np.exp(rng.uniform(0, np.log(cond), m))
    M = (Q * lam) @ Q.T
    return 0.5 * (M + M.T)

SPARSE_MV = os.environ.get("FUZZ_SPARSE_MV") is not None    # all-sparse models, pattern-restricted mat-vec forced
dev = loraine_jl_amd.Device(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for s in range(seed0, seed0 + count):
    rng = np.random.default_rng(s)
    nvar = int(rng.integers(4, 70)); nblk = int(rng.integers(1, 4))
    sizes = [int(rng.integers(3, 60)) for _ in range(nblk)]
    A = []
    for m in sizes:
        blk = [sp.csc_matrix((m, m))]
        for k in range(nvar):
            kind = rng.integers(0, 4 if SPARSE_MV else 5)
            if kind == 0 and len(A) > 0: M = np.zeros((m, m))
            elif kind <= 1:
                M = np.zeros((m, m)); i, j = rng.integers(0, m, 2); M[i, j] += 1.3; M[j, i] += 1.3
            else:
                R = rng.standard_normal((m, m)) * (rng.random((m, m)) < (1.0 if kind == 4 else 0.1)); M = R + R.T
            blk.append(sp.csc_matrix(M))
        A.append(blk)
    nlin = int(rng.integers(0, 6))
    C_lin = sp.csr_matrix(rng.standard_normal((nvar, nlin)) * (rng.random((nvar, nlin)) < 0.4)) if nlin else None
    om = lo.make_model(A, np.ones(nvar), 0.0, np.ones(nlin) if nlin else None, C_lin)
    erank = int(rng.integers(1, 4))
    if erank >= min(sizes) - 1: erank = 1
    if min(sizes) <= 2: continue
    so = lo.MySolver(om, dict(kit=1, preconditioner=1, erank=erank, verb=0))
    lo.setup_solver(so, lo.Halpha(1)); lo.initial_point(so)
    dev.upload_model(om.AA, om.sigmaA, om.qA, om.msizes, C_lin=om.C_lin if nlin else None)
    msgs = []
    for i, m in enumerate(sizes):
        so.X[i], so.S[i] = spd(rng, m, 10.0 ** rng.uniform(0, 5)), spd(rng, m, 10.0 ** rng.uniform(0, 5))
        info, out = dev.prepare_w(i, so.X[i], so.S[i])
        if info != 0: msgs.append(f"prepare_w info {info}"); continue
        so.W[i], so.G[i] = out["W"], out["G"]
        e = relerr(out["W"] @ so.S[i] @ out["W"], so.X[i])
        if e > 1e-8: msgs.append(f"W S W != X block {i}: {e:.1e}")
    if nlin:
        so.X_lin, so.S_lin = rng.random(nlin) + 0.1, rng.random(nlin) + 0.1
        so.S_lin_inv = 1.0 / so.S_lin
        dev.set_lin(so.X_lin, so.S_lin_inv)
    x = rng.standard_normal(nvar)
    Ao = lo.MyA(so.W, om.AA, om.nlin, om.C_lin, so.X_lin, so.S_lin_inv)
    ref = np.zeros(nvar); Ao(ref, x)
    if SPARSE_MV: dev.set_option("matvec_sparse", 2)
    e = relerr(dev.matvec(x), ref)
    if e > 1e-11: msgs.append(f"matvec {e:.1e}")
    if SPARSE_MV:
        from loraine_jl_amd._capi import ptr
        acc = np.zeros(nvar)
        for r in range(3):
            dev.set_shard(r, 3); part = np.zeros(nvar)
            dev._chk(dev.lib.lrn_matvec_partial(dev.h, ptr(x), ptr(part)), "lrn_matvec_partial"); acc += part
        dev.set_shard(0, 1); dev.set_option("matvec_sparse", 0)
        e = relerr(acc, ref)
        if e > 1e-11: msgs.append(f"partial mat-vec sum {e:.1e}")
    for prec in (2, 1):
        for eig in (1, 2):
            ha = lo.Halpha(1); so.preconditioner, so.erank = prec, erank
            try:
                if prec == 1:
                    lo.Prec_for_CG_tilS_prep(so, ha); Mo = lo.MyM(om.AA, ha.AAAATtau, ha.Umat, ha.Z, ha.cholS)
                else:
                    lo.Prec_for_CG_beta(so, ha); Mo = lo.MyM_beta(om.AA, ha.AAAATtau)
            except Exception as ex:
                continue                     # the reference formula itself breaks down (PosDefException)
            dev.set_option("prec_eig", eig)
            info = dev.prec_setup(prec, erank, so.aamat)
            dev.set_option("prec_eig", 0)
            if info != 0: msgs.append(f"prec_setup({prec}, eig {eig}) info {info} where the oracle succeeded"); continue
            r2 = np.zeros(nvar); Mo(r2, x)
            e = relerr(dev.prec_apply(x), r2)
            if e > (1e-7 if eig == 1 else 1e-5): msgs.append(f"prec {prec} eig {eig} erank {erank} apply {e:.1e}")
    if msgs:
        bad += 1
        print(f"MISMATCH seed {s}: sizes {sizes} nvar {nvar} nlin {nlin} erank {erank}: " + "; ".join(msgs), flush=True)
    if (s - seed0) % 10 == 9: print(f"... {s - seed0 + 1} models, {bad} with mismatches", flush=True)
print(f"done: {count} models, {bad} with mismatches")
sys.exit(1 if bad else 0)
