"""Per-tile overhead of GEMM1' / GEMM2' (direct-to-LDS 128-tile kernel, triangular K ranges): the same kernels at matrix
sides without edge tiles (multiples of 128) and with every block computed (gemm_no_skip 1) or skipped as in production --
time = b x (tile K-steps) + a x tiles, fitted across the sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from bench import make_scaling
dev = loraine_jl_amd.Device(0)
dev.set_option("profile", 1)
dev.set_option("schur_chol", 1)

rows = []
for msz, nvar in ((1024, 2000), (2048, 1000), (4096, 250), (2000, 1000)):
    dev.synthetic_dense_model(msz, nvar, 20250614)
    W, G = make_scaling(msz, 20250615)
    dev.set_scaling(0, W, G)
    nt = (msz + 127) // 128
    tiles = nt * (nt + 1) // 2
    ks = lambda x: -(-x // 16)
    k1 = sum((nt - tn) * ks(msz - 128 * tn) for tn in range(nt))
    k2 = sum((tm + 1) * ks(msz - 128 * tm) for tm in range(nt))
    for ns in (1, 0):
        dev.set_option("gemm_no_skip", ns)
        dev.schur_assemble(0)
        t1 = t2 = 0.0
        for rep in range(2):
            dev.reset_timing(); dev.schur_assemble(0)
            t1 += dev.timing("gemm1") / 2; t2 += dev.timing("gemm2") / 2
        print(f"msz {msz} nvar {nvar} no_skip {ns}: tiles {tiles} ksteps1 {k1} ksteps2 {k2}  gemm1 {t1:.2f} ms = {t1 / nvar * 1e3:.2f} us/matrix "
              f"({t1 / nvar / k1 * 1e6:.3f} ns/kstep)  gemm2 {t2:.2f} ms = {t2 / nvar * 1e3:.2f} us/matrix ({t2 / nvar / k2 * 1e6:.3f} ns/kstep)", flush=True)
        rows.append((msz, ns, tiles, k1, k2, t1 / nvar * 1e3, t2 / nvar * 1e3))
    dev.set_option("gemm_no_skip", 0)
# fit on the no_skip rows (every K-step of every tile is a full one): t = b * ksteps + a * tiles
for ns in (1, 0):
    A = []; y = []
    for msz, n, tiles, k1, k2, u1, u2 in rows:
        if n != ns or msz % 128:
            continue
        A += [[k1, tiles], [k2, tiles]]; y += [u1, u2]
    (b, a), res, *_ = np.linalg.lstsq(np.array(A, float), np.array(y), rcond=None)
    print(f"no_skip {ns}: b = {b * 1e3:.3f} ns per tile K-step, a = {a * 1e3:.1f} ns per tile = {a / b:.2f} K-steps; residuals",
          np.round(np.array(A, float) @ [b, a] - np.array(y), 2))

# ---- what the per-tile cost is made of (option gemm_lab: bit 0 no epilogue, bit 1 no K loop, bit 2 no first load)
msz, nvar = 2048, 1000
dev.synthetic_dense_model(msz, nvar, 20250614)
W, G = make_scaling(msz, 20250615)
dev.set_scaling(0, W, G)
dev.set_option("gemm_no_skip", 1)
tiles = 136
for lab, what in ((0, "production"), (1, "no epilogue"), (2, "no K loop"), (3, "no K loop, no epilogue"), (7, "empty workgroups"),
                  (6, "epilogue only"), (8, "one workgroup per CU"), (9, "one workgroup per CU, no epilogue"),
                  (16, "every tile the whole K range: 136 x 128 K-steps"), (17, "whole K range, no epilogue"), (24, "whole K range, one workgroup per CU")):
    dev.set_option("gemm_lab", lab)
    dev.schur_assemble(0)
    dev.reset_timing(); dev.schur_assemble(0)
    t1, t2 = dev.timing("gemm1"), dev.timing("gemm2")
    print(f"lab {lab} ({what}): gemm1 {t1:.2f} ms = {t1 / nvar / tiles * 1e6:.1f} ns per tile, gemm2 {t2:.2f} ms = {t2 / nvar / tiles * 1e6:.1f} ns per tile", flush=True)
dev.set_option("gemm_lab", 0)
dev.set_option("gemm_no_skip", 0)
