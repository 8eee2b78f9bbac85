"""A/B on one box, C4 instance: GEMM1' with the blocks above the block diagonal of its diagonal tiles left out
(option gemm1_diag 1, default -- GEMM2' never reads them) vs computed (0).  The Schur matrices must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from bench import make_scaling
msz, nvar = 2000, 4000
dev = loraine_jl_amd.Device(0)
dev.synthetic_dense_model(msz, nvar, 20250614)
W, G = make_scaling(msz, 20250615)
dev.set_scaling(0, W, G)
dev.set_option("profile", 1)
for rep in range(3):
    for on in (0, 1):
        dev.set_option("gemm1_diag", on)
        dev.schur_assemble(0)
        dev.reset_timing(); dev.schur_assemble(0)
        print(f"rep {rep} gemm1_diag {on}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} "
              f"gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} + {dev.timing('gemm3s'):.1f}", flush=True)
dev.set_option("gemm1_diag", 0); H1 = dev.schur_assemble(0, want_H=True)
dev.set_option("gemm1_diag", 1); H2 = dev.schur_assemble(0, want_H=True)
print("identical:", np.array_equal(np.tril(H1), np.tril(H2)), "rel diff", np.linalg.norm(np.tril(H1 - H2)) / np.linalg.norm(np.tril(H1)))
