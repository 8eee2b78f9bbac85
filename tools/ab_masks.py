"""A/B on one box: GEMM1'/GEMM2' with the masked K-steps as straight-line pattern loops (default) vs a branch per block
(option gemm_dyn_masks 1, the round-2 kernel), C4 instance."""
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
    for dyn in (1, 0):
        dev.set_option("gemm_dyn_masks", dyn)
        dev.schur_assemble(0)
        dev.reset_timing(); dev.schur_assemble(0)
        print(f"rep {rep} dyn_masks {dyn}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} "
              f"gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} + {dev.timing('gemm3s'):.1f}", flush=True)
dev.set_option("gemm_dyn_masks", 1); H1 = dev.schur_assemble(0, want_H=True)
dev.set_option("gemm_dyn_masks", 0); H2 = dev.schur_assemble(0, want_H=True)
print("identical:", np.array_equal(np.tril(H1), np.tril(H2)), "rel diff", np.linalg.norm(np.tril(H1 - H2)) / np.linalg.norm(np.tril(H1)))
