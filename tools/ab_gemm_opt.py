"""A/B on one box, C4 instance, of one integer option of the assembly GEMMs: OPT=<name> VALUES=a,b,...  The Schur matrices
of all values must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from bench import make_scaling
opt = os.environ["OPT"]
values = [int(v) for v in os.environ.get("VALUES", "0,1").split(",")]
msz, nvar = 2000, 4000
dev = loraine_jl_amd.Device(0)
dev.synthetic_dense_model(msz, nvar, 20250614)
W, G = make_scaling(msz, 20250615)
dev.set_scaling(0, W, G)
dev.set_option("profile", 1)
for rep in range(3):
    for v in values:
        dev.set_option(opt, v)
        dev.schur_assemble(0)
        dev.reset_timing(); dev.schur_assemble(0)
        print(f"rep {rep} {opt} {v}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} "
              f"gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} + {dev.timing('gemm3s'):.1f}", flush=True)
Hs = []
for v in values:
    dev.set_option(opt, v); Hs.append(np.tril(dev.schur_assemble(0, want_H=True)))
print("identical:", all(np.array_equal(Hs[0], H) for H in Hs[1:]))
