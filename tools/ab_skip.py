"""A/B on one box: block skipping of GEMM1'/2'/3' (interleaved block ownership + masks) on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loraine_jl_amd
from bench import make_scaling
msz, nvar = 2000, 4000
dev = loraine_jl_amd.Device(0)
dev.synthetic_dense_model(msz, nvar, 20250614)
W, G = make_scaling(msz, 20250615)
dev.set_scaling(0, W, G)
dev.set_option("profile", 1)
import numpy as np
ref = None
for rep in range(3):
    for ns in (1, 0):
        dev.set_option("gemm_no_skip", ns)
        dev.schur_assemble(0)
        dev.reset_timing(); dev.schur_assemble(0)
        print(f"rep {rep} no_skip={ns}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f}", flush=True)
