"""A/B on one box: GEMM3' with a row of edge tiles (gemm3_strip 0) vs a last tile row of height 160 on a second stream
(gemm3_strip 1), C4 instance; optional split-K factors."""
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
splits = [int(x) for x in sys.argv[1:]] or [0]
strips = [int(x) for x in os.environ.get('AB_STRIPS', '0,1').split(',')]
for rep in range(2):
    for ks in splits:
        dev.set_option("gemm3_ksplit", ks)
        for strip in strips:
            dev.set_option("gemm3_strip", strip)
            dev.schur_assemble(0)
            dev.reset_timing(); dev.schur_assemble(0)
            print(f"rep {rep} ksplit {ks} strip {strip}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} "
                  f"gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f}", flush=True)
dev.set_option("gemm3_ksplit", 0)
dev.set_option("gemm3_strip", 0); H1 = dev.schur_assemble(0, want_H=True)
dev.set_option("gemm3_strip", 1); H2 = dev.schur_assemble(0, want_H=True)
print("identical:", np.array_equal(np.tril(H1), np.tril(H2)))
