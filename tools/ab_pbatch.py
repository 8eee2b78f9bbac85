"""C4 assembly against the number of constraint matrices per GEMM1'/GEMM2' launch (option p_batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loraine_jl_amd
from bench import make_scaling
msz, nvar = 2000, 4000
for pb in [int(x) for x in sys.argv[1:]] or [0]:
    dev = loraine_jl_amd.Device(0)
    dev.synthetic_dense_model(msz, nvar, 20250614)
    W, G = make_scaling(msz, 20250615)
    dev.set_scaling(0, W, G)
    dev.set_option("profile", 1)
    if pb:
        dev.set_option("p_batch", pb)
    dev.schur_assemble(0)
    for rep in range(2):
        dev.reset_timing(); dev.schur_assemble(0)
        print(f"p_batch {pb} rep {rep}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} ({dev.count('gemm1')}) "
              f"gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} + {dev.timing('gemm3s'):.1f}", flush=True)
    dev.close()
