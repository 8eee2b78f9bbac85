"""Timing of the three assembly GEMMs on the C4 instance (repeat runs on one box; used for the A/B of the packed
store epilogue of GEMM2': build the two versions, run this once each on the same box)."""
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
for rep in range(4):
    dev.reset_timing(); dev.schur_assemble(0)
    print(f"rep {rep}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f}", flush=True)
