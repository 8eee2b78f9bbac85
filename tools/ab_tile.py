"""A/B on one box: GEMM3' workgroup tile 128 vs 160 (option gemm3_tile) on the C4 instance."""
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
ref = None
for rep in range(2):
  for sched in (0, 1):
    dev.set_option("gemm3_sched", sched)
    for tile in (128, 160):
        dev.set_option("gemm3_tile", tile)
        dev.schur_assemble(0)
        dev.reset_timing(); dev.schur_assemble(0)
        print(f"rep {rep} sched {sched} tile {tile}: assemble {dev.timing('assemble'):.1f} gemm1 {dev.timing('gemm1'):.1f} gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} ({dev.count('gemm3')} launches)", flush=True)
# same matrix?
dev.set_option("gemm3_tile", 128); H1 = dev.schur_assemble(0, want_H=True)
dev.set_option("gemm3_tile", 160); H2 = dev.schur_assemble(0, want_H=True)
print("rel diff 128 vs 160:", np.linalg.norm(np.tril(H1 - H2)) / np.linalg.norm(np.tril(H1)))
