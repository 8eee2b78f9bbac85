import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import loraine_jl_amd
dev = loraine_jl_amd.Device(0)
rng = np.random.default_rng(1)
for n in (1601, 1537, 2001):
    A = np.asfortranarray(rng.standard_normal((n, n))); B = np.asfortranarray(rng.standard_normal((n, n)))
    C = dev.dbg_gemm(A, B, False, True)          # C = A B'  (both operands m-contiguous)
    R = A @ B.T
    print(n, "rel err", np.linalg.norm(C - R) / np.linalg.norm(R), flush=True)
