"""Time of plain NT products C = A B' (the form of every product of the NT scaling / Lyapunov / step-length code) by matrix
side, back to back on the device (LRN_DBG_GEMM_REPS).  LRN_GEMM_MID=0: without the mid-size slab kernel."""
import os, sys
os.environ.setdefault("LRN_DBG_GEMM_REPS", "50")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
dev = loraine_jl_amd.Device(0)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [400, 500, 640, 800, 801, 1000, 1200, 1280, 1400, 1499, 1536, 2000]:
    A = np.asfortranarray(rng.standard_normal((n, n))); B = np.asfortranarray(rng.standard_normal((n, n)))
    C = dev.dbg_gemm(A, B, False, True)
    print(n, "rel err %.1e" % (np.linalg.norm(C - A @ B.T) / np.linalg.norm(A @ B.T)), flush=True)
