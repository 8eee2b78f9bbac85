import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
dev = loraine_jl_amd.Device(0)
for n, dec in [(90, 6), (300, 6), (300, 10), (1000, 8), (2000, 6)]:
    rng = np.random.default_rng(n)
    U0, _ = np.linalg.qr(rng.standard_normal((n, n))); V0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    sv = np.logspace(0, -dec, n)
    A = (U0 * sv) @ V0.T
    US, s, V, sw = dev.dbg_svd_jacobi(A)
    U = US / s[None, :]
    so = np.sort(s)[::-1]
    print(f"n={n} cond=1e{dec} sweeps={sw} max|U'U-I|={np.abs(U.T@U-np.eye(n)).max():.2e} max|V'V-I|={np.abs(V.T@V-np.eye(n)).max():.2e} "
          f"sv relerr vs truth={np.max(np.abs(so-sv)/sv):.2e}  vs LAPACK={np.max(np.abs(np.linalg.svd(A,compute_uv=False)-sv)/sv):.2e}")
