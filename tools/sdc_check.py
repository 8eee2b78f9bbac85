import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, loraine_jl_amd
dev = loraine_jl_amd.Device(0)
rng = np.random.default_rng(0)
for n, leaf in [(600, 128), (1500, 256), (3000, 768)]:
    dev.set_option("sdc_leaf", leaf)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(rng.uniform(np.log(1.0), np.log(50.0), n))       # cond 50, like a centred iterate
    K = (Q * lam) @ Q.T; K = 0.5 * (K + K.T)
    t = time.perf_counter(); V = dev.dbg_sdc(K); dt = time.perf_counter() - t
    G = V.T @ K @ V
    off = np.linalg.norm(G - np.diag(np.diag(G))) / np.linalg.norm(np.diag(G))
    print(f"n={n} leaf={leaf} time {dt*1e3:.1f} ms  orth {np.abs(V.T@V-np.eye(n)).max():.2e}  rel off {off:.2e}  splits {dev.count('sdc_splits')} leaves {dev.count('sdc_leaves')} qdwh its {dev.count('sdc_qdwh_its')} fallbacks {dev.count('sdc_fallbacks')}", flush=True)
