"""One GPU plays rank r of `world` in turn on the C5 instance (msz 10^4, nvar 2*10^4, kit=1): time of the rank's share of
the Schur assembly (the column blocks it owns; nothing is exchanged) and of one application of the CG operator through
the assembled matrix (its column chunks of the triangular mat-vec; the all-reduce of the nvar-vector needs the real
multi-GPU job).  usage: shard_balance_c5.py [msz nvar] [--worlds=1,2,4,8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd
from loraine_jl_amd.synthetic import LowRankProblem
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
msz = int(pos[0]) if len(pos) > 0 else 10000
nvar = int(pos[1]) if len(pos) > 1 else 20000
worlds = [int(w) for w in opts.get("worlds", "1,2,4,8").split(",")]
model = LowRankProblem(msz, nvar, 4).model()
dev = loraine_jl_amd.Device(0)
dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes)
rng = np.random.default_rng(1)
G = rng.standard_normal((msz, 64)) / 8.0
W = np.eye(msz) + G @ G.T
x = rng.standard_normal(nvar)
dev.set_option("profile", 1)
dev.set_option("matvec_h", 2)
dev.set_option("profile_symv", 1)
base = None
for world in worlds:
    ts = []
    for r in range(world):
        if world > 1:
            dev.comm_init_host(r, world, lambda buf, op: None, lambda s, rcv: None)
        dev.set_scaling(0, W)                       # new scaling version: the share is assembled for this rank
        dev.matvec(x)                               # assembles, applies once (warm-up of the workspaces)
        dev.reset_timing()
        dev.set_scaling(0, W)
        dev.matvec(x)
        t_asm = dev.timing("assemble")
        dev.reset_timing()
        for _ in range(5):
            dev.matvec(x)
        t_mv = dev.timing("hop_symv") / 5
        ts.append((t_asm, t_mv))
        print(f"   world {world} rank {r}: assemble {t_asm:.1f} ms, H x share {t_mv * 1e3:.0f} us (shard_bs {dev.shard_bs()})", flush=True)
        if world > 1:
            dev.comm_destroy()
    a = max(t[0] for t in ts); m = max(t[1] for t in ts)
    if base is None:
        base = (a, m)
    print(f"world {world}: slowest rank assemble {a:.1f} ms ({base[0] / a:.2f}x), H x {m * 1e3:.0f} us ({base[1] / m:.2f}x)", flush=True)
