"""One GPU plays rank r of `world` in turn: per-rank assembly time of the C4 instance (load balance and
per-rank efficiency of the column-block sharding; the exchange itself needs the real multi-GPU job)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import loraine_jl_amd
from bench import make_scaling
msz = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nvar = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = loraine_jl_amd.Device(0)
dev.synthetic_dense_model(msz, nvar, 20250614)
W, G = make_scaling(msz, 20250615)
dev.set_scaling(0, W, G)
dev.set_option("profile", 1)
if len(sys.argv) > 3:
    dev.set_option("shard_bs", int(sys.argv[3]))
for world in (1, 2, 4, 8):
    ts = []
    for r in range(world):
        dev.set_shard(r, world)
        bs_now = dev.shard_bs()
        dev.schur_assemble(0)                      # warm-up (workspace sizes)
        dev.reset_timing(); dev.schur_assemble(0)
        ts.append(dev.timing("assemble"))
        if world in (4, 8):
            print(f"   rank {r}: gemm1 {dev.timing('gemm1'):.1f} gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} launches {dev.count('gemm1')}/{dev.count('gemm2')}/{dev.count('gemm3')}", flush=True)
    print(f"world {world} (shard_bs {bs_now}): per-rank assembly ms min {min(ts):.1f} max {max(ts):.1f}  ideal {ts and (sum(ts)/world):.1f}  "
          f"(1-GPU time / world = {base/world:.1f})" if world > 1 else f"world 1: {ts[0]:.1f} ms", flush=True)
    if world == 1:
        base = ts[0]
dev.set_shard(0, 1)
