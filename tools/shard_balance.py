"""One GPU plays rank r of `world` in turn: per-rank assembly time of the C4 instance (load balance and
per-rank efficiency of the sharding; the exchange itself needs the real multi-GPU job).  Two layouts:
  split   the default for dense data: columns of the matrix variable, partial sums + all-reduce
  blocks  north_star's wording: Schur column blocks (T_k = L (L'A_kL) L' for the owned k) + all-gather (schur_chol = 2)
usage: shard_balance.py [msz nvar] [--layouts split,blocks] [--worlds 1,2,4,8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import loraine_jl_amd
from loraine_jl_amd import sharding
from bench import make_scaling
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
msz = int(pos[0]) if len(pos) > 0 else 2000
nvar = int(pos[1]) if len(pos) > 1 else 4000
layouts = opts.get("layouts", "split,blocks").split(",")
worlds = [int(w) for w in opts.get("worlds", "1,2,4,8").split(",")]
dev = loraine_jl_amd.Device(0)
dev.synthetic_dense_model(msz, nvar, 20250614)
W, G = make_scaling(msz, 20250615)
dev.set_scaling(0, W, G)
dev.set_option("profile", 1)
base = None
for layout in layouts:
    dev.set_option("schur_chol", 2 if layout == "blocks" else -1)
    for world in worlds:
        if layout == "blocks" and world == 1:
            continue
        ts = []
        for r in range(world):
            dev.set_shard(r, world)
            if world > 1:
                dev.set_option("schur_plan", 0 if layout == "blocks" else 1)
            dev.schur_assemble(0)                      # warm-up (workspace sizes)
            dev.reset_timing(); dev.schur_assemble(0)
            ts.append(dev.timing("assemble"))
            cols = sharding.column_range(msz, nvar, r, world) if layout == "split" else None
            print(f"   [{layout}] world {world} rank {r}: assemble {ts[-1]:.1f}  wchol {dev.timing('wchol'):.1f} gemm1 {dev.timing('gemm1'):.1f} "
                  f"gemm2 {dev.timing('gemm2'):.1f} gemm3 {dev.timing('gemm3'):.1f} reduce {dev.timing('reduce3'):.1f}  "
                  f"launches {dev.count('gemm1')}/{dev.count('gemm2')}/{dev.count('gemm3')}  columns {cols}", flush=True)
        if world == 1:
            base = ts[0]
            print(f"[{layout}] world 1: {ts[0]:.1f} ms", flush=True)
        else:
            print(f"[{layout}] world {world}: per-rank assembly ms min {min(ts):.1f} max {max(ts):.1f} mean {sum(ts)/world:.1f}"
                  + (f"  (1-GPU {base:.1f} / world = {base/world:.1f}; slowest rank = {base/max(ts):.2f}x)" if base else ""), flush=True)
dev.set_option("schur_plan", -1)
dev.set_option("schur_chol", -1)
dev.set_shard(0, 1)
