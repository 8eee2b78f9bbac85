"""Randomised check of the Schur column sharding on one GPU: random single-block models (dense / sparse /
single-entry / empty constraint matrices, optional linear rows, optional rank-one data), random world
sizes and block widths; the shards assembled one rank at a time and glued by the exchange layout must
equal the unsharded assembly bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import loraine_jl_amd
from loraine_jl_amd.model import build_model

dev = loraine_jl_amd.Device(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for s in range(seed0, seed0 + count):
    rng = np.random.default_rng(s)
    m = int(rng.integers(3, 150)); nvar = int(rng.integers(5, 300)); rank1 = rng.random() < 0.25
    blk = [sp.csc_matrix((m, m))]
    for k in range(nvar):
        if rank1:
            v = rng.standard_normal(m) * (rng.random(m) < max(0.1, 2.0 / m)); v[rng.integers(0, m)] += 1.0
            blk.append(sp.csc_matrix(np.outer(v, v))); continue
        kind = rng.integers(0, 5)
        if kind == 0: M = np.zeros((m, m))
        elif kind == 1:
            M = np.zeros((m, m)); i, j = rng.integers(0, m, 2); M[i, j] += 1.5; M[j, i] += 1.5
        else:
            R = rng.standard_normal((m, m)) * (rng.random((m, m)) < (1.0 if kind == 4 else 0.05)); M = R + R.T
        blk.append(sp.csc_matrix(M))
    nlin = int(rng.integers(0, 6)) if not rank1 else 0
    C_lin = sp.csr_matrix(rng.standard_normal((nvar, nlin)) * (rng.random((nvar, nlin)) < 0.3)) if nlin else None
    model = build_model([blk], np.ones(nvar), 0.0, np.ones(nlin) if nlin else None, C_lin, datarank=-1 if rank1 else 0)
    world = int(rng.integers(2, 7)); bs = int(rng.choice([0, 16, 128, 48]))
    dev.set_option("dense_threshold", float(rng.choice([-1, 20])))
    dev.upload_model(model.AA, model.sigmaA, model.qA, model.msizes, B=model.B if rank1 else None,
                     C_lin=model.C_lin if nlin else None)
    dev.set_option("dense_threshold", -1)
    G = rng.standard_normal((m, m)) / np.sqrt(m) + np.eye(m)
    dev.set_scaling(0, G @ G.T, G)
    if nlin: dev.set_lin(rng.random(nlin) + 0.1, rng.random(nlin) + 0.1)
    mode = -1 if rank1 else 0
    dev.set_option("shard_bs", 0); dev.set_shard(0, 1)
    Hf = dev.schur_assemble(mode, want_H=True)
    dev.set_option("shard_bs", bs)
    parts = []
    for r in range(world):
        dev.set_shard(r, world); dev.schur_assemble(mode)
        buf = torch.zeros(dev.shard_doubles(), dtype=torch.float64, device="cuda")
        dev.schur_export_shard(buf); parts.append(buf)
    dev.schur_import_all(torch.cat(parts))
    H2 = dev.schur_get()
    dev.set_shard(0, 1); dev.set_option("shard_bs", 0)
    if not np.array_equal(np.tril(H2), np.tril(Hf)):
        bad += 1
        Hg = dev.schur_assemble(0, want_H=True)          # unsharded general path as the referee
        ng = max(np.linalg.norm(np.tril(Hg)), 1e-300)
        print(f"   unsharded vs general {np.linalg.norm(np.tril(Hf - Hg)) / ng:.2e}; sharded vs general {np.linalg.norm(np.tril(H2 - Hg)) / ng:.2e}; "
              f"differing entries {int((np.tril(H2) != np.tril(Hf)).sum())}, first {np.argwhere(np.tril(H2) != np.tril(Hf))[:3].tolist()}")
        print(f"MISMATCH seed {s}: m={m} nvar={nvar} rank1={rank1} nlin={nlin} world={world} bs={bs} "
              f"rel err {np.linalg.norm(np.tril(H2 - Hf)) / max(np.linalg.norm(np.tril(Hf)), 1e-300):.2e}", flush=True)
    if (s - seed0) % 10 == 9: print(f"... {s - seed0 + 1} models, {bad} mismatches", flush=True)
print(f"done: {count} models, {bad} mismatches")
sys.exit(1 if bad else 0)
