"""1-GPU check that a world=2 sharded assembly (two contexts on one device, exchange by
concatenation) gives the same H and the same solve as the unsharded path, dense MFMA path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import loraine_jl_amd

msz, nvar = 300, 700
W = None
def mk():
    d = loraine_jl_amd.Device(0)
    d.synthetic_dense_model(msz, nvar, 7)
    rng = np.random.default_rng(1)
    G = rng.standard_normal((msz, msz)) / np.sqrt(msz) + np.eye(msz)
    d.set_scaling(0, G @ G.T, G)
    return d
full = mk()
Hf = full.schur_assemble(0, want_H=True)
parts, devs = [], []
for r in range(2):
    d = mk(); d.set_shard(r, 2); d.schur_assemble(0)
    if d.schur_is_partial_sum():          # dense data: column split of the matrix variable, partial sums
        buf = torch.zeros(nvar * nvar, dtype=torch.float64, device="cuda")
        d.schur_export_full(buf)
    else:
        buf = torch.zeros(d.shard_doubles(), dtype=torch.float64, device="cuda")
        d.schur_export_shard(buf)
    parts.append(buf); devs.append(d)
torch.cuda.synchronize()
partial = devs[0].schur_is_partial_sum()
allb = (parts[0] + parts[1]) if partial else torch.cat(parts)      # what the all-reduce / all-gather delivers
torch.cuda.synchronize()
for d in devs:
    if partial:
        d.schur_import_full(allb)
    else:
        d.schur_import_all(allb)
    H2 = d.schur_get()
    err = np.linalg.norm(H2 - Hf) / np.linalg.norm(Hf)
    assert err < 1e-13, err
    assert d.schur_factor() == 0
h = np.random.default_rng(2).standard_normal(nvar)
x0 = devs[0].schur_solve(h); x1 = devs[1].schur_solve(h)
assert np.array_equal(x0, x1)
assert np.linalg.norm(Hf @ x0 - h) / np.linalg.norm(h) < 1e-9
print("sharded dense assembly == unsharded: OK, rel err", err)
