"""First-contact GPU script: probes + timing of the dense assembly phases at a C4-like shape."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import loraine_jl_amd

msz = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nvar = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = loraine_jl_amd.Device(0)
out = {}
out["mfma_f64_peak_tflops"] = dev.mfma_f64_peak()
out["hbm_copy_gbps"] = dev.hbm_copy_peak(1 << 31)
print(out, flush=True)
t = time.time()
dev.synthetic_dense_model(msz, nvar, 20250614)
out["synth_s"] = time.time() - t
rng = np.random.default_rng(0)
G = rng.standard_normal((msz, msz)) / np.sqrt(msz) + np.eye(msz)
W = G @ G.T
dev.set_scaling(0, W, G)
dev.set_option("profile", 1)
for rep in range(3):
    dev.reset_timing()
    t = time.time()
    dev.schur_assemble(0)
    info = dev.schur_factor()
    x = dev.schur_solve(rng.standard_normal(nvar))
    wall = time.time() - t
    r = {k: dev.timing(k) for k in ("gemm1", "gemm2", "gemm3", "sparse", "assemble", "factor", "solve")}
    r["wall_s"] = wall; r["info"] = info
    fl1 = 2.0 * msz**3 * nvar
    r["gemm1_tflops"] = fl1 / (r["gemm1"] * 1e-3) / 1e12 if r["gemm1"] else 0
    tl = (msz + 127) // 128
    r["gemm2_tflops"] = (tl * (tl + 1) / 2) * 128 * 128 * 2.0 * msz * nvar / (r["gemm2"] * 1e-3) / 1e12 if r["gemm2"] else 0
    kp = sum(msz - (c // 128) * 128 for c in range(msz))
    tn = (nvar + 127) // 128
    r["gemm3_tflops"] = (tn * (tn + 1) / 2) * 128 * 128 * 2.0 * kp / (r["gemm3"] * 1e-3) / 1e12 if r["gemm3"] else 0
    print(json.dumps(r), flush=True)
    out[f"rep{rep}"] = r
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/gpu_first_{msz}_{nvar}.json", "w"), indent=1)
