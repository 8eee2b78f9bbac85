#!/usr/bin/env python3
"""bench.py -- ms per IP iteration (Schur assembly + solve) on the BASELINE metric config:
synthetic dense SDP, matrix side 2000, 4000 constraints (SURVEY.md section 8d "C4"),
kit=0, FP64, data generated on the device (128 GB of constraint matrices).

One "step" = one pass of the hot path over one iterate: assemble H (GEMM1/2/3 on the FP64
MFMA; on one or two ranks through the Cholesky factor of W), factor it
(blocked Cholesky) and run the predictor and corrector solves.  Inputs
(constraint data, NT scaling W, right-hand sides) are resident in HBM when the timed region
starts.  N > 1: one process per GPU; the ranks split the columns of the matrix variable
(every GEMM of the assembly shards) and one RCCL all-reduce adds their partial Schur matrices before
the replicated factorisation; total work is fixed (`"scaling": "strong"`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--msz 2000] [--nvar 4000]

`python bench.py --gpus N` with N > 1 and no torchrun environment starts its own N ranks
(`python -m torch.distributed.run --nproc-per-node N`) BEFORE anything touches a GPU and forwards their
output and exit code; launched under torchrun it is one rank.  WORLD_SIZE != --gpus is an error.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # AMD datasheet, MI355X FP64 matrix; the
#                                  local microarch guide has no FP64 MFMA row -- the measured
#                                  issue-rate probe is reported next to it as `peak_probe`.


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--msz", type=int, default=2000)
    ap.add_argument("--nvar", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=20250614)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-msz", type=int, default=300)
    ap.add_argument("--cpu-nvar", type=int, default=400)
    ap.add_argument("--cpu-msz2", type=int, default=500)        # SURVEY.md 8d's scaled instance (bounded sample)
    ap.add_argument("--cpu-nvar2", type=int, default=1000)
    ap.add_argument("--cpu-sample2", type=int, default=96)      # constraints of it timed on the CPU
    return ap.parse_args()


def flops_model(msz, nvar):
    """algorithmic work of one kit=0 IP iteration on dense data."""
    return 4.0 * nvar * msz ** 3 + float(nvar) ** 2 * msz ** 2 + nvar ** 3 / 3.0 + 8.0 * nvar ** 2


def pmc_traffic_bytes(kernel_pat):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/,
    same command, same launch shape): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports
    half the bytes of a 16 B/lane stream (MI355X_MICROARCH.md section HBM), which is what the direct-to-LDS
    staging of these kernels issues; WRITE_SIZE is exact."""
    import csv
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.csv"))):
        vals = {}
        try:
            for r in csv.DictReader(open(f)):
                if kernel_pat in r.get("kernel", "") and r.get("counter") in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[r["counter"]] = float(r["mean"])
        except (OSError, ValueError, KeyError):      # a malformed summary must never take the bench line down
            continue
        if len(vals) == 2:
            best = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return best


def make_scaling(msz, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((msz, msz)) / np.sqrt(msz) + np.eye(msz)
    return G @ G.T, G


class _DenseRows:
    """A[ilmi]-like view of dense constraint rows (index k >= 1 -> CSC of A_k), so that the oracle's makeBBBBsi can
    be fed without the oracle's general (slow, per-entry) model builder."""

    def __init__(self, rows, msz):
        self.rows, self.msz = rows, msz

    def __getitem__(self, k):
        import scipy.sparse as sp
        return sp.csc_matrix(self.rows[k - 1].reshape(self.msz, self.msz))


def _dense_instance(msz, nvar, seed):
    """Same generator as the metric config (A_k = (R + R')/2, R iid N(0,1)) in the oracle's model form: AA = -vec(A_k)
    rows as CSR (src/model.jl:219-226), sigmaA = identity (all nnz equal: the stable nnz sort of src/model.jl:159
    keeps the order), qA = nvar (every constraint takes branch 1, src/makeBBBB.jl:81)."""
    import numpy as np
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    mm = msz * msz
    rows = np.empty((nvar, mm))
    for k in range(nvar):
        R = rng.standard_normal((msz, msz))
        rows[k] = ((R + R.T) / 2).reshape(-1)
    AA = sp.csr_matrix((-rows.reshape(-1), np.tile(np.arange(mm, dtype=np.int32), nvar),
                        np.arange(nvar + 1, dtype=np.int64) * mm), shape=(nvar, mm))
    sigmaA = np.arange(nvar, dtype=np.int64).reshape(nvar, 1)
    qA = np.full((2, 1), nvar, dtype=np.int64)
    return rows, AA, sigmaA, qA, rng


def _gpu_same_sample(rows, AA, sigmaA, qA, msz, W, G, h1, h2):
    import numpy as np
    import torch
    import loraine_jl_amd
    dev = loraine_jl_amd.Device(0)
    dev.set_option("dense_threshold", 1)
    dev.upload_model([AA], sigmaA, qA, [msz])
    dev.set_scaling(0, W, G)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.schur_assemble(0)
        assert dev.schur_factor() == 0
        xg = dev.schur_solve(h1)
        xg = dev.schur_solve(h2)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) * 1e3
    dev.close()
    return gpu_ms, np.asarray(xg)


def cpu_baseline(msz, nvar, seed, msz2=500, nvar2=1000, sample2=96):
    """The CPU restatement of the reference path (oracle, kind 'port') timed on the host cores on bounded samples of
    the same workload, with the GPU path on the very same inputs:
      * `value`: the whole hot path (assembly, Cholesky, two solve pairs) of a small instance of the same generator;
      * `scaled_instance`: SURVEY.md section 8d's scaled instance (matrix side 500, 1000 constraints): the
        constraint loop of makeBBBBsi (src/makeBBBB.jl:77) is timed for its first `sample2` constraints -- every
        pass does the same work: two msz^3 products and one AA * vec(tmp) over all nvar rows -- and scaled by
        nvar / sample2; factor + solves are timed on the GPU-assembled matrix of the full instance."""
    import numpy as np
    import scipy.linalg as sla
    from oracle import loraine_oracle as lo
    rows, AA, sigmaA, qA, rng = _dense_instance(msz, nvar, seed)
    W, G = make_scaling(msz, seed + 1)
    h1, h2 = rng.standard_normal(nvar), rng.standard_normal(nvar)
    t0 = time.perf_counter()
    H = lo.makeBBBBsi(0, _DenseRows(rows, msz), AA, W, nvar, qA, sigmaA)
    Hl = np.tril(H)
    L = np.linalg.cholesky(Hl + np.tril(Hl, -1).T)
    for h in (h1, h2):
        x = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    gpu_ms, xg = _gpu_same_sample(rows, AA, sigmaA, qA, msz, W, G, h1, h2)
    err = float(np.linalg.norm(xg - x) / np.linalg.norm(x))
    fl = flops_model(msz, nvar)
    out = {
        "value": cpu_ms, "unit": "ms/IP-iteration (Schur assembly + solve) on the sample instance",
        "cores": os.cpu_count(), "kind": "port",
        "sample": f"same generator, dense SDP matrix side {msz}, {nvar} constraints "
                  f"({fl / 1e9:.1f} GFLOP algorithmic); CPU = NumPy/SciPy restatement of the reference path "
                  f"(OpenBLAS GEMM threads = all host cores, SciPy SpMV single-threaded like SparseArrays')",
        "cpu_gflops": fl / cpu_ms / 1e6,
        "gpu_ms_same_sample": gpu_ms,
        "gpu_vs_cpu_rel_err_dely": err,
    }
    if msz2 > 0 and nvar2 > 0 and sample2 > 0:
        del rows, AA, H, Hl, L
        rows, AA, sigmaA, qA, rng = _dense_instance(msz2, nvar2, seed + 11)
        W, G = make_scaling(msz2, seed + 12)
        h1, h2 = rng.standard_normal(nvar2), rng.standard_normal(nvar2)
        q = min(sample2, nvar2)
        t0 = time.perf_counter()
        Hs = lo.makeBBBBsi(0, _DenseRows(rows, msz2), AA, W, nvar2, qA, sigmaA, ii_stop=q)
        loop_ms = (time.perf_counter() - t0) * 1e3
        gpu2_ms, xg2 = _gpu_same_sample(rows, AA, sigmaA, qA, msz2, W, G, h1, h2)
        # factor + solves on the full Schur matrix: rebuild it from the solution-independent identity H = AA T' ...
        # cheaper: time them on the GPU-assembled H of the same instance
        import loraine_jl_amd
        dev = loraine_jl_amd.Device(0)
        dev.set_option("dense_threshold", 1)
        dev.upload_model([AA], sigmaA, qA, [msz2])
        dev.set_scaling(0, W, G)
        Hfull = dev.schur_assemble(0, want_H=True)
        dev.close()
        # the sampled columns against the full GPU matrix: the CPU sample computed columns sigma[0:q] completely
        cols = sigmaA[:q, 0]
        col_err = float(np.linalg.norm(Hs[:, cols] - Hfull[:, cols]) / np.linalg.norm(Hfull[:, cols]))
        t0 = time.perf_counter()
        L = np.linalg.cholesky(Hfull)
        for h in (h1, h2):
            x = sla.solve_triangular(L.T, sla.solve_triangular(L, h, lower=True), lower=False)
        fs_ms = (time.perf_counter() - t0) * 1e3
        fl2 = flops_model(msz2, nvar2)
        scaled = loop_ms * nvar2 / q + fs_ms
        out["scaled_instance"] = {
            "msz": msz2, "nvar": nvar2, "sampled_constraints": q, "constraint_loop_ms_sampled": loop_ms,
            "factor_and_solves_ms": fs_ms, "value_scaled_ms": scaled,
            "scaling": f"constraint-loop time of the first {q} of {nvar2} constraints x {nvar2}/{q} + factor and "
                       f"two solve pairs of the full {nvar2} x {nvar2} Schur matrix (timed in full)",
            "cpu_gflops": fl2 / scaled / 1e6, "gpu_ms_same_instance": gpu2_ms,
            "sampled_columns_rel_err_vs_gpu": col_err,
            "gpu_vs_cpu_rel_err_dely": float(np.linalg.norm(xg2 - x) / np.linalg.norm(x)),
        }
    return out


def self_launch(args):
    """--gpus N > 1 without a torchrun environment: this process becomes the launcher.  It starts the N ranks as
    children, never imports torch or touches a GPU itself, forwards the ranks' output (rank 0 prints the JSON line)
    and exits with their exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    import numpy as np
    import torch
    import torch.distributed as dist

    # rehearsal knobs (1-GPU box): LRN_BENCH_BACKEND=gloo LRN_BENCH_ONE_GPU=1 run all ranks on
    # device 0 and stage the exchange through host memory; the driver's runs use RCCL.
    backend = os.environ.get("LRN_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("LRN_BENCH_ONE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    # LRN_BENCH_DIST1=1: take the sharded path (process group, export, all-gather, import) with
    # world_size 1 -- exercises RCCL and the device-pointer exchange on a one-GPU box
    sharded = world > 1 or bool(os.environ.get("LRN_BENCH_DIST1"))
    if sharded and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import loraine_jl_amd
    dev = loraine_jl_amd.Device(dev_index)
    msz, nvar = args.msz, args.nvar
    t0 = time.perf_counter()
    dev.synthetic_dense_model(msz, nvar, args.seed)
    t_gen = time.perf_counter() - t0
    W, G = make_scaling(msz, args.seed + 1)
    dev.set_scaling(0, W, G)
    rng = np.random.default_rng(args.seed + 2)
    h_pred = torch.from_numpy(rng.standard_normal(nvar)).cuda()
    h_corr = torch.from_numpy(rng.standard_normal(nvar)).cuda()
    dely = torch.zeros(nvar, dtype=torch.float64, device="cuda")
    from loraine_jl_amd import sharding
    if sharded:
        # the exchange lives in the library (csrc/comm.hip): RCCL communicator created from a unique id that travels
        # through the launcher's process group; every lrn_schur_assemble then agrees on the path (first call), reduces a
        # status word every rank enters, and all-reduces / all-gathers on the library's stream.  With the rehearsal
        # backend (gloo, ranks sharing one GPU) the same entry points run over host callbacks.
        transport = dev.comm_init_torch(rank, world)
    from loraine_jl_amd._capi import ptr
    lib = dev.lib

    def step():
        dev.schur_assemble(0)                                   # makeBBBBs on the owned share + the exchange
        info = dev.schur_factor()                               # cholesky(BBBB)
        assert info == 0, f"Schur matrix not PD (info={info})"
        dev._chk(lib.lrn_schur_solve(dev.h, ptr(h_pred), ptr(dely)), "solve")   # predictor
        dev._chk(lib.lrn_schur_solve(dev.h, ptr(h_corr), ptr(dely)), "solve")   # corrector

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("LRN_GEMM3_KSPLIT"):                     # experiment knob (split-K factor of GEMM3')
        dev.set_option("gemm3_ksplit", int(os.environ["LRN_GEMM3_KSPLIT"]))
    if os.environ.get("LRN_P_BATCH"):                          # experiment knob: constraint matrices per GEMM1'/2' launch
        dev.set_option("p_batch", int(os.environ["LRN_P_BATCH"]))
    if os.environ.get("LRN_GEMM3_STAGGER"):
        dev.set_option("gemm3_stagger", int(os.environ["LRN_GEMM3_STAGGER"]))
    dev.set_option("profile", 0)
    for _ in range(args.warmup):
        step()
    dev.set_option("profile", 1)          # per-kernel HIP-event timing on the library's stream
    dev.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if sharded:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3

    # the sharded path must leave the same solution on every rank (replicated factor + solves on the
    # all-gathered matrix): compare a checksum of the corrector solve across ranks
    chk = float(dely.double().sum().item())
    if sharded:
        chks = [None] * world
        dist.all_gather_object(chks, chk)
        assert all(c_ == chks[0] for c_ in chks), f"ranks disagree on dely: {chks}"
    # ---- every rank prices ITS dominant kernel; the line reports the slowest rank (the one the step waits for)
    def rank_report():
        own = sharding.owned_columns(nvar, rank, world, bs=dev.shard_bs())
        nown = len(own)
        chol_path = dev.count("schur_chol") > 0
        via_l = dev.count("schur_via_l") > 0                        # W path with T_k = L (L'A_kL) L'  (mixed data / forced)
        per_step = {k: dev.timing(k) / args.steps for k in ("gemm1", "gemm2", "gemm3")}
        dom = max(per_step, key=per_step.get)                       # dominant kernel of this rank
        nl = max(1, dev.count(dom))
        t1 = dev.timing(dom) / nl                                   # ms per launch (HIP events on the library's stream)
        launches_per_step = nl / args.steps
        # algorithmic flop of one launch = flop the formulation needs (no tile padding, symmetry and
        # triangularity used) for the units the launch processes
        if dom == "gemm3":
            # packed-symmetric inner products <.,.> of msz x msz symmetric matrices: msz (msz + 1) flop per
            # Schur entry; units = lower-triangle entries of the owned columns (position space)
            entries = float(sum(nvar - int(j) for j in own))
            alg_flops_launch = entries * msz * (msz + 1.0) / launches_per_step
            strip = chol_path and dev.count("gemm3s") > 0
            if chol_path:
                # every rank forms ALL entries over ITS columns of the matrix variable: its share of the packed
                # length (1 on one GPU).  With nvar % 128 in (0, 32] the last 128 + nvar % 128 rows of H are a launch
                # of their own ("gemm3s", 128 x 160 tiles): the launch priced here holds the leading rows
                lead = nvar - 128 - nvar % 128 if strip else nvar
                alg_flops_launch = (lead * (lead + 1.0) / 2.0) * msz * (msz + 1.0) * dev.timing("gemm3_share") / launches_per_step
            kname = (("gemm_f64_kseg_lds_kernel<true, 4, 4> GEMM3' H[j,i] = <L'A_jL, L'A_iL>, rows j < %d (packed lower tiles, "
                      "split-K; the last %d rows: a second launch of 128 x 160 tiles, phase gemm3s)" % (lead, nvar - lead))
                     if strip else
                     "gemm_f64_kseg_lds_kernel<true> GEMM3' H[j,i] = <L'A_jL, L'A_iL> (packed lower tiles, split-K)"
                     if chol_path else "gemm_f64_kseg_lds_kernel<false> GEMM3 H[j,i] = <A_j, W A_i W> (lower tiles, split-K)")
            kpat = (("gemm_f64_kseg_lds_kernel<true, 4, 4>" if strip else "gemm_f64_kseg_lds_kernel<true") if chol_path
                    else "gemm_f64_kseg_lds_kernel<false")     # (<FLAT, tile blocks>)
        elif dom == "gemm1":
            units_per_launch = (nvar if chol_path else nown) / launches_per_step      # constraint matrices per launch
            alg_flops_launch = (2.0 / 3.0 if chol_path else (1.0 if via_l else 2.0)) * msz ** 3 * units_per_launch
            if chol_path:
                alg_flops_launch *= dev.timing("gemm1_share")       # this rank's columns of the matrix variable
            kname = ("gemm_f64_lds_kernel<false> GEMM1' P_k = A_k L (lower tiles, triangular K)" if chol_path
                     else ("gemm_f64_lds_kernel P_k = A_k L and At_k = L'P_k (timed together; lower tiles, triangular K)" if via_l
                           else "gemm_f64_lds_kernel<false> GEMM1 P_k = A_k W (batched, direct-to-LDS staging)"))
            kpat = "gemm_f64_lds_kernel<false>"
        else:
            units_per_launch = (nvar if chol_path else nown) / launches_per_step
            alg_flops_launch = (1.0 / 3.0 if chol_path else 1.0) * msz ** 3 * units_per_launch
            if chol_path:
                alg_flops_launch *= dev.timing("gemm2_share")
            kname = ("gemm_f64_lds_kernel Q_k = L At_k and T_k = Q_k L' (timed together; lower tiles, triangular K)" if via_l
                     else "gemm_f64_lds_kernel<true> GEMM2 (lower tiles)")
            kpat = "gemm_f64_lds_kernel<true>"
        achieved = alg_flops_launch / (t1 * 1e-3) / 1e12
        phases = {k: dev.timing(k) / args.steps
                  for k in ("wchol", "gemm1", "gemm2", "gemm3", "gemm3s", "reduce3", "assemble", "exchange", "factor", "solve")}
        cols = sharding.column_range(msz, nvar, rank, world) if chol_path else None
        return {"rank": rank, "chol_path": chol_path, "via_l": via_l, "kpat": kpat, "phases": phases,
                "columns": list(cols) if cols else None,
                "shares": {k: dev.timing(k + "_share") for k in ("gemm1", "gemm2", "gemm3")} if chol_path else None,
                "roofline": {"bound": "mfma", "kernel": kname, "rank": rank,
                             "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                             "avg_launch_ms": t1, "launches_per_step": launches_per_step,
                             "alg_flops_per_launch": alg_flops_launch}}

    mine = rank_report()
    reports = [mine]
    if sharded:
        reports = [None] * world
        dist.all_gather_object(reports, mine)
    if rank == 0:
        slow = max(reports, key=lambda r_: r_["phases"]["assemble"])          # the rank the step waits for
        chol_path, via_l = slow["chol_path"], slow["via_l"]
        roof = slow["roofline"]
        if (msz, nvar, world) == (2000, 4000, 1):
            roof["traffic"] = pmc_traffic_bytes(slow["kpat"])
        roof["peak_probe"] = dev.mfma_f64_peak()
        out = {
            "metric": "ms/IP-iteration (Schur assembly + solve), dense SDP n=2000 m=4000",
            "value": ms_per_step, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ranks_seen": dist.get_world_size() if sharded else 1,
            "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C4 synthetic dense SDP, matrix side {msz}, {nvar} constraints, kit=0 "
                                   f"(assembly + Cholesky + predictor/corrector solves)",
                       "msz": msz, "nvar": nvar, "seed": args.seed,
                       "parallelism": "1 GPU" if world == 1 else (
                           f"columns of the matrix variable over {world} GPUs + RCCL all-reduce of the partial Schur matrices"
                           if chol_path else f"Schur column blocks over {world} GPUs + RCCL all-gather")},
            "algorithmic_tflops": flops_model(msz, nvar) / (ms_per_step * 1e-3) / 1e12,
            "assembly_path": ("cholesky (H_ij = <L'A_iL, L'A_jL>, W = LL')" if chol_path
                              else ("T_k = L (L'A_kL) L', W = LL'" if via_l else "T_k = W A_k W")),
            "roofline": roof,
            "phase_ms_per_step": slow["phases"],
            "dely_checksum": chk,
            "data_gen_s": t_gen,
        }
        if world > 1:
            out["per_rank"] = [{"rank": r_["rank"], "columns": r_["columns"], "assemble_ms": r_["phases"]["assemble"],
                                "gemm1_ms": r_["phases"]["gemm1"], "gemm2_ms": r_["phases"]["gemm2"],
                                "gemm3_ms": r_["phases"]["gemm3"], "roofline_frac": r_["roofline"]["frac"]}
                               for r_ in reports]
            out["exchange_ms_per_step"] = dev.timing("exchange") / args.steps     # status reduction excluded: HIP events around the collective
            out["exchange_transport"] = transport
        if world == 1 and not os.environ.get("LRN_BENCH_NO_ALONGSIDE"):
            # prepare_W, the right-hand sides and find_step "reported alongside" -- from a short
            # device-resident solve of the same instance (first iterations of the real predictor-corrector loop, outside
            # the timed region): mean over iterations 2.. of the phase times (HIP events on the library's stream)
            from loraine_jl_amd.synthetic import synthetic_dense_solver
            solver, ha = synthetic_dense_solver(dev, msz, nvar, seed=args.seed, options=dict(kit=0, verb=0, maxit=4))
            solver.solve(ha)
            tr = solver.trace[1:] if len(solver.trace) > 1 else solver.trace
            mean = lambda f: float(np.mean([f(x) for x in tr])) if tr else None
            out["alongside"] = {
                "source": "resident IP solve of the same instance, iterations 2..%d (outside the timed region)" % len(solver.trace),
                "prepare_w_ms": mean(lambda x: x["gpu_ms"]["prepare_w"]),
                "rhs_ms": mean(lambda x: x["rhs_ms"]), "residual_d_ms": mean(lambda x: x["residual_d_ms"]),
                "find_step_ms": mean(lambda x: x["find_step_ms"]), "lyapunov_ms": mean(lambda x: x["lyap_ms"]),
                "assemble_ms": mean(lambda x: x["gpu_ms"]["assemble"]), "factor_ms": mean(lambda x: x["gpu_ms"]["factor"]),
                "solve_ms": mean(lambda x: x["gpu_ms"]["solve"]),
                "full_iteration_ms": mean(lambda x: x["itertime"] * 1e3)}
        if world == 1 and not args.no_cpu_baseline:
            dev.close()
            out["cpu_baseline"] = cpu_baseline(args.cpu_msz, args.cpu_nvar, args.seed + 7, args.cpu_msz2,
                                               args.cpu_nvar2, args.cpu_sample2)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
