"""Interior-point solves with the hot path sharded over the ranks of a torch.distributed job (one
process per GPU; RCCL).  Rehearsal on a one-GPU box:
  LRN_BENCH_BACKEND=gloo LRN_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \\
      --master-addr 127.0.0.1 --master-port 29541 tools/solve_multirank.py
Every rank runs the same iteration; only Schur columns (kit=0) / the mat-vec (kit=1) are split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
backend = os.environ.get("LRN_BENCH_BACKEND", "nccl")
dev_index = 0 if os.environ.get("LRN_BENCH_ONE_GPU") else local
torch.cuda.set_device(dev_index)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
else:
    dist.init_process_group(backend)
import loraine_jl_amd
from loraine_jl_amd import resident
from loraine_jl_amd.model import model_from_sdpa
from loraine_jl_amd.sharding import DistributedHotPath
from loraine_jl_amd.synthetic import LowRankProblem

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = loraine_jl_amd.Device(dev_index)
cases = [("theta1 kit=0", lambda: model_from_sdpa(os.path.join(G, "theta1.dat-s")), dict(kit=0, eDIMACS=1e-6, initpoint=1), 23.0, -1),
         ("control1 kit=0", lambda: model_from_sdpa(os.path.join(G, "control1.dat-s")), dict(kit=0), 17.78463, -1),
         ("maxG11 rank-one", lambda: model_from_sdpa(os.path.join(G, "maxG11.dat-s"), datarank=-1), dict(kit=0, datarank=-1), 629.1648, -1),
         ("thetaG11 kit=1 H_alpha", lambda: model_from_sdpa(os.path.join(G, "thetaG11.dat-s")), dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5), 400.0, -1)]
P = LowRankProblem(int(os.environ.get("C5_MSZ", "1600")), int(os.environ.get("C5_NVAR", "3200")), 4)
cases.append(("lowrank kit=1 H_beta", P.model, dict(kit=1, preconditioner=2, erank=4, eDIMACS=1e-5), -P.optimum, +1))
ok = True
for name, mk, opts, expect, sgn in cases:
    model = mk()
    solver, ha = resident.load(model, dict(opts, verb=0), device=dev)
    DistributedHotPath(solver, rank, world)
    t = time.perf_counter(); solver.solve(ha); wall = time.perf_counter() - t
    obj = -(float(model.b @ np.ravel(solver.y)) - model.b_const)
    objs = [None] * world
    dist.all_gather_object(objs, obj)
    same = all(o == objs[0] for o in objs)          # replicated iteration stays in lockstep
    good = solver.status == 1 and abs(obj - expect) <= 2e-5 * (1 + abs(expect)) and same
    ok &= good
    if rank == 0:
        print(json.dumps(dict(case=name, world=world, status=solver.status, iters=solver.iter, obj=obj, expect=expect,
                              identical_on_all_ranks=same, wall_s=round(wall, 3), cg=solver.cg_iter_tot, ok=good)), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
