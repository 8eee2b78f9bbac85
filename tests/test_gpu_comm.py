"""The multi-GPU exchange inside the C library (csrc/comm.hip) with two ranks that share this box's one GPU: the
communicator's host transport over a gloo group carries exactly the calls RCCL carries on a multi-GPU node
(lrn_comm_init_host vs lrn_comm_init).  Whole interior-point solves: kit=0 general data (Schur column blocks +
all-gather), kit=0 dense data on the Cholesky path (column split of the matrix variable + all-reduce of the lower
triangle), rank-one data, kit=1 (partial operator + all-reduce of an nvar-vector inside lrn_pcg)."""
import json
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DENSE_MSZ = 256      # even and >= 256: the 16-byte kernels of the passes over dense constraint data run, rank by rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import loraine_jl_amd
    from loraine_jl_amd import resident
    from loraine_jl_amd.model import model_from_sdpa
    from loraine_jl_amd.sharding import DistributedHotPath
    from loraine_jl_amd.synthetic import LowRankProblem, synthetic_dense_solver
    dev = loraine_jl_amd.Device(0)
    out = {}
    cases = [("theta1", lambda: model_from_sdpa(os.path.join(GOLD, "theta1.dat-s")), dict(kit=0, eDIMACS=1e-6)),
             ("maxG11", lambda: model_from_sdpa(os.path.join(GOLD, "maxG11.dat-s"), datarank=-1), dict(kit=0, datarank=-1)),
             ("theta1_cg", lambda: model_from_sdpa(os.path.join(GOLD, "theta1.dat-s")),
              dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5)),
             ("lowrank_cg", lambda: LowRankProblem(60, 120, 4, seed=7).model(), dict(kit=1, preconditioner=2, erank=4, eDIMACS=1e-6))]
    # round 4: the CG operator through the assembled Schur matrix, every rank multiplying the column blocks it assembled
    # (matvec_h = 2 forces it at this size); the n^3 products of the resident path (Newton-Schulz, Lyapunov CG, step) as
    # column blocks + all-gather (shard_products_min lowered to this size)
    cases += [("lowrank_cg_h", lambda: LowRankProblem(60, 120, 4, seed=7).model(),
               dict(kit=1, preconditioner=2, erank=4, eDIMACS=1e-6), dict(matvec_h=2)),
              ("theta1_products", lambda: model_from_sdpa(os.path.join(GOLD, "theta1.dat-s")), dict(kit=0, eDIMACS=1e-6),
               dict(shard_products_min=16)),
              ("lowrank_products", lambda: LowRankProblem(60, 120, 4, seed=7).model(),
               dict(kit=1, preconditioner=2, erank=4, eDIMACS=1e-6), dict(shard_products_min=16, matvec_h=2))]
    for name, mk, opts, *lib in cases:
        model = mk()
        libopts = lib[0] if lib else {}
        for k, v in libopts.items():
            dev.set_option(k, v)
        dev.reset_timing()
        solver, ha = resident.load(model, dict(opts, verb=0), device=dev)
        hot = DistributedHotPath(solver, rank, world)
        solver.solve(ha)
        out[name] = dict(status=solver.status, iters=solver.iter, obj=-(float(model.b @ np.ravel(solver.y)) - model.b_const),
                         transport=getattr(hot, "transport", None), exchanges=dev.count("exchange"),
                         hop=dev.count("hop_matvec"), pgemm=dev.count("pgemm_sharded"))
        dev.comm_destroy()
        for k in libopts:
            dev.set_option(k, dict(matvec_h=0, shard_products_min=4096)[k])
    # dense data: the Cholesky path splits the columns of the matrix variable, the exchange is an all-reduce
    dev.set_option("schur_chol", 1)            # (auto takes the path from msz 256 on)
    solver, ha = synthetic_dense_solver(dev, DENSE_MSZ, 160, seed=11, options=dict(kit=0, verb=0))
    hot = DistributedHotPath(solver, rank, world)
    solver.solve(ha)
    out["dense"] = dict(status=solver.status, iters=solver.iter, obj=float(solver.primal_obj),
                        schur_chol=dev.count("schur_chol"), plan=dev.count("schur_plan_agreed"))
    dev.comm_destroy()
    # a rank that cannot allocate its exchange buffers must take its peers down with it, not leave them in the
    # collective.  Rank 1's next exchange fails its allocation (test hook "comm_fail_ensure"): BOTH ranks must come back
    # from lrn_schur_assemble with an error (rank 0 would sit in the all-gather for ever otherwise), and the next
    # assembly works again.
    model = model_from_sdpa(os.path.join(GOLD, "theta1.dat-s"))
    solver, ha = resident.load(model, dict(kit=0, verb=0), device=dev)
    hot = DistributedHotPath(solver, rank, world)
    dev.set_scaling(0, np.eye(int(model.msizes[0])))
    if rank == 1:
        dev.set_option("comm_fail_ensure", 1)
    try:
        dev.schur_assemble(0)
        out["inject"] = dict(raised=False, msg="")
    except loraine_jl_amd.LoraineHipError as e:
        out["inject"] = dict(raised=True, msg=str(e))
    try:
        dev.schur_assemble(0)
        out["inject"]["second_ok"] = dev.schur_factor() == 0
    except loraine_jl_amd.LoraineHipError as e:
        out["inject"]["second_ok"] = False
        out["inject"]["msg2"] = str(e)
    dev.comm_destroy()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_solve_through_the_library_communicator(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    for name, expect in (("theta1", 23.0), ("maxG11", 629.1648), ("theta1_cg", 23.0)):
        for k in range(2):
            assert r[k][name]["status"] == 1
            assert r[k][name]["obj"] == pytest.approx(expect, rel=3e-5)
            assert r[k][name]["transport"] == "host"
        assert r[0][name]["obj"] == r[1][name]["obj"]              # the replicated iteration stays in lock-step
        assert r[0][name]["iters"] == r[1][name]["iters"]
    assert r[0]["theta1"]["exchanges"] > 0                         # one exchange per assembly, inside lrn_schur_assemble
    assert r[0]["lowrank_cg"]["status"] == 1 and r[0]["lowrank_cg"]["obj"] == r[1]["lowrank_cg"]["obj"]
    # round 4 (see _worker): both ranks in lock-step, the sharded code ran, the optimum is that of the plain two-rank run
    for name, twin in (("lowrank_cg_h", "lowrank_cg"), ("theta1_products", "theta1"), ("lowrank_products", "lowrank_cg")):
        assert r[0][name]["status"] == 1 and r[0][name]["obj"] == r[1][name]["obj"] and r[0][name]["iters"] == r[1][name]["iters"]
        assert r[0][name]["obj"] == pytest.approx(r[0][twin]["obj"], rel=1e-7)
    assert r[0]["lowrank_cg_h"]["hop"] > 0 and r[1]["lowrank_cg_h"]["hop"] > 0
    assert r[0]["theta1_products"]["pgemm"] > 0 and r[0]["lowrank_products"]["pgemm"] > 0
    assert r[0]["theta1_products"]["iters"] == r[0]["theta1"]["iters"]
    assert r[0]["dense"]["status"] == 1 and r[0]["dense"]["obj"] == r[1]["dense"]["obj"]
    assert r[0]["dense"]["iters"] == r[1]["dense"]["iters"]
    assert r[0]["inject"]["raised"] and r[1]["inject"]["raised"]            # nobody was left in the collective
    assert "another rank" in r[0]["inject"]["msg"] and "injected" in r[1]["inject"]["msg"]
    assert r[0]["inject"]["second_ok"] and r[1]["inject"]["second_ok"]
    # the same dense problem on ONE rank (here, in the parent): the two-rank run -- column split of the assembly, the passes
    # over the constraint data split by constraints, both summed by all-reduces -- must land on the same iterates
    import loraine_jl_amd
    from loraine_jl_amd.synthetic import synthetic_dense_solver
    dev = loraine_jl_amd.Device(0)
    dev.set_option("schur_chol", 1)
    solver, ha = synthetic_dense_solver(dev, DENSE_MSZ, 160, seed=11, options=dict(kit=0, verb=0))
    solver.solve(ha)
    dev.close()
    assert solver.status == 1 and solver.iter == r[0]["dense"]["iters"]
    assert r[0]["dense"]["obj"] == pytest.approx(float(solver.primal_obj), rel=1e-9)
