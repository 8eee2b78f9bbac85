#!/bin/bash
# 2-rank rehearsal of bench.py's sharded path on a 1-GPU box (gloo exchange through host memory).
# Compares the sharded solve against the 1-rank solve on the same (small) synthetic problem.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LRN_BENCH_BACKEND=gloo LRN_BENCH_ONE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --steps 2 --warmup 1 --msz 512 --nvar 700 --no-cpu-baseline
# 3 ranks: the W path through the Cholesky factor (T_k = L (L'A_kL) L') on the owned columns
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29537 \
   bench.py --gpus 3 --steps 2 --warmup 1 --msz 512 --nvar 900 --no-cpu-baseline
python tools/check_sharded_solve.py
# full interior-point solves with the hot path sharded over the 2 ranks (kit=0 all-gather, kit=1 all-reduce)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
   tools/solve_multirank.py
