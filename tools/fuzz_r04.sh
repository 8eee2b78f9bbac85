#!/bin/bash
# the round-3 fuzz record (profiles/r04_fuzz_parity.txt): default options and strict mode
cd ${GRAFT_REPO_ROOT:-/root/repo}/tools
run() { echo "# $*"; env "$@" 2>&1 | grep -E "MISMATCH|done"; }
echo "# tools/fuzz_parity.py, round 4 (final tree), default options (pivot boosting on; resident driver = eigen-free NT scaling, host driver = SVD route)"
run timeout -k 10 300 python fuzz_parity.py 0 150
run FUZZ_HUGE=1 timeout -k 10 400 python fuzz_parity.py 600 16
run FUZZ_RANK1=1 timeout -k 10 300 python fuzz_parity.py 3000 30
run FUZZ_BIG=1 FUZZ_KIT1=1 timeout -k 10 400 python fuzz_parity.py 5000 10
echo "# FUZZ_STRICT=1 (pivot_boost = 0, the literal reference behaviour)"
run FUZZ_STRICT=1 timeout -k 10 300 python fuzz_parity.py 0 150
run FUZZ_STRICT=1 FUZZ_HUGE=1 timeout -k 10 400 python fuzz_parity.py 600 16
echo "# round 4 paths forced: CG operator through the assembled Schur matrix, H_alpha as a dense matrix (needs nvar >= 256: off at these sizes), Lanczos-scaled Newton-Schulz, right-hand sides through the constraint pattern"
run FUZZ_LRN_OPTS=matvec_h=2,prec_dense=2,ns_lanczos_min=8,wmw_pattern_min=2 FUZZ_BIG=1 FUZZ_KIT1=1 timeout -k 10 500 python fuzz_parity.py 5000 16
run FUZZ_LRN_OPTS=ns_lanczos_min=8,wmw_pattern_min=2 timeout -k 10 300 python fuzz_parity.py 0 100
