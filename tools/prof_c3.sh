#!/bin/bash
# rocprofv3 kernel stats of the C3 solve; MODE = matvec_sparse option (1 dense, 0 auto)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c3
cat > /tmp/c3_one.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
dev = loraine_jl_amd.Device(0)
o = Optimizer(resident=True, device=dev); o.set_silent(True)
for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5).items(): o.set_attribute(k, v)
o.read_from_file(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "thetaG11.dat-s"))
o._copy_to()
dev.set_option("matvec_sparse", int(os.environ.get("MODE", "0")))
from loraine_jl_amd import solvers
solvers.solve(o.solver, o.halpha)
print("iterations", o.solver.iter, "cg", o.solver.cg_iter_tot, "objective", o.objective_value())
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -- python3 /tmp/c3_one.py > gpurun_out/prof_c3.log 2>&1
grep "^iterations" gpurun_out/prof_c3.log
f=$(find gpurun_out/prof_c3 -name "*kernel_stats.csv" | head -1)
head -${TOP:-25} $f | cut -c1-200
