import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
dev = loraine_jl_amd.Device(0)
o = Optimizer(resident=True, device=dev); o.set_silent(True)
for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5).items(): o.set_attribute(k, v)
o.read_from_file("/root/repo/tests/golden/thetaG11.dat-s")
o._copy_to()
s = o.solver
orig = s.myIPstep
def step(ha):
    orig(ha)
    print(s.iter, "plain", dev.count("lanczos_plain"), "steps", dev.count("prec_lanczos_steps"), "prec_ms %.2f pcg_ms %.2f cg %d" % (dev.timing("prec_setup"), dev.timing("pcg"), s.cg_iter_pre + s.cg_iter_cor), flush=True)
s.myIPstep = step
from loraine_jl_amd import solvers
solvers.solve(s, o.halpha)
