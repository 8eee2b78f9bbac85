import os, sys
sys.path.insert(0, "/root/repo")
import loraine_jl_amd
from loraine_jl_amd.optimizer import Optimizer
for resident in (False, True):
    o = Optimizer(resident=resident)
    for k, v in dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5, verb=1).items(): o.set_attribute(k, v)
    o.read_from_file("/root/repo/tests/golden/thetaG11.dat-s")
    o.optimize()
    print("resident", resident, "iterations", o.solver.iter, "objective", o.objective_value(), flush=True)
