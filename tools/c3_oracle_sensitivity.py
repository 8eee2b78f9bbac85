"""C3 (thetaG11, PCG + H_alpha): the CPU oracle against ITSELF from an initial X scaled by (1 + 1e-13).  Shows how far two
correct implementations of the kit=1 path can be expected to agree per iteration (truncated CG is not a contraction)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, json
from oracle import loraine_oracle as lo
path=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'thetaG11.dat-s')
opts=dict(kit=1, preconditioner=1, erank=1, eDIMACS=1e-5, verb=0, maxit=7)
res=[]
for eps in (0.0, 1e-13):
    s=lo.MySolver(lo.model_from_sdpa(path), opts)
    orig=lo.initial_point
    def ip(sol, eps=eps):
        orig(sol)
        if eps: sol.X[0] = sol.X[0]*(1.0+eps)
    lo.initial_point=ip
    lo.solve(s)
    lo.initial_point=orig
    res.append([(t["primal_obj"], t["dual_obj"], t["cg_pre"], t["cg_cor"]) for t in s.trace])
for k,(a,b) in enumerate(zip(*res)):
    print(f"it {k}: primal rel diff {abs(a[0]-b[0])/abs(a[0]):.2e} dual rel diff {abs(a[1]-b[1])/abs(a[1]):.2e} cg {a[2]},{a[3]} vs {b[2]},{b[3]}", flush=True)
