"""`Optimizer`: the user-facing surface of the reference (`Loraine.Optimizer`,
src/MOI_wrapper.jl:42-354) with the same 15 raw attribute names, status mapping and getters,
driving the MI355X hot path.  MathOptInterface itself is a Julia package; this mirrors the
calls a JuMP user makes:

    set_optimizer(model, Loraine.Optimizer)      ->  opt = Optimizer()
    set_attribute(model, "kit", 0)               ->  opt.set_attribute("kit", 0)
    read_from_file(".../theta1.dat-s")           ->  opt.read_from_file(path)
    optimize!(model)                             ->  opt.optimize()
    objective_value(model)                       ->  opt.objective_value()
"""
import numpy as np

from . import solvers
from .model import MyModel, build_model, model_from_sdpa

# MOI.TerminationStatus values used by the reference (MOI_wrapper.jl:252-265)
OPTIMIZE_NOT_CALLED = "OPTIMIZE_NOT_CALLED"
OPTIMAL = "OPTIMAL"
INFEASIBLE = "INFEASIBLE"
INFEASIBLE_OR_UNBOUNDED = "INFEASIBLE_OR_UNBOUNDED"
ITERATION_LIMIT = "ITERATION_LIMIT"
FEASIBLE_POINT, INFEASIBLE_POINT, UNKNOWN_RESULT_STATUS, NO_SOLUTION = (
    "FEASIBLE_POINT", "INFEASIBLE_POINT", "UNKNOWN_RESULT_STATUS", "NO_SOLUTION")


class UnsupportedAttribute(KeyError):
    """MOI.UnsupportedAttribute (MOI_wrapper.jl:90-94)."""


class Optimizer:
    def __init__(self, device=None, device_index=0, resident=True, exact_regularised_solve=False):
        # resident=True: X, S and all msz x msz work stay on the device (resident.ResidentSolver);
        # resident=False: step-length search / convergence test in host NumPy (solvers.MySolver)
        self.resident = resident
        # False (default): regularised iterations solve twice, like the reference (src/predictor_corrector.jl:85,90);
        # True: (H + d I)^-1 h.  Not one of the reference's 15 options -- a constructor argument.
        self.exact_regularised_solve = bool(exact_regularised_solve)
        self.solver = None
        self.halpha = None
        self.max_sense = False
        self.silent = False
        self.options = dict(solvers.DEFAULT_OPTIONS)
        self._device = device
        self._device_index = device_index
        self._pending = None

    # ---- MOI.RawOptimizerAttribute (MOI_wrapper.jl:86-103)
    def supports(self, name):
        return name in solvers.DEFAULT_OPTIONS

    def set_attribute(self, name, value):
        if not self.supports(name):
            raise UnsupportedAttribute(name)
        self.options[name] = value

    def get_attribute(self, name):
        if not self.supports(name):
            raise UnsupportedAttribute(name)
        return self.options[name]

    def set_silent(self, flag=True):          # MOI.Silent (:107-114)
        self.silent = bool(flag)

    def solver_name(self):
        return "Loraine"

    # ---- model input (copy_to, :142-232): options are frozen here, like in the reference (:224)
    def read_from_file(self, path):
        self._pending = ("sdpa", path)
        return self

    def load_model(self, A, b, b_const=0.0, d_lin=None, C_lin=None, max_sense=False):
        """Problem in the reference's own form: max/min over y with LMIs sum_j y_j A_ij - A_i0 >= 0
        given as A[i] = [F_0, F_1, ..., F_n] and rows  C_lin' y <= d_lin."""
        self._pending = ("arrays", (A, np.asarray(b, float), float(b_const), d_lin, C_lin, bool(max_sense)))
        return self

    def _copy_to(self):
        kind, payload = self._pending
        drank = int(self.options.get("datarank", 0))
        kappa = int(self.options.get("datasparsity", 8))
        if kind == "sdpa":
            model = model_from_sdpa(payload, datarank=drank, kappa=kappa)
            self.max_sense = False
        else:
            A, b, b_const, d_lin, C_lin, max_sense = payload
            self.max_sense = max_sense
            model = build_model(A, b, b_const, d_lin, C_lin, datarank=drank, kappa=kappa)
        opts = dict(self.options)
        if self.silent:
            opts["verb"] = 0
        if self._device is None:
            from .device import Device
            self._device = Device(self._device_index)
        if self.resident:
            from . import resident
            self.solver, self.halpha = resident.load(model, opts, device=self._device)
        else:
            self.solver, self.halpha = solvers.load(model, opts, device=self._device)
        self.solver.exact_regularised_solve = self.exact_regularised_solve

    def optimize(self):                        # MOI.optimize! (:136-140)
        if self._pending is None:
            raise RuntimeError("no model loaded")
        self._copy_to()
        solvers.solve(self.solver, self.halpha)
        return self

    # ---- results (MOI_wrapper.jl:241-354)
    def termination_status(self):
        if self.solver is None or self.solver.status == 0:
            return OPTIMIZE_NOT_CALLED
        return {1: OPTIMAL, 2: INFEASIBLE, 3: INFEASIBLE_OR_UNBOUNDED, 4: ITERATION_LIMIT}[self.solver.status]

    def raw_status(self):
        return f"Terminated with status {self.solver.status}"

    def primal_status(self):
        t = self.termination_status()
        return {OPTIMIZE_NOT_CALLED: NO_SOLUTION, OPTIMAL: FEASIBLE_POINT, INFEASIBLE: INFEASIBLE_POINT}.get(
            t, UNKNOWN_RESULT_STATUS)

    def dual_status(self):
        t = self.termination_status()
        return {OPTIMIZE_NOT_CALLED: NO_SOLUTION, OPTIMAL: FEASIBLE_POINT}.get(t, UNKNOWN_RESULT_STATUS)

    def result_count(self):
        return 0 if self.termination_status() == OPTIMIZE_NOT_CALLED else 1

    def solve_time(self):
        return self.solver.tottime

    def objective_value(self):
        s = self.solver
        val = float(s.model.b @ s.y) - s.model.b_const
        return val if self.max_sense else -val

    def dual_objective_value(self):
        s = self.solver
        m = s.model
        val = sum(solvers._cdot(m.C[i], s.X[i]) for i in range(m.nlmi)) - m.b_const
        if m.nlin > 0:
            val += float(m.d_lin @ s.X_lin)
        return val if self.max_sense else -val

    def variable_primal(self):
        return self.solver.y.copy()

    def constraint_dual_psd(self, lmi):
        X = self.solver.X[lmi]
        n = X.shape[0]
        return np.array([X[i, j] for j in range(n) for i in range(j + 1)])

    def constraint_dual_lin(self):
        return self.solver.X_lin.copy()
