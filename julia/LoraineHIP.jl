# LoraineHIP.jl -- Julia-side glue that binds libloraine_hip.so (include/loraine_hip.h) behind
# the reference's own function names, so `Loraine.Solvers.predictor/corrector` run unchanged.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia toolchain (SURVEY.md
# section 0, fact 5).  This is the binding a Loraine.jl maintainer adds; the Python host in
# `loraine.jl_amd/solvers.py` drives exactly the same C entry points and is what the parity
# tests and the benchmark run.  See INTEGRATION.md.
#
# Usage (inside Loraine.jl, after `include("Solvers.jl")`):
#     include("LoraineHIP.jl"); LoraineHIP.enable!(solver)     # once per `load`
module LoraineHIP

using SparseArrays, LinearAlgebra

const LIB = get(ENV, "LORAINE_HIP_LIB", joinpath(@__DIR__, "..", "loraine.jl_amd", "libloraine_hip.so"))

mutable struct Ctx
    h::Ptr{Cvoid}
    chol_is_object::Bool      # this IP iteration's factor came out of the +1e-4*I loop (predictor_corrector.jl:85)
end

function check(ctx::Ctx, rc::Cint, what)
    rc == 0 && return
    msg = unsafe_string(ccall((:lrn_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx.h))
    error("$what failed ($rc): $msg")
end

function Ctx(device::Integer = 0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:lrn_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), r, device)
    rc == 0 || error("lrn_create failed ($rc): no MI355X visible? (there is no CPU fallback)")
    ctx = Ctx(r[], false)
    finalizer(c -> ccall((:lrn_destroy, LIB), Cint, (Ptr{Cvoid},), c.h), ctx)
    return ctx
end

# ---- static data: MyModel (src/model.jl:34-87) -> device, once per `load` -------------------
function upload_model!(ctx::Ctx, model)
    nlmi, n = model.nlmi, model.n
    colptr = [Vector{Int64}(model.AA[i].colptr) for i in 1:nlmi]     # SparseMatrixCSC is already
    rowval = [Vector{Int64}(model.AA[i].rowval) for i in 1:nlmi]     # (colptr,rowval,nzval) 1-based
    nzval = [Vector{Float64}(model.AA[i].nzval) for i in 1:nlmi]
    hasB = !isempty(model.B)
    bcol = hasB ? [Vector{Int64}(model.B[i].colptr) for i in 1:nlmi] : Vector{Int64}[]
    brow = hasB ? [Vector{Int64}(model.B[i].rowval) for i in 1:nlmi] : Vector{Int64}[]
    bval = hasB ? [Vector{Float64}(model.B[i].nzval) for i in 1:nlmi] : Vector{Float64}[]
    # The arrays of pointers are locals that `ccall` itself roots (passed as the Vector, converted to
    # Ptr{Ptr{T}} by the call), and the arrays they point into are kept alive by GC.@preserve: nothing the
    # library reads can be collected or moved during the call.  An empty list (no rank-one factors) is C_NULL.
    cp_ptrs, rv_ptrs, nz_ptrs = map(pointer, colptr), map(pointer, rowval), map(pointer, nzval)
    bc_ptrs, br_ptrs, bv_ptrs = map(pointer, bcol), map(pointer, brow), map(pointer, bval)
    cl = model.C_lin
    msizes, sigmaA, qA = Vector{Int64}(model.msizes), Matrix{Int64}(model.sigmaA), Matrix{Int64}(model.qA)
    lcp, lrv, lnz = Vector{Int64}(cl.colptr), Vector{Int64}(cl.rowval), Vector{Float64}(cl.nzval)
    GC.@preserve colptr rowval nzval bcol brow bval cp_ptrs rv_ptrs nz_ptrs bc_ptrs br_ptrs bv_ptrs msizes sigmaA qA lcp lrv lnz begin
        rc = ccall((:lrn_upload_model, LIB), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}},
             Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Int64},
             Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}),
            ctx.h, nlmi, n, msizes, cp_ptrs, rv_ptrs, nz_ptrs,
            hasB ? bc_ptrs : C_NULL, hasB ? br_ptrs : C_NULL, hasB ? bv_ptrs : C_NULL, sigmaA, qA,
            model.nlin, lcp, lrv, lnz)
    end
    check(ctx, rc, "lrn_upload_model")
end

# ---- prepare_W (src/prepare_W.jl:28-94) -------------------------------------------------------
function prepare_W(ctx::Ctx, solver)
    for i in 1:solver.model.nlmi
        m = solver.model.msizes[i]
        tries = 0
        while true
            info = Ref{Cint}(0)
            rc = ccall((:lrn_prepare_w, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Cint}),
                ctx.h, i - 1, solver.X[i], solver.S[i], solver.D[i], solver.G[i], solver.Gi[i],
                solver.W[i], solver.Si[i], solver.DDsi[i], info)
            check(ctx, rc, "lrn_prepare_w")
            info[] == 0 && break
            # try_cholesky's regularisation loop (prepare_W.jl:12-24), replayed from `info`
            M = info[] == 1 ? solver.X : solver.S
            M[i] += 1e-5 .* I(m)
            tries += 1
            if tries > 1000
                solver.status = 4
                return
            end
        end
    end
    solver.Si_lin = solver.model.nlin > 0 ? 1.0 ./ solver.S_lin : []
    return solver.D, solver.G, solver.Gi, solver.W, solver.Si, solver.DDsi, solver.Si_lin
end

# ---- makeBBBBs / makeBBBB_rank1 (src/makeBBBB.jl) + cholesky + solves ------------------------
function makeBBBB!(ctx::Ctx, solver; want_matrix::Bool = false)
    if solver.model.nlin > 0
        check(ctx, ccall((:lrn_set_lin, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
            ctx.h, vec(solver.X_lin), vec(solver.S_lin_inv)), "lrn_set_lin")
    end
    mode = solver.datarank == -1 ? -1 : 0
    H = want_matrix ? Matrix{Float64}(undef, solver.model.n, solver.model.n) : nothing
    check(ctx, ccall((:lrn_schur_assemble, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}),
        ctx.h, mode, want_matrix ? H : C_NULL), "lrn_schur_assemble")
    return H
end

function factor!(ctx::Ctx, solver)          # predictor_corrector.jl:55-85
    info = Ref{Cint}(0)
    ctx.chol_is_object = false
    check(ctx, ccall((:lrn_schur_factor, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), ctx.h, info), "lrn_schur_factor")
    info[] == 0 && return true
    solver.regcount += 1
    solver.regcount > 5 && (solver.status = 3; return false)
    for _ in 1:1001
        check(ctx, ccall((:lrn_schur_add_diag, LIB), Cint, (Ptr{Cvoid}, Cdouble), ctx.h, 1e-4), "lrn_schur_add_diag")
        check(ctx, ccall((:lrn_schur_factor, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), ctx.h, info), "lrn_schur_factor")
        # :85 stores the Cholesky OBJECT here (the factor L otherwise, :57-58): `cholBBBB' \ (cholBBBB \ h)` then
        # solves twice until the next factorisation -- kept, so the iterates stay the reference's
        info[] == 0 && (ctx.chol_is_object = true; return true)
    end
    solver.status = 3
    return false
end

function solve(ctx::Ctx, h::Vector{Float64})   # cholBBBB' \ (cholBBBB \ h)  (predictor_corrector.jl:90,199)
    x = similar(h)
    check(ctx, ccall((:lrn_schur_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.h, h, x), "lrn_schur_solve")
    if ctx.chol_is_object                       # regularised iteration: H_reg^-1 (H_reg^-1 h), as the reference
        y = similar(h)
        check(ctx, ccall((:lrn_schur_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.h, x, y), "lrn_schur_solve")
        return y
    end
    return x
end

function makeRHS(ctx::Ctx, solver)             # makeBBBB.jl:221-228
    mats = [solver.Rd[i] + solver.S[i] for i in 1:solver.model.nlmi]
    h = Vector{Float64}(undef, solver.model.n)
    GC.@preserve mats begin
        ptrs = map(pointer, mats)
        check(ctx, ccall((:lrn_make_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Ptr{Float64}}, Ptr{Float64}),
            ctx.h, vec(solver.Rp), ptrs, h), "lrn_make_rhs")
    end
    return h
end

# ---- operator protocol consumed by cg (Solvers.jl:582,620,670,866) ---------------------------
struct MyA_hip;  ctx::Ctx; end
struct MyM_hip;  ctx::Ctx; end
function (t::MyA_hip)(Ax::Vector{Float64}, x::Vector{Float64})
    check(t.ctx, ccall((:lrn_matvec, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), t.ctx.h, x, Ax), "lrn_matvec")
end
function (t::MyM_hip)(Mx::Vector{Float64}, x::Vector{Float64})
    check(t.ctx, ccall((:lrn_prec_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), t.ctx.h, x, Mx), "lrn_prec_apply")
end

# Prec_for_CG_tilS_prep (prec = 1) / Prec_for_CG_beta (prec = 2 or 4) / none (0)
function prec_setup!(ctx::Ctx, solver)
    kind = solver.preconditioner == 1 ? 1 : (solver.preconditioner in (2, 4) ? 2 : 0)
    info = Ref{Cint}(0)
    check(ctx, ccall((:lrn_prec_setup, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ref{Cint}),
        ctx.h, kind, solver.erank, solver.aamat, info), "lrn_prec_setup")
    info[] == 0 || throw(PosDefException(info[]))
end

# whole PCG on the device: replaces cg(A, h; tol, maxIter, precon) at predictor_corrector.jl:134,235
function cg(ctx::Ctx, h::Vector{Float64}; tol::Float64, maxIter::Int = 10000)
    x = similar(h); ec = Ref{Cint}(0); it = Ref{Cint}(0)
    check(ctx, ccall((:lrn_pcg, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cdouble, Cint, Ptr{Float64}, Ref{Cint}, Ref{Cint}),
        ctx.h, h, tol, maxIter, x, ec, it), "lrn_pcg")
    return x, Int(ec[]), Int(it[])
end


# ---- device-resident iterate (SURVEY.md 8f ranks 1-3): X, S, delX, delS, Xn, Sn, RNT never leave HBM ----
# What loraine.jl_amd/resident.py does; only nvar-vectors and scalars cross the boundary.
set_C!(ctx::Ctx, i::Int, C::Matrix{Float64}) =                      # once, C = -A[i,1]  (model.jl:133)
    check(ctx, ccall((:lrn_ip_set_c, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), ctx.h, i - 1, C), "lrn_ip_set_c")
set_iterate!(ctx::Ctx, i::Int, X::Matrix{Float64}, S::Matrix{Float64}) =   # initial_point.jl:33,42
    check(ctx, ccall((:lrn_ip_set_iterate, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}), ctx.h, i - 1, X, S), "lrn_ip_set_iterate")
function get_iterate(ctx::Ctx, i::Int, m::Int)                       # results: constraint duals, dual objective
    X = Matrix{Float64}(undef, m, m); S = similar(X)
    check(ctx, ccall((:lrn_ip_get_iterate, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}), ctx.h, i - 1, X, S), "lrn_ip_get_iterate")
    return X, S
end

function prepare_W_resident!(ctx::Ctx, solver)                       # prepare_W.jl:5-94 incl. try_cholesky's loop
    for i in 1:solver.model.nlmi
        info = Ref{Cint}(0)
        for _ in 1:1001
            check(ctx, ccall((:lrn_ip_prepare_w, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}), ctx.h, i - 1, info), "lrn_ip_prepare_w")
            info[] == 0 && break
            check(ctx, ccall((:lrn_ip_add_diag, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cdouble), ctx.h, i - 1, info[], 1e-5), "lrn_ip_add_diag")
        end
    end
end

function residuals!(ctx::Ctx, solver)                                # predictor_corrector.jl:12-13
    aax = Vector{Float64}(undef, solver.model.n)
    check(ctx, ccall((:lrn_ip_aa_x, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, aax), "lrn_ip_aa_x")
    check(ctx, ccall((:lrn_ip_residual_d, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, vec(solver.y)), "lrn_ip_residual_d")
    solver.Rp = solver.model.b - aax            # minus C_lin * X_lin on the host when nlin > 0
end

function rhs_pred(ctx::Ctx, n::Int)                                  # makeRHS without Rp (makeBBBB.jl:221-228)
    h = Vector{Float64}(undef, n)
    check(ctx, ccall((:lrn_ip_rhs_pred, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, h), "lrn_ip_rhs_pred")
    return h
end
function rhs_pred2(ctx::Ctx, n::Int)                                 # AA*vec(X) (:12) and makeRHS (:44): dense data read once
    aax = zeros(n); h = zeros(n)
    check(ctx, ccall((:lrn_ip_rhs_pred2, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.h, aax, h), "lrn_ip_rhs_pred2")
    return aax, h
end
function rhs_corr(ctx::Ctx, n::Int, sigma_mu::Float64)               # predictor_corrector.jl:186
    h = Vector{Float64}(undef, n)
    check(ctx, ccall((:lrn_ip_rhs_corr, LIB), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}), ctx.h, sigma_mu, h), "lrn_ip_rhs_corr")
    return h
end

# find_step (predictor_corrector.jl:248-326): directions and per-block step lengths on the device; the scalar
# rule  min([alpha; alpha_lin])  and the y / X_lin / S_lin updates stay in Julia
function find_step!(ctx::Ctx, solver)
    nl = solver.model.nlmi
    alpha = Vector{Float64}(undef, nl); beta = similar(alpha)
    check(ctx, ccall((:lrn_ip_find_step, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, solver.predict ? 1 : 0, solver.sigma * solver.mu, solver.tau, vec(solver.dely), alpha, beta), "lrn_ip_find_step")
    solver.alpha, solver.beta = alpha, beta
    tr = Vector{Float64}(undef, nl)
    if solver.predict
        check(ctx, ccall((:lrn_ip_update, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx.h, 1, alpha, beta, tr), "lrn_ip_update")
        return tr                                # tr(Xn Sn) per block for sigma_update (Solvers.jl:513-540)
    end
    a = [minimum([alpha; solver.alpha_lin])]; b = [minimum([beta; solver.beta_lin])]
    check(ctx, ccall((:lrn_ip_update, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx.h, 0, a, b, tr), "lrn_ip_update")
    solver.y .+= b[1] .* solver.dely
    return tr
end

# per block: <X,S>, eigmin(X), eigmin(S), ||Rd||_F, <C,X>   (find_mu, check_convergence: Solvers.jl:480-511)
function stats(ctx::Ctx, nlmi::Int)
    out = Matrix{Float64}(undef, 5, nlmi)
    check(ctx, ccall((:lrn_ip_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, out), "lrn_ip_stats")
    return out
end

# ---- multi-GPU: one process per GPU, the exchange INSIDE the library (csrc/comm.hip) --------------
# `predictor` / `corrector` stay as they are: after `comm_init!` the calls above (`makeBBBB!`, `cg`, `MyA_hip`) assemble
# this rank's share, agree on the path, check every rank's outcome and exchange over RCCL on the library's stream.
#   using MPI; MPI.Init(); comm = MPI.COMM_WORLD; r = MPI.Comm_rank(comm); P = MPI.Comm_size(comm)
#   ctx = Ctx(r % ngpus_per_node)
#   id = r == 0 ? comm_unique_id() : Vector{UInt8}(undef, 128); MPI.Bcast!(id, 0, comm)
#   comm_init!(ctx, id, r, P)
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:lrn_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("lrn_comm_unique_id failed ($rc)")
    return id
end
comm_init!(ctx::Ctx, id::Vector{UInt8}, rank::Int, world::Int) =
    check(ctx, ccall((:lrn_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), ctx.h, id, rank, world), "lrn_comm_init")
comm_destroy!(ctx::Ctx) = check(ctx, ccall((:lrn_comm_destroy, LIB), Cint, (Ptr{Cvoid},), ctx.h), "lrn_comm_destroy")
# replicated scalars of the host loop (op: 0 sum, 1 min, 2 max), e.g. a sanity check that all ranks hold the same mu
function comm_allreduce!(ctx::Ctx, v::Vector{Float64}, op::Int = 0)
    check(ctx, ccall((:lrn_comm_allreduce, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Cint), ctx.h, v, length(v), op), "lrn_comm_allreduce")
    return v
end

# ---- the pieces of the exchange, for a host that brings its own collectives (ROCm-aware MPI): SURVEY.md 8e ----
set_shard!(ctx::Ctx, rank::Int, world::Int) =
    check(ctx, ccall((:lrn_set_shard, LIB), Cint, (Ptr{Cvoid}, Cint, Cint), ctx.h, rank, world), "lrn_set_shard")
shard_doubles(ctx::Ctx) = ccall((:lrn_schur_shard_doubles, LIB), Int64, (Ptr{Cvoid},), ctx.h)
# buf / buf_all are DEVICE pointers (ROCm-aware MPI_Allgather or RCCL between the two calls)
export_shard!(ctx::Ctx, buf::Ptr{Float64}) =
    check(ctx, ccall((:lrn_schur_export_shard, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, buf), "lrn_schur_export_shard")
import_all!(ctx::Ctx, buf_all::Ptr{Float64}) =
    check(ctx, ccall((:lrn_schur_import_all, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, buf_all), "lrn_schur_import_all")
# dense data on several GPUs: every rank holds a partial SUM of the Schur matrix (the ranks split the columns of the
# matrix variable); exchange = export_full! -> all-reduce(sum) of nvar^2 doubles -> import_full!
is_partial_sum(ctx::Ctx) = ccall((:lrn_schur_is_partial_sum, LIB), Cint, (Ptr{Cvoid},), ctx.h) != 0
export_full!(ctx::Ctx, buf::Ptr{Float64}) =
    check(ctx, ccall((:lrn_schur_export_full, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, buf), "lrn_schur_export_full")
import_full!(ctx::Ctx, buf::Ptr{Float64}) =
    check(ctx, ccall((:lrn_schur_import_full, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.h, buf), "lrn_schur_import_full")
function matvec_partial!(ctx::Ctx, Ax::Vector{Float64}, x::Vector{Float64})   # caller all-reduces Ax
    check(ctx, ccall((:lrn_matvec_partial, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx.h, x, Ax), "lrn_matvec_partial")
end

end # module
