/*
 * loraine_hip.h -- C ABI of libloraine_hip.so, the MI355X (gfx950) implementation of the
 * per-IP-iteration linear-algebra hot path of Loraine.jl v0.2.5.
 *
 * The reference has no FFI layer: the boundary is the set of plain Julia calls made by
 * myIPstep / predictor / corrector (src/Solvers.jl:448-478, src/predictor_corrector.jl).
 * Each entry point below names the reference call it replaces (file:line relative to the
 * reference checkout).  INTEGRATION.md shows the Julia `ccall` glue that binds them.
 *
 * Conventions
 *   - return 0 = OK; negative = API / HIP error (text from lrn_last_error);
 *     numerical failure is reported LAPACK-style through `info` out-parameters so the host
 *     can replay the reference's regularisation loops (prepare_W.jl:12-24,
 *     predictor_corrector.jl:59-85).
 *   - FP64, column-major, sparse data in the reference's own format: SparseMatrixCSC
 *     (colptr, rowval, nzval), Int64, 1-based.
 *   - every data pointer may be HOST or DEVICE memory (detected per call); the caller owns
 *     it and it is only accessed during the call.  Device state is owned by the context.
 *   - one context per GPU / per process; calls are blocking and not re-entrant
 *     (the reference is single-threaded).
 */
#ifndef LORAINE_HIP_H
#define LORAINE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrn_ctx lrn_ctx;

/* ---- lifetime ------------------------------------------------------------------------- */
int lrn_create(lrn_ctx** out, int device);
int lrn_destroy(lrn_ctx* ctx);
const char* lrn_last_error(lrn_ctx* ctx);
int lrn_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int lrn_device_count(void);

/* ---- static model data: the outputs of _prepare_A (src/model.jl:120-150) -------------- */
/* AA[ilmi]: nvar x msz^2 CSC, row j = -vec(A_j) (model.jl:199-229).
 * B[ilmi] : nvar x msz CSC rank-one factors (model.jl:176-197) or NULL arrays.
 * sigmaA  : nvar x nlmi (col-major, 1-based), qA: 2 x nlmi (model.jl:153-174).
 * C_lin   : nvar x nlin CSC (MOI_wrapper.jl:145-149) or nlin = 0.                        */
int lrn_upload_model(lrn_ctx* ctx, int nlmi, int nvar, const int64_t* msizes,
                     const int64_t* const* AA_colptr, const int64_t* const* AA_rowval,
                     const double* const* AA_nzval,
                     const int64_t* const* B_colptr, const int64_t* const* B_rowval,
                     const double* const* B_nzval,
                     const int64_t* sigmaA, const int64_t* qA,
                     int nlin, const int64_t* Clin_colptr, const int64_t* Clin_rowval,
                     const double* Clin_nzval);
/* Builder-defined synthetic dense SDP data generated on the device:
 * A_k = (R_k + R_k')/2, R_k iid N(0,1) from a counter-based Philox stream; nlmi = 1. */
int lrn_synthetic_dense_model(lrn_ctx* ctx, int msz, int nvar, uint64_t seed);
/* Completes the synthetic data to a strictly feasible SDP: X0 = I + QQ'/msz,
 * b = AA vec(X0), y0 ~ N(0,1)/sqrt(nvar), S0 = I, C = S0 + mat(AA' y0).  C stays on the device
 * (resident path); b_out[nvar], y0_out[nvar] and ||C||_F are returned. */
int lrn_synthetic_dense_problem(lrn_ctx* ctx, uint64_t seed, double* b_out, double* y0_out, double* normC);
/* dense copy of constraint matrix A_k (0-based k) of block ilmi, msz x msz */
int lrn_get_constraint(lrn_ctx* ctx, int ilmi, int k, double* A_out);
/* tuning knobs (per context): "dense_threshold" (nnz above which a branch-1 constraint takes
 * the MFMA path), "profile" (0/1), "t_batch", "p_batch", "shard_bs" (0 = auto), "jacobi_warm" (0/1), "jacobi_block",
 * "jacobi_inner", "jacobi_wgs", "pivot_boost" (relative pivot level boosted in lrn_schur_factor, 0 = off),
 * "prec_eig" (0 auto / 1 full Jacobi eigendecomposition / 2 Lanczos extremes in lrn_prec_setup),
 * "matvec_sparse" (0 auto -- pattern-restricted mat-vec when every constraint is sparse and M = mat(AA'x) is sparse
 * enough: msz >= 1500, or msz in [256, 1500) with at most 1/12 of M stored -- / 1 dense GEMM mat-vec / 2 pattern-restricted
 * whenever every constraint is sparse),
 * "schur_chol" (dense Schur assembly through the Cholesky factor of W: -1 auto -- H_ij = <L'A_iL, L'A_jL> when
 * every constraint of the block is dense, T_k = L (L'A_kL) L' otherwise, both for msz >= 256 --,
 * 0 never (T_k = W A_k W), 1 as auto without the size threshold, 2 the T_k form only),
 * "schur_plan" (-1 decide locally / 0 / 1: the assembly path every rank of a sharded run agreed on, see lrn_schur_plan),
 * "gemm3_ksplit" (split-K factor of the inner-product GEMM, 0 = auto), "gemm3_sched" (1: regular and masked tiles of
 * GEMM3' in one launch, 0: two launches), "gemm3_tile" (0 auto / 128 / 160), "gemm3_strip" (1: nvar % 128 in (0, 32] -> the last
 * 128 + nvar % 128 rows of H as a launch of 128 x 160 tiles instead of a row of edge tiles), "gemm_no_skip" (1: no block masks
 * in the three GEMMs of the factor path -- A/B switch), "gemm_dyn_masks" (1: the masked K-steps of GEMM1'/2' branch per
 * block as in round 2 -- A/B switch), "gemm1_diag" (1, default: GEMM1' leaves out the blocks of its diagonal tiles that
 * GEMM2' never reads and stores zeros there; 0: computes them -- A/B switch, bit-identical), "gemm_lab" (measurement only:
 * GEMM1'/2' without epilogue / K loop / first load, one workgroup per CU, whole K range, one matrix for every batch
 * element -- tools/gemm12_overhead.py), "gemm3_stagger" (experiment: K-walk stagger of the workgroups of
 * GEMM3' in chunks of 16, 0 = off), "pair_lanes" (lanes per entry of the sparse pair kernel: 0 auto / 4 / 8 / 16 / 64),
 * "jacobi_cross" (1: cross-pair rotations only after round 0), "jacobi_early" (relative level below which a sweep's
 * rotations make it the last one; 0 = always run the confirming sweep), "eigmin_pair" (the two
 * smallest-eigenvalue searches of a step-length computation: 2, default = their Lanczos runs in lock-step, one launch per
 * pair of steps; 1 = as two launch chains on two streams; 0 = one after the other -- same results),
 * "lz_resident" (1, default: the Lanczos steps of these searches and of the H_alpha setup, matrix side <= 1024, as resident
 * launches of 16 / 24 steps -- the workgroup's columns of the matrix in registers, y and the partial dot products exchanged
 * through relaxed agent-scope atomics with no barrier (unwritten words hold a mark), every wait bounded by the wall clock; a launch that
 * gives up sends the context back to one launch per step; 0: one launch per step -- same results bit for bit),
 * "prepw_streams" (1: lrn_prepare_w runs the S side and the Gi solve on a second stream),
 * "nt_mode" (lrn_ip_prepare_w: 1 = NT scaling without singular vectors -- Newton-Schulz square roots of K = L_X'SL_X,
 * Lyapunov solve for the second-order term, SVD route as fallback --, 0 = the reference's SVD route always; lrn_prepare_w
 * with output pointers always takes the SVD route), "ns_l0" (assumed lower end of spec(K)/c of the Newton-Schulz schedule),
 * "ns_maxit", "ns_dual" (-1 auto / 1 / 0: transposed twins of the products from the GEMM epilogue or a transpose pass),
 * "lyap_tol", "lyap_maxit" (relative residual / step limit of the Lyapunov CG),
 * "prec_inv" (H_alpha: -1 auto / 1 / 0: SMW core through an explicit inverse + one refinement step, or two triangular solves),
 * "shard_passes" (multi-GPU, 1: the passes over dense constraint data in AA*vec(.) / mat(AA'x) split by rank),
 * "shard_products" / "shard_products_min" (multi-GPU, 1: the n^3 products of the resident path -- Newton-Schulz, Lyapunov CG,
 * step and right-hand sides -- as column blocks + all-gather from matrix side shard_products_min = 4096 on),
 * "matvec_h" (CG operator of lrn_pcg / lrn_matvec through the ASSEMBLED Schur matrix, one pass over its lower triangle per
 * application: 0 = a static cost model decides once per NT scaling from the CG iterations of the previous one, 1 = never,
 * 2 = always), "prec_dense" (H_alpha inside lrn_pcg as one dense symmetric matrix, nvar <= 8192: 0 cost model / 1 never /
 * 2 always, also in lrn_prec_apply), "pcg_lookahead" (iterations lrn_pcg queues beyond the convergence test it has read,
 * 0..8; the count and the result do not depend on it), "wmw_pattern_min" (right-hand sides AA vec(W M W) with all constraints
 * sparse: from this matrix side on through the pattern entries of W M W), "lyap_form" (second-order term of the corrector:
 * 1 = the better conditioned equivalent Lyapunov equation (Yh/s + s Zh) R + R (.) = C/s + s Zh C Zh, 0 = Yh R + R Yh = C),
 * "ns_lanczos" / "ns_lanczos_min" (1: scale and schedule of the Newton-Schulz iteration from a 24-step Lanczos run on K for
 * blocks of side >= ns_lanczos_min = 1500), "comm_fail_ensure" (test hook: the next exchange of this rank fails its buffer
 * allocation), "reset_timing". */
int lrn_set_option(lrn_ctx* ctx, const char* key, double value);
/* multi-GPU: this context assembles the Schur columns it owns (block-cyclic) */
int lrn_set_shard(lrn_ctx* ctx, int rank, int world);

/* ---- NT scaling: prepare_W (src/prepare_W.jl:28-94) ----------------------------------- */
/* X, S: msz x msz in; outputs may be NULL (kept on the device only).
 * info: 0 ok; 1 = X not PD, 2 = S not PD (host adds 1e-5*I and retries, :5-26). */
int lrn_prepare_w(lrn_ctx* ctx, int ilmi, const double* X, const double* S, double* D, double* G,
                  double* Gi, double* W, double* Si, double* DDsi, int* info);
/* directly set W (and optionally G) -- used when the scaling is produced elsewhere */
int lrn_set_scaling(lrn_ctx* ctx, int ilmi, const double* W, const double* G_or_null);
/* X_lin .* S_lin_inv for the C_lin terms (predictor_corrector.jl:37, Solvers.jl:609) */
int lrn_set_lin(lrn_ctx* ctx, const double* X_lin, const double* S_lin_inv);

/* ---- Schur complement: makeBBBBs / makeBBBB_rank1 (src/makeBBBB.jl:1-218) ------------- */
/* mode 0 = general (makeBBBBs), -1 = rank-one data (makeBBBB_rank1); adds the C_lin term
 * (predictor_corrector.jl:36-38).  H_out (nvar x nvar) may be NULL; when given it receives
 * Matrix(Hermitian(BBBB, :L)) (predictor_corrector.jl:39). */
int lrn_schur_assemble(lrn_ctx* ctx, int mode, double* H_out);
int lrn_schur_get(lrn_ctx* ctx, double* H_out);
/* BBBB + eps*I (predictor_corrector.jl:74) */
int lrn_schur_add_diag(lrn_ctx* ctx, double eps);
/* cholesky(BBBB) (predictor_corrector.jl:57); info > 0: not PD at that column */
int lrn_schur_factor(lrn_ctx* ctx, int* info);
/* dely = L' \ (L \ h) (predictor_corrector.jl:90,199) */
int lrn_schur_solve(lrn_ctx* ctx, const double* h, double* dely);
/* multi-GPU exchange: pack the owned column blocks / unpack the all-gathered buffer */
int64_t lrn_schur_shard_doubles(lrn_ctx* ctx);
int lrn_schur_export_shard(lrn_ctx* ctx, double* buf);
int lrn_schur_import_all(lrn_ctx* ctx, const double* buf_all);
/* multi-GPU, dense data through the Cholesky factor of W: the ranks split the COLUMNS of the matrix variable, so
 * after lrn_schur_assemble every rank holds a partial SUM of the whole Schur matrix (lrn_schur_is_partial_sum = 1)
 * and the exchange is one all-reduce of nvar^2 doubles: export_full -> all-reduce(sum) -> import_full
 * (buffers host or device, position space, lower triangle authoritative).  When it returns 0 the owned column
 * blocks are exchanged with export_shard / all-gather / import_all as above. */
int lrn_schur_is_partial_sum(lrn_ctx* ctx);
/* multi-GPU: which of the two exchanges the next lrn_schur_assemble(mode) of THIS rank would need, from its own view
 * (options, data layout, world size, free device memory): *plan = 1 partial sums + all-reduce, 0 column blocks +
 * all-gather.  Free memory can differ between ranks, so the host all-reduces (MIN) the answers once after
 * lrn_set_shard and pins the result on every rank with lrn_set_option("schur_plan", 0 or 1); a pinned plan is not
 * re-decided by the assembly (a rank that then cannot allocate fails loudly instead of entering another
 * collective).  The reference has no counterpart (single process); this belongs to the sharded loop of
 * src/makeBBBB.jl:77-101. */
int lrn_schur_plan(lrn_ctx* ctx, int mode, int* plan);
int lrn_schur_export_full(lrn_ctx* ctx, double* buf);
int lrn_schur_import_full(lrn_ctx* ctx, const double* buf);

/* ---- right-hand sides: makeRHS (src/makeBBBB.jl:221-228), corrector :186 --------------- */
/* h = Rp + sum AA*vec(W (Rd+S) W) ; Rd_plus_S: msz x msz per block, concatenated */
int lrn_make_rhs(lrn_ctx* ctx, const double* Rp, const double* const* RdS, double* h);

/* ---- CG operator and preconditioners (src/Solvers.jl:572-904) ------------------------- */
/* Ax = sum AA vec(W mat(AA'x) W) + C_lin((X_lin.*S_lin_inv).*(C_lin'x))   (MyA, :582-614).
 * The same linear map is H x for the Schur matrix H of src/makeBBBB.jl:67-218: when lrn_pcg has chosen that form for the
 * current NT scaling (or option "matvec_h" = 2) the library assembles H once per scaling with the kernels of
 * lrn_schur_assemble and applies it as one bandwidth-bound pass over its lower triangle (csrc/hop.hip). */
int lrn_matvec(lrn_ctx* ctx, const double* x, double* Ax);
/* multi-GPU CG operator: this rank's share  AA[:, idx(R_g)] vec((W M W)[R_g,:])  of the mat-vec,
 * R_g = row block `rank` of `world` (lrn_set_shard); the caller all-reduces (sum) the nvar-vector. */
int lrn_matvec_partial(lrn_ctx* ctx, const double* x, double* Ax_partial);
/* prec: 0 none (MyM_no), 1 H_alpha (Prec_for_CG_tilS_prep :674-809), 2 H_beta
 * (Prec_for_CG_beta :624-663).  info > 0: a Cholesky inside the setup failed. */
int lrn_prec_setup(lrn_ctx* ctx, int prec, int erank, int aamat, int* info);
/* Mx = M^{-1} x (MyM :866-904, MyM_beta :670-672, MyM_no :620-622) */
int lrn_prec_apply(lrn_ctx* ctx, const double* x, double* Mx);
/* cg(A, h; tol, maxIter, precon) of ConjugateGradients.jl 0.1 (call sites
 * predictor_corrector.jl:134,235), device-resident: the recurrence is two multi-workgroup launches per iteration, the
 * relative-residual test runs on the device and the host reads it `pcg_lookahead` iterations behind the one it queues
 * (iterations queued beyond the last one leave x untouched: exit code, count and x are those of the loop that tests after
 * every step).  Operator: MyA matrix-free or through the assembled Schur matrix (see lrn_matvec). */
int lrn_pcg(lrn_ctx* ctx, const double* h, double tol, int maxit, double* x, int* exit_code,
            int* iters);

/* ---- device-resident iterate (find_step, check_convergence, RHS) ---
 * The matrix variables X, S, Rd, delX, delS, Xn, Sn, RNT live on the device; only nvar-/nlin-
 * vectors and scalars cross the boundary.  All functions act on every LMI block at once unless
 * they take `ilmi`. */
/* C[ilmi] = -A[ilmi,1] dense msz x msz (src/model.jl:133) */
int lrn_ip_set_c(lrn_ctx* ctx, int ilmi, const double* C);
/* initial point X = Eps*I, S = Eta*I (src/initial_point.jl:33,42) or any iterate */
int lrn_ip_set_iterate(lrn_ctx* ctx, int ilmi, const double* X, const double* S);
int lrn_ip_get_iterate(lrn_ctx* ctx, int ilmi, double* X, double* S);
/* X (which=1) or S (which=2) += eps*I : try_cholesky's regularisation (src/prepare_W.jl:14) */
int lrn_ip_add_diag(lrn_ctx* ctx, int ilmi, int which, double eps);
/* prepare_W on the resident X, S (src/prepare_W.jl:28-94); info as lrn_prepare_w */
int lrn_ip_prepare_w(lrn_ctx* ctx, int ilmi, int* info);
/* out = sum_i AA[i]*vec(X[i])   (src/predictor_corrector.jl:12) */
int lrn_ip_aa_x(lrn_ctx* ctx, double* out);
/* Rd[i] = C[i] - S[i] - mat(AA[i]'y)   (src/predictor_corrector.jl:13) */
int lrn_ip_residual_d(lrn_ctx* ctx, const double* y);
/* out = sum_i AA[i]*vec(W(Rd+S)W)   (makeRHS without Rp, src/makeBBBB.jl:221-228) */
int lrn_ip_rhs_pred(lrn_ctx* ctx, double* out);
/* lrn_ip_aa_x and lrn_ip_rhs_pred in one call: aax_out = sum_i AA[i]*vec(X[i]) (for Rp, src/predictor_corrector.jl:12) and
 * out as lrn_ip_rhs_pred (src/makeBBBB.jl:221-228), with dense constraint data read ONCE for both -- on the 128 GB
 * instance every pass over it costs 25 ms.  The residual Rp is not needed between :12 and :44. */
int lrn_ip_rhs_pred2(lrn_ctx* ctx, double* aax_out, double* out);
/* out = sum_i AA[i]*my_kron(G,G, G'RdG + D - sigma_mu./D - RNT)   (src/predictor_corrector.jl:186) */
int lrn_ip_rhs_corr(lrn_ctx* ctx, double sigma_mu, double* out);
/* delS, delX and the per-block step lengths alpha[nlmi], beta[nlmi]
 * (src/predictor_corrector.jl:248-291; eigmin: Lanczos Ritz value, Cholesky-certified, on the device) */
int lrn_ip_find_step(lrn_ctx* ctx, int predict, double sigma_mu, double tau, const double* dely,
                     double* alpha, double* beta);
/* predict=1: Xn, Sn, RNT with the per-block steps, trXnSn[nlmi] returned (:302-311);
 * predict=0: X, S updated and re-symmetrised with alpha[0], beta[0] (:313-322) */
int lrn_ip_update(lrn_ctx* ctx, int predict, const double* alpha, const double* beta, double* trXnSn);
/* per block 5 numbers: <X,S>, eigmin(X), eigmin(S), ||Rd||_F, <C,X>
 * (find_mu src/Solvers.jl:480-494, check_convergence :496-511) */
int lrn_ip_stats(lrn_ctx* ctx, double* out5);
/* smallest eigenvalue of a symmetric n x n matrix: steps != NULL -> the plain Lanczos Ritz value (unit
 * test of the kernel); steps == NULL -> the Cholesky-certified value lrn_ip_find_step / lrn_ip_stats use
 * (eigmin of src/predictor_corrector.jl:272,285 and src/Solvers.jl:503,505) */
int lrn_dbg_eigmin(lrn_ctx* ctx, int n, const double* M, double* lam, int* steps);
/* k largest eigenpairs (ascending; U_top n x k column-major, may be NULL), smallest eigenvalue and
 * trace of a symmetric matrix: what the preconditioner setup consumes of `eigen(W)`
 * (src/Solvers.jl:642-650,706-722); unit test of the Lanczos path (option "prec_eig") */
int lrn_dbg_lanczos(lrn_ctx* ctx, int n, int k, const double* M, double* lam_top, double* U_top,
                    double* lam_min, double* trace, int* steps);

/* copy one msz x msz (or msz) array of the resident state of block il to `out` (tests of the scaling):
 * name = "W", "Si", "G", "Gi", "D", "DDsi" (valid after the SVD route), "X", "S", "delX", "delS", "RNT",
 * and, after the eigen-free route (option "nt_mode" = 1, the default of lrn_ip_prepare_w; src/prepare_W.jl:28-94
 * without the singular vectors): "LX", "LXi", "LS", "LSi" (Cholesky factors and their inverses), "Yh", "Zh"
 * ((K/c)^1/2, (K/c)^-1/2 for K = L_X' S L_X), "Qm" (G RNT G' of the predictor).  *flag (may be NULL) receives 1 when the
 * current scaling of the block is the eigen-free one, and c = lrn_get_timing("ns_c") its scale. */
int lrn_dbg_get_block(lrn_ctx* ctx, int il, const char* name, double* out, int* flag);
/* Host only (no context, no GPU): k-th smallest eigenvalue (k = 0 .. m-1) of the symmetric tridiagonal matrix with diagonal
 * a[0..m) and off-diagonal b[0..m-1), as the Lanczos drivers of the step-length rule and of the H_alpha setup compute it after
 * every batch of steps (csrc/tridiag.h; reference: eigmin(XXX), src/predictor_corrector.jl:272,285, and eigen(W),
 * src/Solvers.jl:642,706).  upper (may be NULL): a value believed to be >= that eigenvalue (checked, never trusted);
 * width: how far it is expected to lie below (<= 0: unknown).  evals (may be NULL): Sturm counts evaluated. */
int lrn_dbg_tridiag_eig(int m, const double* a, const double* b, int k, const double* upper, double width, double* eig,
                        int64_t* evals);

/* ---- multi-GPU: one process per GPU, the exchange inside the library ---------------------------------------
 * The reference is one process (no MPI/NCCL); a sharded run keeps its predictor / corrector loop
 * (src/predictor_corrector.jl:24-40,119-139) unchanged and only creates a communicator:
 *   rank 0: lrn_comm_unique_id(id) -> the launcher distributes the 128 bytes -> every rank: lrn_comm_init(ctx, id, rank, world)
 * (ncclCommInitRank on the context's device; implies lrn_set_shard).  From then on
 *   lrn_schur_assemble  agrees on the assembly path with the other ranks (first call), assembles the rank's share, reduces
 *                       a status word that EVERY rank enters (a rank that failed makes all ranks return an error, nobody
 *                       is left waiting in a collective) and exchanges on the library's stream: one all-reduce of the
 *                       lower triangle, nvar (nvar + 1) / 2 doubles, on the Cholesky path of dense data; one all-gather
 *                       of the owned column blocks otherwise (makeBBBB.jl:24-36, then the replicated cholesky :57);
 *   lrn_pcg, lrn_matvec apply the operator as this rank's rows of W M W plus one all-reduce of the nvar-vector
 *                       (Solvers.jl:582-614) -- the CG recurrence stays on the device on every rank.
 * lrn_comm_init_host: the same entry points over callbacks that reduce / gather HOST buffers (ranks sharing one GPU,
 * launchers whose fabric is MPI or gloo on the CPU); op: 0 sum, 1 min, 2 max; callbacks return 0 on success.
 * lrn_comm_allreduce: in-place all-reduce of `count` doubles (host or device pointer) on the communicator, for the
 * host loop's own replicated scalars. */
typedef int (*lrn_host_allreduce_fn)(void* user, double* buf, int64_t count, int op);
typedef int (*lrn_host_allgather_fn)(void* user, const double* send, double* recv, int64_t count_per_rank);
int lrn_comm_unique_id(void* id128);
int lrn_comm_init(lrn_ctx* ctx, const void* id128, int rank, int world);
int lrn_comm_init_host(lrn_ctx* ctx, int rank, int world, lrn_host_allreduce_fn allreduce,
                       lrn_host_allgather_fn allgather, void* user);
int lrn_comm_destroy(lrn_ctx* ctx);
int lrn_comm_allreduce(lrn_ctx* ctx, double* buf, int64_t count, int op);

/* ---- measurement ----------------------------------------------------------------------- */
/* milliseconds of the named phase in the last call that ran it, measured with HIP events
 * on the context's stream ("gemm1","gemm2","gemm3","sparse","assemble","factor","solve",...);
 * returns LRN_ERR_ARG for an unknown key.  The reference's TimerOutputs section names are accepted as aliases:
 * "BBBBone1" (mul!(tmp1,W,A_i), src/makeBBBB.jl:87) = gemm1, "BBBBone2" (tmp1*W, :90) = gemm2, "BBBBone3"
 * (AA*vec(tmp), :94) = gemm3 + reduce3, "BBBBone4" (:98, the scatter into BBBB) = 0: it is the GEMM3 epilogue,
 * "BBBBone" (:86), "BBBBthree" (:140) = sparse, "BBBB_rank1" (:2) = rank1, "BBBBs" (:30) = assemble, "Ax"
 * (src/Solvers.jl:583) = matvec, "prec" (:676) = prec_setup, "prep W SVD" (src/prepare_W.jl:37) = prepw_svd,
 * "CG predictor" / "CG corrector" (src/predictor_corrector.jl:130,234) = pcg, "find step corrector" (:243). */
int lrn_get_timing(lrn_ctx* ctx, const char* key, double* ms);
/* launch / event counters of the same phases; "shard_bs" returns the column-block width of the Schur
 * sharding in effect (option "shard_bs": 0 = auto, two 128-aligned blocks per rank) */
int64_t lrn_get_count(lrn_ctx* ctx, const char* key);
/* FP64 MFMA issue-rate probe (TFLOP/s of a register-only v_mfma_f64_16x16x4_f64 loop; an 8 ms warm-up launch and 43 ms
 * timed, so that the clock's ramp after an idle period is not what is measured: 77.3-77.8 on an MI355X) */
int lrn_mfma_f64_peak(lrn_ctx* ctx, double* tflops);
/* placement probe: launches a (nx, 1, nz) grid of 256-thread workgroups and writes, for workgroup
 * (x, z), the XCC (XCD) id the hardware ran it on (HW_REG_XCC_ID) to out[x + nx*z] (host or device int32);
 * with hold_us > 0 every workgroup spins that long so that a whole wave of workgroups is resident at once */
int lrn_xcc_probe(lrn_ctx* ctx, int nx, int nz, int hold_us, int32_t* out);
/* streaming-copy probe: achieved HBM GB/s of a 16 B/lane device copy of `bytes` */
int lrn_hbm_copy_peak(lrn_ctx* ctx, int64_t bytes, double* gbps);

/* ---- building blocks exposed for unit tests (tests/ only) ------------------------------ */
int lrn_dbg_gemm(lrn_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha,
                 const double* A, int lda, const double* B, int ldb, double beta, double* C,
                 int ldc, int flags, int ksplit);
int lrn_dbg_mfma_probe(lrn_ctx* ctx, const double* A16x4, const double* B4x16, double* D16x16);
int lrn_dbg_potrf(lrn_ctx* ctx, int n, double* A, int* info);
int lrn_dbg_potrs(lrn_ctx* ctx, int n, const double* A, const double* b, double* x, int* info);
int lrn_dbg_trsm(lrn_ctx* ctx, int n, int nrhs, int trans, const double* A, double* B, int* info);
int lrn_dbg_svd_jacobi(lrn_ctx* ctx, int n, const double* A, double* U_sigma, double* V,
                       double* sigma, int* sweeps);

#ifdef __cplusplus
}
#endif
#endif /* LORAINE_HIP_H */
