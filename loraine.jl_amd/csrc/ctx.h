// Library context: device-resident model data, NT scaling, Schur matrix, solver state.
#pragma once
#include <map>

#include "chol.h"
#include "lrn_common.h"

namespace lrn { struct Prec; struct Comm; }

struct LmiBlock {
  int msz = 0;
  long nent = 0;            // total stored entries (sparse lists)
  // --- constraint data in sigma-POSITION order (position p <-> constraint sigma[p])
  std::vector<int> sigma;   // position -> constraint (0-based)
  std::vector<int> ipos;    // constraint -> position
  std::vector<long> nnz;    // per position
  int qA = 0;               // reference dense/sparse split (positions < qA are "branch 1")
  int nd = 0;               // positions [0,nd): T_i = W A_i W by MFMA GEMM, A stored dense
  int q_wave = 0;           // positions [nd,q_wave): wave-per-pair kernel; [q_wave,n): thread-per-pair
  int npos_nz = 0;          // positions with nnz > 0 form the prefix [0,npos_nz)
  long p_cap = 0, t_cap = 0; // batch capacities (matrices) of the P / T workspaces for this block
  lrn::DBuf ent_ptr;        // int64 [nvar+1]
  lrn::DBuf ent_r, ent_c;   // int32 [nent]
  lrn::DBuf ent_v;          // double [nent]   value of A_j (= -AA)
  lrn::DBuf Adense;         // double [nd * msz^2], slot s = position s
  int dense_sym = -1;       // every dense constraint matrix is exactly symmetric: -1 not checked yet, 0 no, 1 yes
  lrn::DBuf tri_tab;        // chunk table of the half-traffic passes over the dense data (cgops.hip::TriChunk)
  int tri_nch = 0;
  lrn::DBuf hidx;           // int32 [nvar] position -> row/col index of the Schur matrix
  lrn::DBuf sigma_d, ipos_d; // int32 [nvar] device copies of sigma / ipos
  // stored columns of AA (sparse constraints only) for the deterministic AA'x gather
  long ncq = 0;
  lrn::DBuf cq_q, cq_ptr;   // int64 [ncq], [ncq+1]
  lrn::DBuf cq_j;           // int32 constraint (natural index)
  lrn::DBuf cq_v;           // double AA value
  // sparsity pattern of mat(AA'x) = the stored columns above, when no constraint is stored dense and
  // the pattern is symmetric: sparse-aware mat-vec (cgops.hip, Z = W M W only where AA needs it)
  bool sp_ok = false;
  std::vector<int> sp_long_cols;   // pattern columns with more than 64 stored entries (small-msz sparse mat-vec)
  lrn::DBuf pc_ptr;         // int64 [msz+1] pattern column -> range of stored columns
  lrn::DBuf pc_r;           // int32 [ncq] row of stored column t
  lrn::DBuf pc_t;           // int32 [ncq] stored column holding the transposed entry
  lrn::DBuf ent_t;          // int32 [nent] constraint entry -> stored column
  lrn::DBuf Mv, Zs;         // double [ncq] values of M and of Z on the pattern
  // rank-one factors (datarank = -1): CSR by constraint (natural order)
  bool has_B = false;
  long bnnz = 0;
  lrn::DBuf b_ptr, b_col, b_val;
  // --- NT scaling state (device, msz x msz col-major)
  lrn::DBuf X, S, W, G, Gi, Si, D, DDsi;
  lrn::DBuf Vprev;          // right singular vectors of the previous prepare_W (Jacobi warm start)
  // --- device-resident iterate (ipstep.hip): C, residual, directions, predictor point
  lrn::DBuf Cd, Rd, delX, delS, Xn, Sn, RNT, t0, t1, t2;
  bool resident = false;
  bool have_Vprev = false;
  bool have_W = false, have_G = false;
  // --- eigen-free NT scaling (prepw.hip::prepare_w_ns, option nt_mode = 1): L_X and its transpose, Yh = (K/c)^1/2,
  // Zh = (K/c)^-1/2, Ki = (K/c)^-1 for K = L_X' S L_X; per direction Bs = L_X' dS L_X, TX = L_X^-1 dX L_X^-T;
  // Qm = G RNT G' of the predictor; Lyapunov-CG work; dense copy of the rank-one factors
  lrn::DBuf LXf, LXt, LSf, Yh, Zh, Ki, Bs, TX, Qm, lyap, Bd;
  double ns_c = 1.0;        // the scale c of K (host copy)
  bool nt_free = false;     // the current scaling came from prepare_w_ns: G, Gi, D, DDsi are NOT valid
  bool chol_valid = false;  // LXf, LSf hold the Cholesky factors of the CURRENT X, S (nt_factor; lrn_ip_stats computes them as its
                            // positive-definiteness certificate, the next prepare_w_ns re-uses them)
  bool have_Bd = false;     // dense copy of the rank-one factors (rank-one assembly from W)
};

struct lrn_ctx;
namespace lrn { void update_shard_bs(lrn_ctx* c); }

// lrn_set_option knobs: per context (two contexts of one process -- Julia threads, tests -- do not see each
// other's settings)
struct LrnOptions {
  double dense_threshold = -1.0;  // < 0: cost model decides which constraints are stored dense
  long t_batch = 0, p_batch = 0;  // matrices per T workspace / per GEMM1-2 launch (0 = auto)
  int prec_eig = 0;               // 0 auto (Lanczos for msz >= 256), 1 Jacobi eigendecomposition, 2 Lanczos
  double pivot_boost = 1e-12;     // lrn_schur_factor: pivots <= pivot_boost * diag (rounding noise of a matrix that is
                                  // PSD by construction) are boosted instead of failing, counted in "chol_boosted";
                                  // 0 = strict Cholesky, the literal LAPACK behaviour
  int schur_chol = -1;            // -1 auto, 0 never, 1 whenever the data allows it, 2 T-via-L only
  int schur_plan = -1;            // multi-GPU: the exchange all ranks agreed on (-1 undecided, 0 all-gather of
                                  // Schur column blocks, 1 all-reduce of partial sums); see lrn_schur_plan
  int gemm3_ksplit = 0;           // split-K factor of GEMM3 / GEMM3' (0 = auto)
  int gemm_dyn_masks = 0;         // measurement only: GEMM_DYN_MASKS on GEMM1'/2'
  int gemm_lab = 0;               // measurement only: GEMM_LAB_* bits (<< 20) on GEMM1'/2'; bit 4: every tile walks the whole K
                                  // range (tools/gemm12_overhead.py)
  int gemm1_diag = 1;             // GEMM1': diagonal tiles compute only the blocks GEMM2' reads (GEMM_DIAG_LOWER_Z); 0: all
  int gemm_no_skip = 0;           // measurement only: GEMM1'/2'/3' compute every 16x16 block (GEMM_NO_SKIP)
  int gemm3_sched = 1;            // 1: one launch, regular tiles of every split first, short tiles last; 0: two launches
  int gemm3_strip = 1;            // GEMM3': nd % 128 in (0, 32] -> last tile row of height 128 + nd % 128 (second launch)
  int gemm3_tile = 0;             // workgroup tile of GEMM3': 0 auto, 128, 160
  int gemm3_stagger = 0;          // K-walk stagger of GEMM3' in chunks of 16 (GemmDesc::kstagger)
  int jacobi_inner = 0;           // sweeps over the pair's Gram matrix per round (more did not cut the outer sweeps: 1)
  int jacobi_wgs = 0;             // workgroups per Gram / apply launch the row chunking aims for: 0 auto
  int jacobi_block = 0;           // column block width: 0 auto (32 for n >= 5000), 16, 32
  int jacobi_cross = 1;           // block Jacobi: cross-pair rotations only outside round 0 of a sweep
  int prepw_streams = 1;          // prepare_W: the S side and the Gi solve on a second stream beside cholesky(X) / the SVD / the GEMMs
  int lz_resident = 1;            // Lanczos steps of the eigmin searches (n <= 1024) as resident launches of 16 steps: M in registers, relaxed-atomic exchange
  int eigmin_pair = 2;            // the two eigmin calls of a step-length search: 2 = one launch per pair of Lanczos steps, 1 = two streams, 0 = one after the other
  double jacobi_early = 3e-8;     // a sweep whose rotated column pairs were all closer to orthogonal than this ends the SVD
  bool jacobi_warm = true;
  int shard_passes = 1;           // multi-GPU: split AA*vec(.) and mat(AA'x) over dense constraints by rank (+ one all-reduce)
  int prec_inv = -1;              // H_alpha: SMW core applied through an explicit inverse + one refinement step (1), the two
                                  // triangular solves (0), auto (-1: explicit from k msz = 256 on)
  int nt_mode = 1;                // lrn_ip_prepare_w: 1 = eigen-free NT scaling (Newton-Schulz square roots of K = L_X'SL_X, Lyapunov
                                  // solve for the second-order term; falls back to the SVD when it does not converge), 0 = SVD always
  double ns_l0 = 2e-3;            // Newton-Schulz schedule: assumed lower end of spec(K)/c (slower, never wrong, when cond(K) is larger)
  int ns_lanczos = 1;             // Newton-Schulz: scale c and schedule end l from a 24-step Lanczos run on K (blocks of side >= ns_lanczos_min)
  int ns_lanczos_min = 1500;
  int ns_dual = -1;               // Newton-Schulz: transposed twins from the GEMM epilogue (1), a transpose pass (0), auto (-1)
  int ns_maxit = 40;              // Newton-Schulz steps before the SVD fallback
  double lyap_tol = 1e-12;        // relative residual of the Lyapunov CG (second-order term of the corrector)
  int lyap_maxit = 300;
  int lyap_form = 1;              // 1: the better conditioned equivalent equation (Yh/s + s Zh) R + R (.) = C/s + s Zh C Zh, 0: Yh R + R Yh = C
  int pair_lanes = 0;             // pair_wave_kernel: lanes per Schur entry, 0 auto (16 for short products), 16, 64
  int matvec_sparse = 0;          // 0 auto, 1 dense GEMM path, 2 sparse path whenever the pattern allows
  int shard_products = 1;         // multi-GPU: the n^3 products of the resident path (Newton-Schulz, Lyapunov CG, step) by
  int shard_products_min = 4096;  // column blocks + all-gather from this matrix side on (one product of 10^4: 31 ms; below, the
                                  // all-gather costs more than the product)
  int prec_dense = 0;             // H_alpha inside lrn_pcg as ONE dense symmetric matrix M^-1 (nvar <= 8192): 0 auto (cost model), 1 never, 2 always
                                  // (also in lrn_prec_apply)
  int wmw_pattern_min = 1500;     // right-hand sides AA vec(W M W) with all constraints sparse: from this side on through the
                                  // pattern entries of W M W (one n^3 product instead of two)
  int profile_symv = 0;           // measurement: time every application of H x by itself (synchronises: not for solves)
  int matvec_h = 0;               // CG operator through the assembled Schur matrix (hop.hip): 0 auto (cost model), 1 never
                                  // (the matrix-free MyA always), 2 always
  int pcg_lookahead = 2;          // lrn_pcg: iterations the host queues beyond the one whose convergence test it has read
};

struct lrn_ctx {
  int device = 0;
  LrnOptions opt;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;      // second Lanczos run of eigmin_certified_pair (created on first use)
  std::string err;
  int nlmi = 0, nvar = 0, nlin = 0;
  std::vector<LmiBlock> lmi;
  bool pos_space = false;       // Schur matrix kept in sigma-position space (nlmi == 1)
  // linear block C_lin (nvar x nlin) stored by linear constraint (CSC)
  lrn::DBuf cl_ptr, cl_row, cl_val;   // int64 [nlin+1], int32 (Schur-matrix index), double
  lrn::DBuf cl_rown;                  // int32 natural row index
  lrn::DBuf lin_xs;                   // X_lin .* S_lin_inv  [nlin]
  // deterministic gathers (no floating-point atomics): target entries of H with their contributions
  // (C_lin diag(xs) C_lin' term), and C_lin by rows (mat-vec / diagonal of the linear term)
  long lp_n = 0;                      // distinct lower-triangle targets
  lrn::DBuf lp_r, lp_c, lp_ptr, lp_l, lp_w;   // int32 row, col (Schur index); int64 ptr; int32 l; double C[i,l]*C[j,l]
  lrn::DBuf cr_ptr, cr_col, cr_val;   // CSR by natural row: int64 [nvar+1], int32 l, double
  // Schur complement
  lrn::DBuf H;          // assembled (lower triangle authoritative), nvar x nvar
  lrn::DBuf L;          // factor
  lrn::DBuf Linv;       // inverse diagonal blocks
  lrn::DBuf cholwork;   // nvar * NB
  lrn::DBuf info_dev;   // int
  lrn::DBuf v0, v1, v2, v3;   // nvar-vectors (solve scratch)
  lrn::DBuf hdiag;            // diag(H) before the factorisation (pivot boosting)
  bool have_H = false, have_L = false;
  bool H_shifted = false;     // lrn_schur_add_diag since the last assembly: strict Cholesky only
  bool H_partial = false;     // world > 1, Cholesky path: H is this rank's partial SUM (exchange = all-reduce)
  bool H_owned_only = false;  // world > 1: H holds only the column blocks this rank assembled (CG operator, hop.hip)
  // which NT scaling (W of every block, X_lin ./ S_lin) the assembled H belongs to: scal_version counts the changes of the
  // scaling, H_version is its value at the last assembly, H_mode the mode of that assembly (0 general, -1 rank-one)
  long scal_version = 0, H_version = -1;
  int H_mode = 0;
  // CG operator through the assembled matrix (hop.hip): the decision taken for scaling `hop_version`, the operator
  // applications counted under the current and the previous scaling (input of the cost model)
  long hop_version = -1;
  bool hop_use = false;
  long cg_cur_iters = 0, cg_prev_iters = 0;
  lrn::DBuf hopbuf;           // partial sums of the triangular mat-vec
  bool lz_no_persist = false; // a resident multi-step Lanczos launch timed out at a barrier once: per-step launches from then on
  double* pin = nullptr;      // host-mapped words the CG kernels write their exit code to (lrn_pcg), and their device address
  double* pin_dev = nullptr;
  lrn::DBuf cgpart;           // partial sums of the CG recurrence kernels
  hipEvent_t pcg_ev[16] = {};
  // assembly workspaces
  lrn::DBuf P, P2, T, slabs, Hd, BG;
  lrn::DBuf m0, m1, m2, cgbuf;   // msz^2 work matrices (mat-vec / rhs), PCG vectors
  int T_m = 0;                 // matrix side and block the T workspace was last laid out for
  const void* T_owner = nullptr;
  int T_layout = 0;            // 0: msz^2 per matrix, lower tiles (T_k = W A_k W); 1: packed lower tiles (L' A_k L)
  lrn::DBuf wchol;             // Cholesky path of the assembly: factor of W, its transpose, work
  // shard (multi-GPU): this rank assembles owner columns with (pos / shard_bs) % world == rank
  int rank = 0, world = 1, shard_bs = 128;
  int shard_bs_opt = 0;         // option "shard_bs": 0 = auto (see update_shard_bs)
  // timing of the last assembly / factor / solve (ms, HIP events on ctx stream)
  std::map<std::string, double> timing;
  std::map<std::string, long> counts;
  bool profile = true;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t evA = nullptr, evB = nullptr;      // stream <-> stream2 dependencies (prepare_w_block)
  hipStream_t stream3 = nullptr;                // Newton-Schulz: the T Z product beside Y T (prepare_w_ns)
  hipEvent_t evC = nullptr, evD = nullptr;
  // generic scratch
  lrn::DBuf scratch, jscratch, redbuf, redout, lzbuf, lzbuf2, lxbuf, ezbuf;
  lrn::DBuf triw;               // tri-weights of the matrices the dense passes multiply with (2 msz^2)
  lrn::DBuf commvec, commmat;   // multi-GPU: partial results of the sharded passes over dense constraint data
  // preconditioner / CG state
  lrn::Prec* prec = nullptr;
  // multi-GPU communicator (comm.hip; lrn_comm_init / lrn_comm_init_host)
  lrn::Comm* comm = nullptr;
};

namespace lrn {
int set_error(lrn_ctx* c, int code, const char* fmt, ...);
int ensure(lrn_ctx* c, DBuf& b, size_t bytes, bool zero = false);
void release(DBuf& b);
bool is_device_ptr(const void* p);
int copy_in(lrn_ctx* c, void* dst_dev, const void* src, size_t bytes);    // src host or device
int copy_out(lrn_ctx* c, void* dst, const void* src_dev, size_t bytes);   // dst host or device
void tic(lrn_ctx* c);
void toc(lrn_ctx* c, const char* key);

// comm.hip
int comm_allreduce(lrn_ctx* c, double* buf_dev, long count, int op);     // in place on c->stream; op 0 sum, 1 min, 2 max
int comm_allgather(lrn_ctx* c, const double* send_dev, double* recv_dev, long count);
int comm_allgather_cols(lrn_ctx* c, double* C_dev, int n, int cb);   // column blocks of width cb of an n x n matrix, in place
int comm_schur_exchange(lrn_ctx* c, int rc_local, bool gather_blocks = true);
int comm_agree_plan(lrn_ctx* c, int mode);
int comm_status_max(lrn_ctx* c, double* words, int nw);
void comm_free(lrn_ctx* c);
void comm_inject_ensure_failure(lrn_ctx* c);
// schur.hip
int schur_assemble(lrn_ctx* c, int mode);
int schur_plan(lrn_ctx* c, int mode);
int schur_factor(lrn_ctx* c, int* info);
int schur_solve(lrn_ctx* c, const double* h, double* dely);
int schur_add_diag(lrn_ctx* c, double eps);
int schur_get(lrn_ctx* c, double* Hout);
// hop.hip: the CG operator through the assembled matrix
// y = H x; qpart (may be null): receives *nq partial sums of x'y (*nq = 0: not formed, e.g. sharded)
int hop_apply(lrn_ctx* c, const double* x_dev, double* y_dev, double* qpart = nullptr, int* nq = nullptr);
int symv_lower(lrn_ctx* c, const double* A_dev, int n, const int* idx, const double* x_dev, double* y_dev,
               double* qpart = nullptr, int* nq = nullptr, bool sharded = false);
bool hop_worthwhile(lrn_ctx* c, long expected_iters);
int hop_prepare(lrn_ctx* c);
}  // namespace lrn

#define LRN_HIP(c, expr)                                                                   \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess)                                                                 \
      return lrn::set_error((c), LRN_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                            hipGetErrorString(e__), __FILE__, __LINE__);                   \
  } while (0)
#define LRN_TRY(expr)             \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != LRN_OK) return rc__; \
  } while (0)
