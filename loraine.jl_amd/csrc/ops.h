// Shared device-side operations (cgops.hip / prepw.hip) used by the resident IP step.
#pragma once
#include "ctx.h"
namespace lrn {
bool use_sparse_matvec(const lrn_ctx* c, const LmiBlock& b);                // the pattern-restricted form of MyA serves this block
int ensure_m(lrn_ctx* c, int m);                                            // c->m0..m2 >= msz^2
int wmw(lrn_ctx* c, LmiBlock& b, double* M, double* P, double* Z);          // Z = W M W (M symmetric)
int aa_times(lrn_ctx* c, LmiBlock& b, const double* Z, double* y);          // y += AA vec(Z)
// y += AA vec(W M W) through the entries of W M W the (all sparse) constraints read: N (work) = M W, then dots on the pattern
bool wmw_pattern_ok(const lrn_ctx* c, const LmiBlock& b);
int aa_times_wmw_pattern(lrn_ctx* c, LmiBlock& b, const double* M, double* N, double* y);
int aa_times2(lrn_ctx* c, LmiBlock& b, const double* Z1, double* y1, const double* Z2, double* y2);   // both, one pass over dense data
int aat_to_mat(lrn_ctx* c, LmiBlock& b, const double* x, double* M);        // M = mat(AA' x)
int prepare_w_block(lrn_ctx* c, LmiBlock& b, int* info);                    // NT scaling from b.X, b.S (SVD route)
// eigen-free NT scaling from b.X, b.S: W, Si, the Cholesky factors and K^(+-1/2); *converged = false: nothing usable, take
// the SVD route
int prepare_w_ns(lrn_ctx* c, LmiBlock& b, int* info, bool* converged);
// Cholesky factors of b.X, b.S into b.LXf, b.LSf (two streams); info = 0, 1 (X not PD), 2 (S not PD) as prepare_W.jl:33-34;
// minpiv[2] (may be null): smallest pivots L_ii^2 (upper bounds of the smallest eigenvalues).  Sets b.chol_valid.
int nt_factor(lrn_ctx* c, LmiBlock& b, int* info, double* minpiv);
// C = alpha A Bm' (n x n, column-major): the arrangement the direct-to-LDS GEMM kernel takes
int gemm_nt(hipStream_t st, int n, const double* A, const double* Bm, double* C, int flags = 0, double alpha = 1.0,
            double* Ct = nullptr);     // Ct: the transposed result as well
// the same, a mid-size product left as its split-K slabs for a consumer that adds them while it reads (lrn_common.h, SlabSrc;
// src->n == 1: the product is in C)
int gemm_nt_slabs(hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha, SlabSrc* src);
// C and its transposed twin Ct from one pass over the slabs of a product (or over C itself when the product was not split)
void slabs_to_c_and_ct(hipStream_t st, const SlabSrc& src, int n, double* C, double* Ct);
// the same for a product that is symmetric in exact arithmetic; C comes back exactly symmetric
int gemm_nt_sym(hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha = 1.0, int tri = 0);
// The same products for the resident path of a sharded run (one process per GPU): when the communicator has more than one
// rank, `st` is the context's stream and n >= option shard_products_min, this rank computes its block of columns and the
// blocks are all-gathered in place (csrc/comm.hip) -- every rank ends with the same bits; otherwise the plain product.
bool products_sharded(const lrn_ctx* c, hipStream_t st, int n);
// tri: GEMM_KFROM_M / GEMM_KFROM_N / GEMM_KTO_M / GEMM_KTO_N when op(A) / op(B) is triangular with stored zeros (prepw.hip)
int pgemm_nt(lrn_ctx* c, hipStream_t st, int n, const double* A, const double* Bm, double* C, int tri = 0,
             double alpha = 1.0, double* Ct = nullptr);
int pgemm_nt_sym(lrn_ctx* c, hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha = 1.0,
                 int tri = 0);
// k largest eigenpairs (ascending), smallest eigenvalue and trace of a dense symmetric matrix
int lanczos_extremes(lrn_ctx* c, const double* M, int n, int k, double* lam_top, double* U_top, double* lam_min,
                     double* trace, int* steps_out);
// both ends of the spectrum from nsteps plain Lanczos steps (ipstep.hip): lo >= lambda_min, an eigenvalue within res_hi of hi
int lanczos_ends(lrn_ctx* c, const double* M, int n, int nsteps, double* lo, double* hi, double* res_hi);
// single-launch Lanczos steps [j0, j1) keeping every q_j (ipstep.hip); PA2: 2 * ceil(n / 16) doubles, Y2: 2 n doubles
int lz_fused_steps(hipStream_t st, const double* M, int n, int j0, int j1, int qcap, double* Q, double* Y2, double* PA2,
                   double* ab);
static constexpr int LZ_FUSED_LIMIT = 4096;
// the same as one resident launch per batch (option lz_resident, n <= 1024): Y3 / PA3 = three n- / ceil(n/16)-vectors, flag =
// two words (flag[1] != 0 afterwards: the launch gave up, nothing of the batch is valid); lz_resident_prepare before step 0
bool lz_resident_ok(const lrn_ctx* c, int n);
void lz_resident_prepare(hipStream_t st, int n, double* Y3, double* PA3, unsigned* flag);
int lz_resident_steps(hipStream_t st, const double* M, int n, int j0, int j1, int qcap, double* Q, double* Y3, double* PA3,
                      double* ab, unsigned* flag);
}  // namespace lrn
