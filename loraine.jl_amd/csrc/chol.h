// Blocked Cholesky / triangular-solve entry points (chol.hip).
#pragma once
#include "lrn_common.h"

namespace lrn {
static constexpr int CHOL_NB = 64;
inline size_t chol_linv_doubles(int n) { return (size_t)((n + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB; }
// A (n x n, ld, lower, col-major) -> L in place; Linv: inverse of each diagonal block;
// work: n*NB doubles; info_dev: device int (0 = ok, k>0 = not PD at column k).
int potrf_lower(hipStream_t st, double* A, int n, int ld, double* Linv, double* work, int* info_dev);
// same with pivot boosting against the original diagonal diag0[n] (see potrf_diag_wave_kernel); info_dev[1]
// must be zeroed by the caller and returns the number of boosted pivots
int potrf_lower_boost(hipStream_t st, double* A, int n, int ld, double* Linv, double* work, int* info_dev,
                      const double* diag0, double boost, int max_boost);
// x = L^-T L^-1 h ; r,y: scratch n doubles each
int potrs_vec(hipStream_t st, const double* L, int n, int ld, const double* Linv, const double* h,
              double* x, double* r, double* y);
// B <- L^-1 B or L^-T B (n x nrhs); tmp: NB*nrhs doubles
int trsm_left_lower(hipStream_t st, const double* L, int n, int ld, const double* Linv, bool trans,
                    double* B, int nrhs, int ldb, double* tmp);
}  // namespace lrn
