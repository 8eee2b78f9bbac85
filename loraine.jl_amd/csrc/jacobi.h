// One-sided Jacobi SVD entry point (jacobi.hip).
#pragma once
#include "ctx.h"
namespace lrn {
static constexpr int JAC_SMALL = 96;
// A (n x n, col-major) is overwritten by U*Sigma; V (n x n) receives the right singular
// vectors (may be null); sigma[n] the singular values (unsorted).
// v_init: V already holds an orthogonal matrix V0 and A = A0*V0 (warm start); V accumulates from V0.
int jacobi_svd(lrn_ctx* c, double* A, double* V, double* sigma, int n, int* sweeps_out, bool v_init = false);
}  // namespace lrn
