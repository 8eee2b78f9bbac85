// Shared device-side operations (cgops.hip / prepw.hip) used by the resident IP step.
#pragma once
#include "ctx.h"
namespace lrn {
int ensure_m(lrn_ctx* c, int m);                                            // c->m0..m2 >= msz^2
int wmw(lrn_ctx* c, LmiBlock& b, double* M, double* P, double* Z);          // Z = W M W (M symmetric)
int aa_times(lrn_ctx* c, LmiBlock& b, const double* Z, double* y);          // y += AA vec(Z)
int aat_to_mat(lrn_ctx* c, LmiBlock& b, const double* x, double* M);        // M = mat(AA' x)
int prepare_w_block(lrn_ctx* c, LmiBlock& b, int* info);                    // NT scaling from b.X, b.S
// k largest eigenpairs (ascending), smallest eigenvalue and trace of a dense symmetric matrix
int lanczos_extremes(lrn_ctx* c, const double* M, int n, int k, double* lam_top, double* U_top, double* lam_min,
                     double* trace, int* steps_out);
}  // namespace lrn
