// Entry points not implemented yet in this build step.
#include "../../include/loraine_hip.h"
#include "ctx.h"
using namespace lrn;
namespace lrn { void prec_free(lrn_ctx*) {} }
extern "C" {
int lrn_prepare_w(lrn_ctx* c, int, const double*, const double*, double*, double*, double*, double*, double*, double*, int*) { return set_error(c, LRN_ERR_STATE, "lrn_prepare_w: not built"); }
int lrn_make_rhs(lrn_ctx* c, const double*, const double* const*, double*) { return set_error(c, LRN_ERR_STATE, "lrn_make_rhs: not built"); }
int lrn_matvec(lrn_ctx* c, const double*, double*) { return set_error(c, LRN_ERR_STATE, "lrn_matvec: not built"); }
int lrn_prec_setup(lrn_ctx* c, int, int, int, int*) { return set_error(c, LRN_ERR_STATE, "lrn_prec_setup: not built"); }
int lrn_prec_apply(lrn_ctx* c, const double*, double*) { return set_error(c, LRN_ERR_STATE, "lrn_prec_apply: not built"); }
int lrn_pcg(lrn_ctx* c, const double*, double, int, double*, int*, int*) { return set_error(c, LRN_ERR_STATE, "lrn_pcg: not built"); }
int lrn_dbg_svd_jacobi(lrn_ctx* c, int, const double*, double*, double*, double*, int*) { return set_error(c, LRN_ERR_STATE, "lrn_dbg_svd_jacobi: not built"); }
}
