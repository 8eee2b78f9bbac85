// extern "C" entry points of libloraine_hip.so (see include/loraine_hip.h).
#include <cstring>

#include "../../include/loraine_hip.h"
#include "ctx.h"

using namespace lrn;

void lrn_free_model(lrn_ctx* c);
namespace lrn {
void prec_free(lrn_ctx* c);
}

extern "C" {

int lrn_version(void) { return 100; }

int lrn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int lrn_create(lrn_ctx** out, int device) {
  if (!out) return LRN_ERR_ARG;
  *out = nullptr;
  int n = lrn_device_count();
  if (n <= 0 || device < 0 || device >= n) return LRN_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return LRN_ERR_HIP;
  lrn_ctx* c = new lrn_ctx();
  c->device = device;
  c->profile = false;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return LRN_ERR_HIP;
  }
  *out = c;
  return LRN_OK;
}

int lrn_destroy(lrn_ctx* c) {
  if (!c) return LRN_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  prec_free(c);
  comm_free(c);
  lrn_free_model(c);
  release(c->info_dev);
  release(c->scratch);
  release(c->jscratch);
  release(c->redbuf);
  release(c->redout);
  release(c->lzbuf);
  release(c->lzbuf2);
  release(c->lxbuf);
  release(c->ezbuf);
  release(c->commvec);
  release(c->commmat);
  release(c->hopbuf);
  release(c->triw);
  release(c->cgpart);
  if (c->pin) (void)hipHostFree(c->pin);
  for (hipEvent_t e : c->pcg_ev) if (e) (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  if (c->evA) (void)hipEventDestroy(c->evA);
  if (c->evB) (void)hipEventDestroy(c->evB);
  (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  delete c;
  return LRN_OK;
}

const char* lrn_last_error(lrn_ctx* c) {
  if (!c) return "null context";
  if (const char* g = lrn::gemm_last_error()) {        // gemm() cannot reach the context: append its reason once
    if (c->err.find(g) == std::string::npos) c->err += std::string(c->err.empty() ? "" : " | ") + g;
  }
  return c->err.c_str();
}

int lrn_set_option(lrn_ctx* c, const char* key, double value) {
  if (!c || !key) return LRN_ERR_ARG;
  if (!strcmp(key, "dense_threshold")) c->opt.dense_threshold = value;
  else if (!strcmp(key, "profile")) c->profile = value != 0.0;
  else if (!strcmp(key, "t_batch")) { c->opt.t_batch = (long)value; for (auto& b : c->lmi) b.t_cap = b.p_cap = 0; }
  else if (!strcmp(key, "p_batch")) { c->opt.p_batch = (long)value; for (auto& b : c->lmi) b.t_cap = b.p_cap = 0; }
  else if (!strcmp(key, "prec_eig")) c->opt.prec_eig = (int)value;
  else if (!strcmp(key, "pivot_boost")) c->opt.pivot_boost = value;
  else if (!strcmp(key, "schur_chol")) c->opt.schur_chol = (int)value;
  else if (!strcmp(key, "schur_plan")) c->opt.schur_plan = (int)value;
  else if (!strcmp(key, "gemm3_ksplit")) c->opt.gemm3_ksplit = (int)value;
  else if (!strcmp(key, "gemm_no_skip")) c->opt.gemm_no_skip = (int)value;
  else if (!strcmp(key, "gemm_dyn_masks")) c->opt.gemm_dyn_masks = (int)value;
  else if (!strcmp(key, "gemm1_diag")) c->opt.gemm1_diag = (int)value;
  else if (!strcmp(key, "gemm_lab")) c->opt.gemm_lab = (int)value & 127;
  else if (!strcmp(key, "gemm3_sched")) c->opt.gemm3_sched = (int)value;
  else if (!strcmp(key, "gemm3_tile")) c->opt.gemm3_tile = (int)value;
  else if (!strcmp(key, "gemm3_strip")) c->opt.gemm3_strip = (int)value;
  else if (!strcmp(key, "gemm3_stagger")) c->opt.gemm3_stagger = (int)value;
  else if (!strcmp(key, "jacobi_wgs")) c->opt.jacobi_wgs = (int)value;
  else if (!strcmp(key, "jacobi_block")) c->opt.jacobi_block = (int)value;
  else if (!strcmp(key, "jacobi_inner")) c->opt.jacobi_inner = (int)value;
  else if (!strcmp(key, "pair_lanes")) c->opt.pair_lanes = (int)value;
  else if (!strcmp(key, "matvec_sparse")) c->opt.matvec_sparse = (int)value;
  else if (!strcmp(key, "prec_dense")) c->opt.prec_dense = (int)value;
  else if (!strcmp(key, "wmw_pattern_min")) c->opt.wmw_pattern_min = std::max(2, (int)value);
  else if (!strcmp(key, "profile_symv")) c->opt.profile_symv = (int)value;
  else if (!strcmp(key, "matvec_h")) { c->opt.matvec_h = (int)value; c->hop_version = -1; }
  else if (!strcmp(key, "comm_fail_ensure")) lrn::comm_inject_ensure_failure(c);      // test hook (tests/test_gpu_comm.py)
  else if (!strcmp(key, "pcg_lookahead")) c->opt.pcg_lookahead = std::max(0, std::min(8, (int)value));
  else if (!strcmp(key, "jacobi_cross")) c->opt.jacobi_cross = (int)value;
  else if (!strcmp(key, "jacobi_early")) c->opt.jacobi_early = value;
  else if (!strcmp(key, "eigmin_pair")) c->opt.eigmin_pair = (int)value;
  else if (!strcmp(key, "lz_resident")) c->opt.lz_resident = (int)value;
  else if (!strcmp(key, "prepw_streams")) c->opt.prepw_streams = (int)value;
  else if (!strcmp(key, "jacobi_warm")) c->opt.jacobi_warm = value != 0.0;
  else if (!strcmp(key, "nt_mode")) c->opt.nt_mode = (int)value;
  else if (!strcmp(key, "prec_inv")) c->opt.prec_inv = (int)value;
  else if (!strcmp(key, "shard_passes")) c->opt.shard_passes = (int)value;
  else if (!strcmp(key, "shard_products")) c->opt.shard_products = (int)value;
  else if (!strcmp(key, "shard_products_min")) c->opt.shard_products_min = std::max(16, (int)value);
  else if (!strcmp(key, "ns_l0")) c->opt.ns_l0 = value;
  else if (!strcmp(key, "ns_maxit")) c->opt.ns_maxit = (int)value;
  else if (!strcmp(key, "ns_lanczos")) c->opt.ns_lanczos = (int)value;
  else if (!strcmp(key, "ns_lanczos_min")) c->opt.ns_lanczos_min = std::max(8, (int)value);
  else if (!strcmp(key, "ns_dual")) c->opt.ns_dual = (int)value;
  else if (!strcmp(key, "lyap_tol")) c->opt.lyap_tol = value;
  else if (!strcmp(key, "lyap_maxit")) c->opt.lyap_maxit = (int)value;
  else if (!strcmp(key, "lyap_form")) c->opt.lyap_form = (int)value;
  else if (!strcmp(key, "shard_bs")) { if (value < 0) return LRN_ERR_ARG; c->shard_bs_opt = (int)value; lrn::update_shard_bs(c); }
  else if (!strcmp(key, "reset_timing")) { c->timing.clear(); c->counts.clear(); }
  else return set_error(c, LRN_ERR_ARG, "unknown option %s", key);
  return LRN_OK;
}

int lrn_set_shard(lrn_ctx* c, int rank, int world) {
  if (!c || world < 1 || rank < 0 || rank >= world) return LRN_ERR_ARG;
  // the Schur column sharding needs position space (nlmi == 1); the row-sharded CG mat-vec does not
  c->rank = rank;
  c->world = world;
  lrn::update_shard_bs(c);
  c->hop_version = -1;        // the operator choice and an H assembled under another sharding do not carry over
  c->H_version = -1;
  return LRN_OK;
}


int lrn_set_scaling(lrn_ctx* c, int il, const double* W, const double* G) {
  if (!c || il < 0 || il >= c->nlmi || !W) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LmiBlock& b = c->lmi[il];
  size_t mm = (size_t)b.msz * b.msz * 8;
  LRN_TRY(copy_in(c, b.W.p, W, mm));
  b.have_W = true;
  b.have_G = false;
  b.nt_free = false;
  c->scal_version += 1;
  if (G) {
    LRN_TRY(copy_in(c, b.G.p, G, mm));
    b.have_G = true;
  }
  return LRN_OK;
}

int lrn_set_lin(lrn_ctx* c, const double* X_lin, const double* S_lin_inv) {
  if (!c) return LRN_ERR_ARG;
  if (c->nlin == 0) return LRN_OK;
  if (!X_lin || !S_lin_inv) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  // tiny vector: form the product on the host side of the boundary
  std::vector<double> x(c->nlin), s(c->nlin);
  if (is_device_ptr(X_lin)) {
    LRN_HIP(c, hipMemcpy(x.data(), X_lin, (size_t)c->nlin * 8, hipMemcpyDeviceToHost));
    LRN_HIP(c, hipMemcpy(s.data(), S_lin_inv, (size_t)c->nlin * 8, hipMemcpyDeviceToHost));
  } else {
    memcpy(x.data(), X_lin, (size_t)c->nlin * 8);
    memcpy(s.data(), S_lin_inv, (size_t)c->nlin * 8);
  }
  for (int i = 0; i < c->nlin; ++i) x[i] *= s[i];
  c->scal_version += 1;
  return copy_in(c, c->lin_xs.p, x.data(), (size_t)c->nlin * 8);
}

int lrn_schur_assemble(lrn_ctx* c, int mode, double* H_out) {
  if (!c) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  if (c->comm && c->world > 1 && c->pos_space) {
    // multi-GPU (lrn_comm_init): agree on the assembly path once, assemble the owned share, then -- whatever the local
    // outcome -- enter the status reduction and the exchange on the library's stream
    LRN_TRY(comm_agree_plan(c, mode));
    const int rc = schur_assemble(c, mode);
    LRN_TRY(comm_schur_exchange(c, rc));
  } else {
    LRN_TRY(schur_assemble(c, mode));
  }
  if (H_out) return schur_get(c, H_out);
  return LRN_OK;
}

int lrn_schur_get(lrn_ctx* c, double* H_out) {
  if (!c || !H_out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  return schur_get(c, H_out);
}

int lrn_schur_add_diag(lrn_ctx* c, double eps) {
  if (!c) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  return schur_add_diag(c, eps);
}

int lrn_schur_factor(lrn_ctx* c, int* info) {
  if (!c) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  return schur_factor(c, info);
}

int lrn_schur_solve(lrn_ctx* c, const double* h, double* dely) {
  if (!c || !h || !dely) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  return schur_solve(c, h, dely);
}

// ---- multi-GPU exchange buffers: block-cyclic column blocks, rank-major
static int shard_geom(lrn_ctx* c, int* nblk, int* bpr) {
  *nblk = (c->nvar + c->shard_bs - 1) / c->shard_bs;
  *bpr = (*nblk + c->world - 1) / c->world;
  return LRN_OK;
}

int64_t lrn_schur_shard_doubles(lrn_ctx* c) {
  if (!c || c->nvar <= 0) return 0;
  int nblk, bpr;
  shard_geom(c, &nblk, &bpr);
  return (int64_t)bpr * c->shard_bs * c->nvar;
}

int lrn_schur_plan(lrn_ctx* c, int mode, int* plan) {
  if (!c || !plan) return LRN_ERR_ARG;
  if (c->nvar <= 0) return set_error(c, LRN_ERR_STATE, "no model uploaded");
  LRN_HIP(c, hipSetDevice(c->device));
  *plan = schur_plan(c, mode);
  return LRN_OK;
}

int lrn_schur_is_partial_sum(lrn_ctx* c) { return c && c->have_H && c->H_partial ? 1 : 0; }

int lrn_schur_export_full(lrn_ctx* c, double* buf) {
  if (!c || !buf) return LRN_ERR_ARG;
  if (!c->have_H) return set_error(c, LRN_ERR_STATE, "no assembled H");
  LRN_HIP(c, hipSetDevice(c->device));
  return copy_out(c, buf, c->H.p, (size_t)c->nvar * c->nvar * 8);
}

int lrn_schur_import_full(lrn_ctx* c, const double* buf) {
  if (!c || !buf) return LRN_ERR_ARG;
  if (c->nvar <= 0 || !c->H.p) return set_error(c, LRN_ERR_STATE, "no model / Schur matrix");
  LRN_HIP(c, hipSetDevice(c->device));
  LRN_TRY(copy_in(c, c->H.p, buf, (size_t)c->nvar * c->nvar * 8));
  c->have_H = true;
  c->H_partial = false;
  c->H_shifted = false;
  c->have_L = false;
  c->H_version = -1;          // (an imported matrix: the library does not know which scaling it belongs to)
  return LRN_OK;
}

int lrn_schur_export_shard(lrn_ctx* c, double* buf) {
  if (!c || !buf) return LRN_ERR_ARG;
  if (!c->have_H) return set_error(c, LRN_ERR_STATE, "no assembled H");
  if (c->H_partial)
    return set_error(c, LRN_ERR_STATE, "H holds a partial sum: exchange it with lrn_schur_export_full + all-reduce");
  if (!is_device_ptr(buf)) return set_error(c, LRN_ERR_ARG, "shard buffer must be device memory");
  LRN_HIP(c, hipSetDevice(c->device));
  int nblk, bpr;
  shard_geom(c, &nblk, &bpr);
  const long n = c->nvar, bs = c->shard_bs;
  for (int lb = 0; lb < bpr; ++lb) {
    int gb = shard_global_block(c->rank, lb, c->world);
    double* dst = buf + (long)lb * bs * n;
    if (gb >= nblk) {
      LRN_HIP(c, hipMemsetAsync(dst, 0, (size_t)bs * n * 8, c->stream));
      continue;
    }
    long c0 = (long)gb * bs, nc = std::min<long>(bs, n - c0);
    LRN_HIP(c, hipMemcpyAsync(dst, c->H.as<double>() + c0 * n, (size_t)nc * n * 8, hipMemcpyDeviceToDevice, c->stream));
    if (nc < bs) LRN_HIP(c, hipMemsetAsync(dst + nc * n, 0, (size_t)(bs - nc) * n * 8, c->stream));
  }
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  return LRN_OK;
}

int lrn_schur_import_all(lrn_ctx* c, const double* buf_all) {
  if (!c || !buf_all) return LRN_ERR_ARG;
  if (!is_device_ptr(buf_all)) return set_error(c, LRN_ERR_ARG, "gathered buffer must be device memory");
  LRN_HIP(c, hipSetDevice(c->device));
  int nblk, bpr;
  shard_geom(c, &nblk, &bpr);
  const long n = c->nvar, bs = c->shard_bs;
  const long per_rank = (long)bpr * bs * n;
  for (int r = 0; r < c->world; ++r)
    for (int lb = 0; lb < bpr; ++lb) {
      int gb = shard_global_block(r, lb, c->world);
      if (gb >= nblk) continue;
      long c0 = (long)gb * bs, nc = std::min<long>(bs, n - c0);
      LRN_HIP(c, hipMemcpyAsync(c->H.as<double>() + c0 * n, buf_all + (long)r * per_rank + (long)lb * bs * n,
                                (size_t)nc * n * 8, hipMemcpyDeviceToDevice, c->stream));
    }
  LRN_HIP(c, hipStreamSynchronize(c->stream));
  c->have_H = true;
  c->H_shifted = false;
  c->have_L = false;
  return LRN_OK;
}

// ---- measurement
// The reference's own profile vocabulary (TimerOutputs sections, src/makeBBBB.jl:2,30,86-98,140, src/Solvers.jl:583,676,
// src/prepare_W.jl:37-40, src/predictor_corrector.jl:234,243) as aliases of the library's keys, so a Loraine.jl
// maintainer can put the two profiles side by side.
static const struct { const char* ref; const char* keys[5]; } kTimingAlias[] = {
    {"BBBBone1", {"gemm1"}},                                       // mul!(tmp1, W, A_i)
    {"BBBBone2", {"gemm2"}},                                       // tmp = tmp1 * W
    {"BBBBone3", {"gemm3", "reduce3"}},                            // tmp2 = AA * vec(tmp)
    {"BBBBone4", {}},                                              // BBBB[indi,i] = -tmp2[indi]: the GEMM3 epilogue
    {"BBBBone", {"wchol", "gemm1", "gemm2", "gemm3", "reduce3"}},
    {"BBBBthree", {"sparse"}},
    {"BBBB_rank1", {"rank1"}},
    {"BBBBs", {"assemble"}},
    {"Ax", {"matvec"}},
    {"prec", {"prec_setup"}},
    {"prep W SVD", {"prepw_svd"}},
    {"prep W SVD svd", {"prepw_svd"}},
    {"CG predictor", {"pcg"}},
    {"CG corrector", {"pcg"}},
    {"find step corrector", {"find_step"}},
};

int lrn_get_timing(lrn_ctx* c, const char* key, double* ms) {
  if (!c || !key || !ms) return LRN_ERR_ARG;
  for (const auto& a : kTimingAlias)
    if (!strcmp(key, a.ref)) {
      double s = 0.0;
      for (const char* k : a.keys)
        if (k) { auto f = c->timing.find(k); if (f != c->timing.end()) s += f->second; }
      *ms = s;
      return LRN_OK;
    }
  auto it = c->timing.find(key);
  if (it == c->timing.end()) { *ms = 0.0; return LRN_ERR_ARG; }
  *ms = it->second;
  return LRN_OK;
}

int64_t lrn_get_count(lrn_ctx* c, const char* key) {
  if (!c || !key) return 0;
  if (!strcmp(key, "shard_bs")) return c->shard_bs;      // state, not a per-call counter (survives reset_timing)
  auto it = c->counts.find(key);
  return it == c->counts.end() ? 0 : it->second;
}

int lrn_mfma_f64_peak(lrn_ctx* c, double* tflops) {
  if (!c || !tflops) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  return mfma_f64_peak(c->stream, tflops);
}

__global__ __launch_bounds__(256) void xcc_probe_kernel(int* __restrict__ out, long hold_ticks) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (hold_ticks > 0) {
    long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) out[blockIdx.x + (long)gridDim.x * blockIdx.z] = (int)(xcc & 0xf);
}

int lrn_xcc_probe(lrn_ctx* c, int nx, int nz, int hold_us, int32_t* out) {
  if (!c || !out || nx <= 0 || nz <= 0 || nz > 65535 || hold_us < 0 || hold_us > 100000) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf d;
  LRN_TRY(ensure(c, d, (size_t)nx * nz * 4, true));
  hipLaunchKernelGGL(xcc_probe_kernel, dim3(nx, 1, nz), dim3(256), 0, c->stream, d.as<int>(), (long)hold_us * 100);
  int rc = copy_out(c, out, d.p, (size_t)nx * nz * 4);      // wall_clock64 ticks at 100 MHz
  release(d);
  return rc;
}

__global__ void copy16_kernel(const double2* __restrict__ src, double2* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

int lrn_hbm_copy_peak(lrn_ctx* c, int64_t bytes, double* gbps) {
  if (!c || !gbps || bytes < 4096) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf a, b;
  LRN_TRY(ensure(c, a, (size_t)bytes, true));
  LRN_TRY(ensure(c, b, (size_t)bytes, true));
  long n = bytes / 16;
  hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, c->stream, a.as<double2>(), b.as<double2>(), n);
  (void)hipEventRecord(c->ev0, c->stream);
  const int reps = 5;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, c->stream, a.as<double2>(), b.as<double2>(), n);
  (void)hipEventRecord(c->ev1, c->stream);
  (void)hipEventSynchronize(c->ev1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  *gbps = 2.0 * (double)n * 16.0 * reps / (ms * 1e-3) / 1e9;
  release(a);
  release(b);
  return LRN_OK;
}

// ---- debug / unit-test building blocks
int lrn_dbg_gemm(lrn_ctx* c, int transA, int transB, int M, int N, int K, double alpha, const double* A,
                 int lda, const double* B, int ldb, double beta, double* C, int ldc, int flags, int ksplit) {
  if (!c || !A || !B || !C) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  // host column-major operands: A is (transA ? K x M : M x K) with leading dim lda, etc.
  size_t a_el = (size_t)lda * (transA ? M : K), b_el = (size_t)ldb * (transB ? K : N), c_el = (size_t)ldc * N;
  DBuf dA, dB, dC, dS;
  LRN_TRY(ensure(c, dA, a_el * 8));
  LRN_TRY(ensure(c, dB, b_el * 8));
  LRN_TRY(ensure(c, dC, c_el * 8));
  LRN_TRY(copy_in(c, dA.p, A, a_el * 8));
  LRN_TRY(copy_in(c, dB.p, B, b_el * 8));
  LRN_TRY(copy_in(c, dC.p, C, c_el * 8));
  GemmDesc g;
  g.A = dA.as<double>(); g.B = dB.as<double>();
  g.M = M; g.N = N; g.K = K;
  if (!transA) { g.sAm = 1; g.sAk = lda; } else { g.sAm = lda; g.sAk = 1; }
  if (!transB) { g.sBk = 1; g.sBn = ldb; } else { g.sBk = ldb; g.sBn = 1; }
  g.alpha = alpha; g.flags = flags;
  int rc;
  if (ksplit > 1 || (flags & GEMM_KSEG_TRI)) {
    if (ksplit < 1) ksplit = 1;
    LRN_TRY(ensure(c, dS, (size_t)ksplit * M * N * 8, true));
    g.C = dS.as<double>(); g.sCm = 1; g.sCn = M; g.ksplit = ksplit; g.sCs = (long)M * N; g.beta = 0.0;
    if (flags & GEMM_KSEG_TRI) {
      // test convention: K = ld*ld with ld = isqrt(K)
      int ld = 1;
      while ((long)(ld + 1) * (ld + 1) <= K) ++ld;
      g.kseg_ld = ld; g.kseg_cols = ld;
    }
    rc = gemm(c->stream, g);
    if (rc == LRN_OK) {
      // compact reduce then C = beta*C + sum
      DBuf dR;
      LRN_TRY(ensure(c, dR, (size_t)M * N * 8, true));
      rc = reduce_slabs(c->stream, dS.as<double>(), (long)M * N, ksplit, dR.as<double>(), (long)M * N, 0.0);
      std::vector<double> r((size_t)M * N), cc(c_el);
      LRN_TRY(copy_out(c, r.data(), dR.p, (size_t)M * N * 8));
      LRN_TRY(copy_out(c, cc.data(), dC.p, c_el * 8));
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) cc[(size_t)i + (size_t)j * ldc] = beta * cc[(size_t)i + (size_t)j * ldc] + r[(size_t)i + (size_t)j * M];
      LRN_TRY(copy_in(c, dC.p, cc.data(), c_el * 8));
      release(dR);
    }
  } else {
    g.C = dC.as<double>(); g.sCm = 1; g.sCn = ldc; g.beta = beta;
    rc = gemm(c->stream, g);
    static const int reps = getenv("LRN_DBG_GEMM_REPS") ? atoi(getenv("LRN_DBG_GEMM_REPS")) : 0;   // (measurement: time the product)
    if (reps > 0 && rc == LRN_OK && beta == 0.0) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, c->stream);
      for (int i = 0; i < reps; ++i) (void)gemm(c->stream, g);
      (void)hipEventRecord(e1, c->stream); (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      fprintf(stderr, "[dbg_gemm] M %d N %d K %d flags %d: %.2f us per product (%d back to back)\n", M, N, K, flags, ms * 1e3 / reps, reps);
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
  }
  if (rc != LRN_OK) return set_error(c, rc, "gemm launch failed");
  rc = copy_out(c, C, dC.p, c_el * 8);
  release(dA); release(dB); release(dC); release(dS);
  return rc;
}

int lrn_dbg_mfma_probe(lrn_ctx* c, const double* A, const double* B, double* D) {
  if (!c || !A || !B || !D) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf dA, dB, dD;
  LRN_TRY(ensure(c, dA, 64 * 8));
  LRN_TRY(ensure(c, dB, 64 * 8));
  LRN_TRY(ensure(c, dD, 256 * 8, true));
  LRN_TRY(copy_in(c, dA.p, A, 64 * 8));
  LRN_TRY(copy_in(c, dB.p, B, 64 * 8));
  LRN_TRY(mfma_f64_probe(c->stream, dA.as<double>(), dB.as<double>(), dD.as<double>()));
  int rc = copy_out(c, D, dD.p, 256 * 8);
  release(dA); release(dB); release(dD);
  return rc;
}

static int dbg_factor(lrn_ctx* c, int n, const double* A, DBuf& dA, DBuf& dI, int* info) {
  DBuf dW;
  LRN_TRY(ensure(c, dA, (size_t)n * n * 8));
  LRN_TRY(ensure(c, dI, chol_linv_doubles(n) * 8));
  LRN_TRY(ensure(c, dW, (size_t)n * CHOL_NB * 8));
  LRN_TRY(ensure(c, c->info_dev, 64));
  LRN_TRY(copy_in(c, dA.p, A, (size_t)n * n * 8));
  LRN_HIP(c, hipMemsetAsync(c->info_dev.p, 0, 4, c->stream));
  tic(c);
  LRN_TRY(potrf_lower(c->stream, dA.as<double>(), n, n, dI.as<double>(), dW.as<double>(), c->info_dev.as<int>()));
  toc(c, "dbg_potrf");
  int h = 0;
  LRN_TRY(copy_out(c, &h, c->info_dev.p, 4));
  if (info) *info = h;
  release(dW);
  return LRN_OK;
}

int lrn_dbg_potrf(lrn_ctx* c, int n, double* A, int* info) {
  if (!c || !A || n <= 0) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf dA, dI;
  LRN_TRY(dbg_factor(c, n, A, dA, dI, info));
  int rc = copy_out(c, A, dA.p, (size_t)n * n * 8);
  release(dA); release(dI);
  return rc;
}

int lrn_dbg_potrs(lrn_ctx* c, int n, const double* A, const double* b, double* x, int* info) {
  if (!c || !A || !b || !x || n <= 0) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf dA, dI, v;
  int inf = 0;
  LRN_TRY(dbg_factor(c, n, A, dA, dI, &inf));
  if (info) *info = inf;
  if (inf != 0) { release(dA); release(dI); return LRN_OK; }
  LRN_TRY(ensure(c, v, (size_t)(4 * n + 64) * 8, true));
  double* vb = v.as<double>();
  LRN_TRY(copy_in(c, vb, b, (size_t)n * 8));
  LRN_TRY(potrs_vec(c->stream, dA.as<double>(), n, n, dI.as<double>(), vb, vb + n, vb + 2 * n, vb + 3 * n));
  int rc = copy_out(c, x, vb + n, (size_t)n * 8);
  release(dA); release(dI); release(v);
  return rc;
}

int lrn_dbg_trsm(lrn_ctx* c, int n, int nrhs, int trans, const double* A, double* B, int* info) {
  if (!c || !A || !B || n <= 0 || nrhs <= 0) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf dA, dI, dB, dT;
  int inf = 0;
  LRN_TRY(dbg_factor(c, n, A, dA, dI, &inf));
  if (info) *info = inf;
  if (inf != 0) { release(dA); release(dI); return LRN_OK; }
  LRN_TRY(ensure(c, dB, (size_t)n * nrhs * 8));
  LRN_TRY(ensure(c, dT, (size_t)CHOL_NB * nrhs * 8));
  LRN_TRY(copy_in(c, dB.p, B, (size_t)n * nrhs * 8));
  LRN_TRY(trsm_left_lower(c->stream, dA.as<double>(), n, n, dI.as<double>(), trans != 0, dB.as<double>(), nrhs, n,
                          dT.as<double>()));
  int rc = copy_out(c, B, dB.p, (size_t)n * nrhs * 8);
  release(dA); release(dI); release(dB); release(dT);
  return rc;
}

}  // extern "C"
