// Blocked dense Cholesky (lower, column-major) + triangular solves for gfx950.
//
// Replaces  cholesky(Hermitian(BBBB,:L)) / L'\(L\h)  (reference src/predictor_corrector.jl:
// 39,57,90,199), cholesky(X), cholesky(S) (src/prepare_W.jl:7,33-34) and cholesky(S+I)
// (src/Solvers.jl:805).
//
//  * right-looking, NB = 64: the diagonal block is factored by ONE workgroup entirely in
//    LDS, which also forms inv(L_kk); the panel solve (A21 * inv(L_kk)^T) and the
//    trailing update (A22 -= L21 L21^T, lower tiles only) run on the FP64 MFMA GEMM.
//  * a non-positive pivot is reported LAPACK-style through a device `info` word
//    (first failing 1-based column); later blocks then skip their work.
//  * triangular solves use the stored inv(L_kk) blocks: per block one launch that applies
//    the 64x64 inverse and streams the panel once (bandwidth-bound, coalesced).
#include "lrn_common.h"
#include "chol.h"

namespace lrn {

static constexpr int NB = CHOL_NB;

// ------------------------------------------------------------------ diagonal block
// A (nb x nb, lower, ld) -> L in place; inv(L) -> Linv (NB x NB, ld NB, zero upper).
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ A, int ld, int nb,
                                                         double* __restrict__ Linv, int col0,
                                                         int* __restrict__ info) {
  __shared__ double a[NB][NB + 1];
  __shared__ double x[NB][NB + 1];
  __shared__ double part[4][NB];
  __shared__ int bad;
  const int t = threadIdx.x;
  const int ti = t & 63, tg = t >> 6;            // row, column group (4 groups)
  if (t == 0) bad = 0;
  if (*info != 0) return;                        // an earlier block already failed
  for (int e = t; e < NB * NB; e += 256) {
    int i = e % NB, j = e / NB;
    a[i][j] = (i < nb && j < nb && i >= j) ? A[(long)i + (long)j * ld] : (i == j ? 1.0 : 0.0);
    x[i][j] = 0.0;
  }
  __syncthreads();
  // right-looking Cholesky: thread (ti, tg) owns row ti of the columns k == tg (mod 4)
  for (int j = 0; j < NB; ++j) {
    double piv = a[j][j];
    if (!(piv > 0.0)) {                          // also catches NaN; uniform across the workgroup
      if (t == 0) bad = col0 + j + 1;
      break;
    }
    double rl = 1.0 / sqrt(piv);
    double lij = a[ti][j] * rl;                  // scaled column entry of my row (valid for ti >= j)
    __syncthreads();
    if (tg == 0 && ti >= j) a[ti][j] = (ti == j) ? sqrt(piv) : lij;
    __syncthreads();
    for (int k = j + 1 + ((tg - (j + 1)) & 3); k <= ti; k += 4) a[ti][k] -= lij * a[k][j];
    __syncthreads();
  }
  __syncthreads();
  if (bad) {
    if (t == 0) atomicCAS(info, 0, bad);
    return;
  }
  // inverse: L X = I, column c by the 4 threads {c, c+64, c+128, c+192}: partial sums over
  // k == tg (mod 4), combined through LDS
  for (int i = 0; i < NB; ++i) {
    const int c = ti;
    double s = 0.0;
    if (c < i)
      for (int k = c + tg; k < i; k += 4) s += a[i][k] * x[k][c];
    part[tg][c] = s;
    __syncthreads();
    if (tg == 0) {
      if (c == i) x[i][c] = 1.0 / a[i][i];
      else if (c < i) x[i][c] = -(part[0][c] + part[1][c] + part[2][c] + part[3][c]) / a[i][i];
    }
    __syncthreads();
  }
  for (int e = t; e < NB * NB; e += 256) {
    int i = e % NB, j = e / NB;
    if (i < nb && j < nb && i >= j) A[(long)i + (long)j * ld] = a[i][j];
    Linv[e] = (i < nb && j < nb) ? x[i][j] : 0.0;
  }
}

__global__ void copy_panel_kernel(const double* __restrict__ src, int lds_, double* __restrict__ dst,
                                  int ldd, int rows, int cols, const int* __restrict__ info) {
  if (info && *info != 0) return;
  long n = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % rows), j = (int)(e / rows);
    dst[(long)i + (long)j * ldd] = src[(long)i + (long)j * lds_];
  }
}

int potrf_lower(hipStream_t st, double* A, int n, int ld, double* Linv, double* work, int* info_dev) {
  // work: n x NB doubles
  int nblk = (n + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    int k0 = b * NB;
    int nb = n - k0 < NB ? n - k0 : NB;
    double* Akk = A + (long)k0 + (long)k0 * ld;
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, st, Akk, ld, nb,
                       Linv + (long)b * NB * NB, k0, info_dev);
    int rem = n - k0 - nb;
    if (rem <= 0) break;
    // panel: Wk = A21 * inv(Lkk)^T      (rem x nb)
    GemmDesc g;
    g.A = A + (long)(k0 + nb) + (long)k0 * ld; g.sAm = 1; g.sAk = ld;
    g.B = Linv + (long)b * NB * NB;            g.sBk = NB; g.sBn = 1;   // B[k][n] = Linv[n][k]
    g.C = work; g.sCm = 1; g.sCn = rem;
    g.M = rem; g.N = nb; g.K = nb;
    int rc = gemm(st, g);
    if (rc) return rc;
    unsigned blocks = (unsigned)(((long)rem * nb + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy_panel_kernel, dim3(blocks), dim3(256), 0, st, work, rem,
                       A + (long)(k0 + nb) + (long)k0 * ld, ld, rem, nb, info_dev);
    // trailing: A22 -= Wk Wk^T (lower tiles)
    GemmDesc u;
    u.A = work; u.sAm = 1; u.sAk = rem;
    u.B = work; u.sBk = rem; u.sBn = 1;
    u.C = A + (long)(k0 + nb) + (long)(k0 + nb) * ld; u.sCm = 1; u.sCn = ld;
    u.M = rem; u.N = rem; u.K = nb;
    u.alpha = -1.0; u.beta = 1.0; u.flags = GEMM_TRI_LOWER;
    rc = gemm(st, u);
    if (rc) return rc;
  }
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// ------------------------------------------------------------------ triangular solves (vector)
// forward step for block b:  y_b = Linv_b * r_b ; r[i] -= L[i, b] * y_b for i > block b
__global__ __launch_bounds__(256) void trsv_fwd_step(const double* __restrict__ L, int ld, int n,
                                                     const double* __restrict__ Linv, int k0, int nb,
                                                     double* __restrict__ r, double* __restrict__ y) {
  __shared__ double yb[NB];
  const int t = threadIdx.x;
  if (t < NB) {
    double s = 0.0;
    if (t < nb)
      for (int c = 0; c <= t; ++c) s += Linv[t + c * NB] * r[k0 + c];
    yb[t] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < nb) y[k0 + t] = yb[t];
  int row = k0 + nb + blockIdx.x * 256 + t;
  if (row < n) {
    double s = 0.0;
    const double* Lr = L + row + (long)k0 * ld;
#pragma unroll 8
    for (int c = 0; c < nb; ++c) s += Lr[(long)c * ld] * yb[c];
    r[row] -= s;
  }
}

// backward step for block b:  x_b = Linv_b^T * r_b ; r[c] -= sum_i L[k0+i, c] x_b[i], c < k0
__global__ __launch_bounds__(256) void trsv_bwd_step(const double* __restrict__ L, int ld, int n,
                                                     const double* __restrict__ Linv, int k0, int nb,
                                                     double* __restrict__ r, double* __restrict__ x) {
  __shared__ double xb[NB];
  const int t = threadIdx.x;
  if (t < NB) {
    double s = 0.0;
    if (t < nb)
      for (int i = t; i < nb; ++i) s += Linv[i + t * NB] * r[k0 + i];
    xb[t] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < nb) x[k0 + t] = xb[t];
  int c = blockIdx.x * 256 + t;
  if (c < k0) {
    const double* Lc = L + k0 + (long)c * ld;
    double s = 0.0;
#pragma unroll 8
    for (int i = 0; i < nb; ++i) s += Lc[i] * xb[i];
    r[c] -= s;
  }
}

// whole L^-T L^-1 h in ONE workgroup (n <= POTRS_SMALL): 2*nblk dependent block steps without
// launch gaps; the right-hand side lives in LDS.
static constexpr int POTRS_SMALL = 2048;
__global__ __launch_bounds__(1024) void potrs_small_kernel(const double* __restrict__ L, int ld, int n,
                                                           const double* __restrict__ Linv,
                                                           const double* __restrict__ h, double* __restrict__ x) {
  __shared__ double r[POTRS_SMALL];
  __shared__ double yb[NB];
  const int t = threadIdx.x;
  for (int i = t; i < n; i += 1024) r[i] = h[i];
  __syncthreads();
  const int nblk = (n + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {               // forward
    const int k0 = b * NB, nbk = n - k0 < NB ? n - k0 : NB;
    const double* Li = Linv + (long)b * NB * NB;
    if (t < NB) {
      double s = 0.0;
      if (t < nbk)
        for (int c = 0; c <= t; ++c) s += Li[t + c * NB] * r[k0 + c];
      yb[t] = s;
    }
    __syncthreads();
    if (t < nbk) r[k0 + t] = yb[t];
    for (int row = k0 + nbk + t; row < n; row += 1024) {
      const double* Lr = L + row + (long)k0 * ld;
      double s = 0.0;
#pragma unroll 8
      for (int c = 0; c < nbk; ++c) s += Lr[(long)c * ld] * yb[c];
      r[row] -= s;
    }
    __syncthreads();
  }
  for (int b = nblk - 1; b >= 0; --b) {          // backward
    const int k0 = b * NB, nbk = n - k0 < NB ? n - k0 : NB;
    const double* Li = Linv + (long)b * NB * NB;
    if (t < NB) {
      double s = 0.0;
      if (t < nbk)
        for (int i = t; i < nbk; ++i) s += Li[i + t * NB] * r[k0 + i];
      yb[t] = s;
    }
    __syncthreads();
    if (t < nbk) r[k0 + t] = yb[t];
    for (int c = t; c < k0; c += 1024) {
      const double* Lc = L + k0 + (long)c * ld;
      double s = 0.0;
#pragma unroll 8
      for (int i = 0; i < nbk; ++i) s += Lc[i] * yb[i];
      r[c] -= s;
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += 1024) x[i] = r[i];
}

// x = L^{-T} L^{-1} h ; r is scratch (n doubles); x may alias h? no: h is read-only.
int potrs_vec(hipStream_t st, const double* L, int n, int ld, const double* Linv, const double* h,
              double* x, double* r, double* y) {
  if (n <= POTRS_SMALL) {
    hipLaunchKernelGGL(potrs_small_kernel, dim3(1), dim3(1024), 0, st, L, ld, n, Linv, h, x);
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  hipMemcpyAsync(r, h, (size_t)n * 8, hipMemcpyDeviceToDevice, st);
  int nblk = (n + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    int k0 = b * NB, nb = n - k0 < NB ? n - k0 : NB;
    int rem = n - k0 - nb;
    unsigned blocks = rem > 0 ? (unsigned)((rem + 255) / 256) : 1u;
    hipLaunchKernelGGL(trsv_fwd_step, dim3(blocks), dim3(256), 0, st, L, ld, n,
                       Linv + (long)b * NB * NB, k0, nb, r, y);
  }
  for (int b = nblk - 1; b >= 0; --b) {
    int k0 = b * NB, nb = n - k0 < NB ? n - k0 : NB;
    unsigned blocks = k0 > 0 ? (unsigned)((k0 + 255) / 256) : 1u;
    hipLaunchKernelGGL(trsv_bwd_step, dim3(blocks), dim3(256), 0, st, L, ld, n,
                       Linv + (long)b * NB * NB, k0, nb, y, x);
  }
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// ------------------------------------------------------------------ triangular solves (matrix)
// X = L^{-1} B  (trans=false)  or  X = L^{-T} B (trans=true); B (n x nrhs, ldb) overwritten.
// tmp: NB x nrhs doubles.
int trsm_left_lower(hipStream_t st, const double* L, int n, int ld, const double* Linv, bool trans,
                    double* B, int nrhs, int ldb, double* tmp) {
  int nblk = (n + NB - 1) / NB;
  for (int bi = 0; bi < nblk; ++bi) {
    int b = trans ? nblk - 1 - bi : bi;
    int k0 = b * NB, nb = n - k0 < NB ? n - k0 : NB;
    const double* Li = Linv + (long)b * NB * NB;
    // tmp = op(Linv_b) * B_b
    GemmDesc g;
    g.A = Li;
    if (!trans) { g.sAm = 1; g.sAk = NB; } else { g.sAm = NB; g.sAk = 1; }
    g.B = B + k0; g.sBk = 1; g.sBn = ldb;
    g.C = tmp; g.sCm = 1; g.sCn = NB;
    g.M = nb; g.N = nrhs; g.K = nb;
    int rc = gemm(st, g);
    if (rc) return rc;
    unsigned blocks = (unsigned)(((long)nb * nrhs + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy_panel_kernel, dim3(blocks), dim3(256), 0, st, tmp, NB, B + k0, ldb, nb,
                       nrhs, (const int*)nullptr);
    GemmDesc u;
    u.B = tmp; u.sBk = 1; u.sBn = NB;
    u.N = nrhs; u.K = nb; u.alpha = -1.0; u.beta = 1.0;
    if (!trans) {
      int rem = n - k0 - nb;
      if (rem <= 0) continue;
      u.A = L + (k0 + nb) + (long)k0 * ld; u.sAm = 1; u.sAk = ld;
      u.C = B + (k0 + nb); u.sCm = 1; u.sCn = ldb;
      u.M = rem;
    } else {
      if (k0 <= 0) continue;
      u.A = L + k0; u.sAm = ld; u.sAk = 1;          // A[m][k] = L[k0+k][m]
      u.C = B; u.sCm = 1; u.sCn = ldb;
      u.M = k0;
    }
    rc = gemm(st, u);
    if (rc) return rc;
  }
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

}  // namespace lrn
