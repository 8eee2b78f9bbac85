// Blocked dense Cholesky (lower, column-major) + triangular solves for gfx950.
//
// Replaces  cholesky(Hermitian(BBBB,:L)) / L'\(L\h)  (reference src/predictor_corrector.jl:
// 39,57,90,199), cholesky(X), cholesky(S) (src/prepare_W.jl:7,33-34) and cholesky(S+I)
// (src/Solvers.jl:805).
//
//  * right-looking, NB = 64: the diagonal block is factored by ONE workgroup entirely in
//    LDS; the panel solve X L_kk' = A21 is a forward substitution with one thread per row
//    (L_kk in LDS, the row in registers) and the trailing update (A22 -= L21 L21', lower
//    tiles only) runs on the FP64 MFMA GEMM.
//  * a non-positive pivot is reported LAPACK-style through a device `info` word
//    (first failing 1-based column); later blocks then skip their work.
//  * triangular solves substitute through the 64x64 diagonal blocks (no explicit inverses:
//    the Schur matrix, S + I of H_alpha and X, S late in the solve have condition numbers
//    beyond 1e12, where inv(L_kk) costs the digits LAPACK's backward-stable solves keep);
//    the off-diagonal panels are streamed once per block (bandwidth-bound, coalesced).
//    The `Linv` arguments of the entry points are kept for the callers' workspaces but unused.
#include "lrn_common.h"
#include "chol.h"

namespace lrn {

static constexpr int NB = CHOL_NB;

// ------------------------------------------------------------------ diagonal block
// A (nb x nb, lower, ld) -> L in place (Linv: unused, kept for the signature).
// diag0 != NULL (Schur matrix only): pivots at or below the rounding level of their original diagonal
// entry, pivot <= boost * diag0[j] (zero and negative ones included), are replaced by a huge value --
// the row drops out of the factor and the solves return 0 for it (the usual pivot boosting of
// interior-point Cholesky codes).  info[1] counts them; more than `max_boost` is a failure.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// The factorisation of the block by ONE wavefront without LDS or barriers: lane i keeps row i of the block in
// registers, the pivot and the column entries l_kj travel by v_readlane (wave-uniform SGPRs feeding
// the FMAs).  64 columns x (63 - j) rank-one updates = 2016 FMA per lane; 44 us per block against 57 us
// for a 256-thread LDS version with 3 barriers per column.  Entries above the diagonal of a lane's row
// are scratch.
__global__ __launch_bounds__(64) void potrf_diag_wave_kernel(double* __restrict__ A, int ld, int nb, int col0,
                                                             int* __restrict__ info, const double* __restrict__ diag0,
                                                             double boost, int max_boost) {
  const int i = threadIdx.x;
  if (*info != 0) return;
  double a[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) a[j] = (i < nb && j < nb && i >= j) ? A[(long)i + (long)j * ld] : (i == j ? 1.0 : 0.0);
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (bad == 0) {
      double piv = readlane_f64(a[j], j);
      if (diag0 && j < nb) {
        const double d0 = diag0[col0 + j];
        // d0 > 0: a structurally empty row (a variable that occurs in no constraint) is not rounding noise;
        // it fails like in the reference, whose +1e-4 I loop and regularisation count then decide
        if (d0 > 0.0 && piv <= boost * d0 && piv == piv) {
          piv = 1e40 * fmax(fabs(d0), 1.0);
          int cnt = 0;
          if (i == 0) cnt = atomicAdd(info + 1, 1) + 1;
          cnt = __builtin_amdgcn_readfirstlane(cnt);
          if (cnt > max_boost) bad = col0 + j + 1;
        }
      }
      if (!(piv > 0.0)) bad = col0 + j + 1;        // also catches NaN; wave-uniform
      if (bad == 0) {
        const double rl = 1.0 / sqrt(piv);
        const double lij = a[j] * rl;
        a[j] = (i == j) ? sqrt(piv) : lij;
#pragma unroll
        for (int k = j + 1; k < NB; ++k) a[k] -= lij * readlane_f64(lij, k);
      }
    }
  }
  if (bad) {
    if (i == 0) atomicCAS(info, 0, bad);
    return;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (i < nb && j < nb && i >= j) A[(long)i + (long)j * ld] = a[j];
}

// Panel of the factorisation: rows of A21 (rem x NB, ld) solve  x L_kk' = a  by forward substitution,
// one thread per row with the row in registers and L_kk (NB x NB, lower, full block) in LDS.
// Writes the result back in place and into the contiguous work panel W (rem x NB, ld rem).
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ A21, int ld, int rem,
                                                          const double* __restrict__ Lkk, double* __restrict__ W,
                                                          const int* __restrict__ info) {
  __shared__ double l[NB][NB + 1];
  __shared__ double rinv[NB];
  if (*info != 0) return;
  const int t = threadIdx.x;
  for (int e = t; e < NB * NB; e += 256) {
    int i = e % NB, j = e / NB;
    l[i][j] = i >= j ? Lkk[(long)i + (long)j * ld] : 0.0;
  }
  __syncthreads();
  if (t < NB) rinv[t] = 1.0 / l[t][t];
  __syncthreads();
  const int row = blockIdx.x * 256 + t;
  if (row >= rem) return;
  // (measured on tru9, nvar = 3240: this left-looking LDS form 35 us per panel; right-looking 44 us;
  // L_kk through scalar loads instead of LDS 57 us)
  double x[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) x[j] = A21[(long)row + (long)j * ld];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double s = x[j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= x[k] * l[j][k];
    x[j] = s * rinv[j];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    A21[(long)row + (long)j * ld] = x[j];
    W[(long)row + (long)j * rem] = x[j];
  }
}

// Diagonal step of the matrix solves: X_b = L_kk^-1 B_b (trans = 0) or L_kk^-T B_b (trans = 1) for the
// block rows [k0, k0 + nb) of B (ldb), one thread per right-hand side; result in place and in
// tmp (NB x nrhs, ld NB).
__global__ __launch_bounds__(256) void trsm_diag_kernel(const double* __restrict__ Lkk, int ld, int nb, int trans,
                                                        double* __restrict__ Bb, int ldb, int nrhs,
                                                        double* __restrict__ tmp) {
  __shared__ double l[NB][NB + 1];
  __shared__ double rinv[NB];
  const int t = threadIdx.x;
  for (int e = t; e < NB * NB; e += 256) {
    int i = e % NB, j = e / NB;
    l[i][j] = (i < nb && j < nb && i >= j) ? Lkk[(long)i + (long)j * ld] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  if (t < NB) rinv[t] = 1.0 / l[t][t];
  __syncthreads();
  const int col = blockIdx.x * 256 + t;
  if (col >= nrhs) return;
  double* bc = Bb + (long)col * ldb;
  double x[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) x[j] = j < nb ? bc[j] : 0.0;
  if (!trans) {        // L x = b, right-looking
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      x[j] *= rinv[j];
#pragma unroll
      for (int k = j + 1; k < NB; ++k) x[k] -= l[k][j] * x[j];
    }
  } else {             // L' x = b
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
      x[j] *= rinv[j];
#pragma unroll
      for (int k = 0; k < j; ++k) x[k] -= l[j][k] * x[j];
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < nb) bc[j] = x[j];
    tmp[(long)j + (long)col * NB] = j < nb ? x[j] : 0.0;
  }
}

// Vector version for one block, executed by wave 0 of the calling workgroup: lb[NB][NB+1] holds L_kk
// (identity-padded), v[] the block of the right-hand side in LDS; lane i owns component i.
__device__ __forceinline__ void block_subst_wave(double (*lb)[NB + 1], double* v, int nb, bool trans) {
  const int lane = threadIdx.x & 63;
  double r = lane < nb ? v[lane] : 0.0;
  if (!trans) {
    for (int c = 0; c < nb; ++c) {
      double xc = __shfl(r, c, 64) / lb[c][c];
      if (lane == c) r = xc;
      else if (lane > c) r -= lb[lane][c] * xc;
    }
  } else {
    for (int c = nb - 1; c >= 0; --c) {
      double xc = __shfl(r, c, 64) / lb[c][c];
      if (lane == c) r = xc;
      else if (lane < c) r -= lb[c][lane] * xc;
    }
  }
  if (lane < nb) v[lane] = r;
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
  return potrf_lower_boost(st, A, n, ld, Linv, work, info_dev, nullptr, 0.0, 0);
}

int potrf_lower_boost(hipStream_t st, double* A, int n, int ld, double* Linv, double* work, int* info_dev,
                      const double* diag0, double boost, int max_boost) {
  // work: n x NB doubles
  int nblk = (n + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    int k0 = b * NB;
    int nb = n - k0 < NB ? n - k0 : NB;
    double* Akk = A + (long)k0 + (long)k0 * ld;
    hipLaunchKernelGGL(potrf_diag_wave_kernel, dim3(1), dim3(64), 0, st, Akk, ld, nb, k0, info_dev, diag0, boost,
                       max_boost);
    int rem = n - k0 - nb;
    if (rem <= 0) break;
    // panel: Wk = A21 * Lkk^-T      (rem x nb), by substitution
    hipLaunchKernelGGL(potrf_panel_kernel, dim3((rem + 255) / 256), dim3(256), 0, st,
                       A + (long)(k0 + nb) + (long)k0 * ld, ld, rem, Akk, work, info_dev);
    int rc;
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
  (void)Linv;
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// ------------------------------------------------------------------ triangular solves (vector)
// stage the diagonal block L_kk (identity-padded) into LDS
__device__ __forceinline__ void load_diag_block(const double* __restrict__ L, int ld, int k0, int nb,
                                                double (*lb)[NB + 1], int nthreads) {
  for (int e = threadIdx.x; e < NB * NB; e += nthreads) {
    int i = e % NB, j = e / NB;
    lb[i][j] = (i < nb && j < nb && i >= j) ? L[(long)(k0 + i) + (long)(k0 + j) * ld] : (i == j ? 1.0 : 0.0);
  }
}

// forward step for block b:  y_b = L_kk^-1 r_b ; r[i] -= L[i, b] * y_b for i > block b
__global__ __launch_bounds__(256) void trsv_fwd_step(const double* __restrict__ L, int ld, int n,
                                                     const double* __restrict__ Linv, int k0, int nb,
                                                     double* __restrict__ r, double* __restrict__ y) {
  __shared__ double lb[NB][NB + 1];
  __shared__ double yb[NB];
  const int t = threadIdx.x;
  load_diag_block(L, ld, k0, nb, lb, 256);
  if (t < NB) yb[t] = t < nb ? r[k0 + t] : 0.0;
  __syncthreads();
  if (t < 64) block_subst_wave(lb, yb, nb, false);
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

// backward step for block b:  x_b = L_kk^-T r_b ; r[c] -= sum_i L[k0+i, c] x_b[i], c < k0
__global__ __launch_bounds__(256) void trsv_bwd_step(const double* __restrict__ L, int ld, int n,
                                                     const double* __restrict__ Linv, int k0, int nb,
                                                     double* __restrict__ r, double* __restrict__ x) {
  __shared__ double lb[NB][NB + 1];
  __shared__ double xb[NB];
  const int t = threadIdx.x;
  load_diag_block(L, ld, k0, nb, lb, 256);
  if (t < NB) xb[t] = t < nb ? r[k0 + t] : 0.0;
  __syncthreads();
  if (t < 64) block_subst_wave(lb, xb, nb, true);
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
  __shared__ double lb[NB][NB + 1];
  __shared__ double yb[NB];
  const int t = threadIdx.x;
  for (int i = t; i < n; i += 1024) r[i] = h[i];
  const int nblk = (n + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {               // forward
    const int k0 = b * NB, nbk = n - k0 < NB ? n - k0 : NB;
    load_diag_block(L, ld, k0, nbk, lb, 1024);
    __syncthreads();
    if (t < 64) {
      block_subst_wave(lb, r + k0, nbk, false);
      if (t < nbk) yb[t] = r[k0 + t];
    }
    __syncthreads();
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
    load_diag_block(L, ld, k0, nbk, lb, 1024);
    __syncthreads();
    if (t < 64) {
      block_subst_wave(lb, r + k0, nbk, true);
      if (t < nbk) yb[t] = r[k0 + t];
    }
    __syncthreads();
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
  (void)Linv;
  int nblk = (n + NB - 1) / NB;
  for (int bi = 0; bi < nblk; ++bi) {
    int b = trans ? nblk - 1 - bi : bi;
    int k0 = b * NB, nb = n - k0 < NB ? n - k0 : NB;
    // X_b = op(L_kk)^-1 B_b by substitution, one thread per right-hand side; tmp = X_b (NB x nrhs)
    hipLaunchKernelGGL(trsm_diag_kernel, dim3((nrhs + 255) / 256), dim3(256), 0, st,
                       L + (long)k0 + (long)k0 * ld, ld, nb, trans ? 1 : 0, B + k0, ldb, nrhs, tmp);
    int rc;
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
