// Spectral divide-and-conquer eigensolver for the symmetric positive definite matrix
// K = CC' CC = L_X' S L_X of the NT scaling at large msz (prepare_W, reference src/prepare_W.jl:39-60).
//
// The blocked one-sided Jacobi needs 11-19 sweeps at msz = 1e4, each streaming the matrix once per
// round.  Near the central path K is well conditioned (cond(K) = cond(XS), typically < 1e3), so a
// GEMM-rich method applies: split the spectrum at a shift sigma with the matrix sign function
//     P = (sign(K - sigma I) + I) / 2        (projector on the eigenvalues above sigma)
// computed by the QDWH iteration (Nakatsukasa, Bai, Gygi 2010; Cholesky variant: one SYRK-like
// product, one Cholesky, two triangular solves per step, cubic convergence), take orthonormal bases
// Q1, Q2 of range(P), range(I - P) by CholeskyQR2 of P R, (I - P) R for random R, and recurse on
// Q1' K Q1 and Q2' K Q2; leaves go to the Jacobi kernel.  The result only has to be a good STARTING
// basis: the caller finishes with one-sided Jacobi sweeps on the original CC, whose convergence test
// is what guarantees the accuracy of (D, V) -- any failure in here (unbalanced split, Cholesky
// breakdown) falls back to Jacobi on the block in question.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ctx.h"
#include "jacobi.h"
#include "ops.h"

namespace lrn {


namespace {

struct Bump {
  double* base;
  size_t cap, off;
  double* take(size_t n) {
    n = (n + 31) & ~size_t(31);
    if (off + n > cap) return nullptr;
    double* p = base + off;
    off += n;
    return p;
  }
};

inline unsigned nbl(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

__global__ void k_shift_scale(const double* __restrict__ K, int n, double sigma, double inv_alpha, double* __restrict__ X) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    X[e] = (K[e] - (i == j ? sigma : 0.0)) * inv_alpha;
  }
}

// Z = I + c * M, lower triangle of M authoritative (mirrored)
__global__ void k_eye_plus_sym(const double* __restrict__ M, int n, double c, double* __restrict__ Z) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    double v = i >= j ? M[e] : M[(long)j + (long)i * n];
    Z[e] = c * v + (i == j ? 1.0 : 0.0);
  }
}

// out = sym(a X + b Y)
__global__ void k_axpby_sym(const double* __restrict__ X, const double* __restrict__ Y, int n, double a, double b,
                            double* __restrict__ out) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    long et = (long)j + (long)i * n;
    out[e] = 0.5 * ((a * X[e] + b * Y[e]) + (a * X[et] + b * Y[et]));
  }
}

__global__ void k_proj(double* __restrict__ X, int n) {          // X = (X + I) / 2
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    X[e] = 0.5 * (X[e] + (i == j ? 1.0 : 0.0));
  }
}

__global__ void k_rand(double* __restrict__ G, long total, unsigned seed) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)e * 2654435761u + seed * 40503u + 977u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    unsigned g = (unsigned)(e >> 32) * 2246822519u + h;
    g ^= g >> 15; g *= 0x2c1b3c6du; g ^= g >> 12;
    G[e] = ((double)(h ^ g) / 4294967296.0) - 0.5;
  }
}

__global__ void k_copy(const double* __restrict__ A, long total, double* __restrict__ B) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) B[e] = A[e];
}

__global__ void k_mirror_lower(double* __restrict__ A, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    if (i < j) A[e] = A[(long)j + (long)i * n];
  }
}

// d[j] = K[j,j] ; s[j] = sum_i |K[i,j]|
__global__ __launch_bounds__(256) void k_diag_colabs(const double* __restrict__ K, int n, double* __restrict__ d,
                                                     double* __restrict__ s) {
  __shared__ double sh[4];
  const double* col = K + (long)blockIdx.x * n;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += fabs(col[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    s[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    d[blockIdx.x] = col[blockIdx.x];
  }
}

__global__ __launch_bounds__(256) void k_trace(const double* __restrict__ M, int n, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += M[(long)i * n + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = sh[0] + sh[1] + sh[2] + sh[3];
}

// Qt2[a + i*k2] = R[i + (k + a)*n] - Qt2[a + i*k2]     (rows of (I - P) applied to the random block R2)
__global__ void k_rt_minus(const double* __restrict__ R, int n, int k, int k2, double* __restrict__ Qt2) {
  long total = (long)k2 * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int a = (int)(e % k2), i = (int)(e / k2);
    Qt2[e] = R[(long)i + (long)(k + a) * n] - Qt2[e];
  }
}

// y = (K - sigma I) x, one workgroup per 64 rows (K symmetric: column access is coalesced)
__global__ __launch_bounds__(256) void k_symv_shift(const double* __restrict__ K, int n, double sigma,
                                                    const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (i < n)
    for (int j = w; j < n; j += 4) s += K[(long)i + (long)j * n] * x[j];
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) y[i] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] - sigma * x[i];
}

// x <- y / ||y||, nrm[0] = ||y||     (single workgroup)
__global__ __launch_bounds__(1024) void k_normalize(const double* __restrict__ y, int n, double* __restrict__ x,
                                                    double* __restrict__ nrm) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += y[i] * y[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < 16; ++i) t += sh[i];
  t = sqrt(t);
  const double r = t > 0.0 ? 1.0 / t : 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) x[i] = y[i] * r;
  if (threadIdx.x == 0) nrm[0] = t;
}

int GM(hipStream_t st, int M, int N, int K, const double* A, long sAm, long sAk, const double* B, long sBk, long sBn,
       double* C, long sCm, long sCn, double alpha = 1.0, double beta = 0.0, int flags = 0) {
  GemmDesc g;
  g.A = A; g.sAm = sAm; g.sAk = sAk;
  g.B = B; g.sBk = sBk; g.sBn = sBn;
  g.C = C; g.sCm = sCm; g.sCn = sCn;
  g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.beta = beta; g.flags = flags;
  return gemm(st, g);
}

struct Work {            // Cholesky / TRSM scratch shared by all levels (sized for the top level)
  double* chw;           // n * CHOL_NB
  double* trw;           // CHOL_NB * n
  double* vec;           // 2 n
  int* info;
};

int chol_ok(lrn_ctx* c, double* Z, int n, const Work& w, bool* ok) {
  hipStream_t st = c->stream;
  LRN_HIP(c, hipMemsetAsync(w.info, 0, 4, st));
  LRN_TRY(potrf_lower(st, Z, n, n, nullptr, w.chw, w.info));
  int h = 0;
  LRN_TRY(copy_out(c, &h, w.info, 4));
  *ok = h == 0;
  return LRN_OK;
}

// X (n x n symmetric, ||X||_2 <= 1) -> sign(X) in place; Z, Y: n^2 scratch each
int qdwh_sign(lrn_ctx* c, double* X, int n, double* Z, double* Y, const Work& w, int* its_out, bool* ok,
              double l_start, int max_it) {
  hipStream_t st = c->stream;
  const long nn = (long)n * n;
  double l = l_start;          // l = 1: Halley steps  X (3I + X^2)(I + 3X^2)^-1
  int it = 0;
  *ok = true;
  while (it < max_it) {
    const double l2 = l * l;
    double a = 3.0;
    if (l < 1.0 - 1e-12) {
      const double dd = std::cbrt(4.0 * (1.0 - l2) / (l2 * l2));
      const double sq = std::sqrt(1.0 + dd);
      a = sq + 0.5 * std::sqrt(8.0 - 4.0 * dd + 8.0 * (2.0 - l2) / (l2 * sq));
    }
    const double b = (a - 1.0) * (a - 1.0) / 4.0;
    const double cc = a + b - 1.0;
    // Z = I + c X^2   (X symmetric: lower tiles of X*X, mirrored on the fly)
    LRN_TRY(GM(st, n, n, n, X, 1, n, X, 1, n, Y, 1, n, 1.0, 0.0, GEMM_TRI_LOWER));
    hipLaunchKernelGGL(k_eye_plus_sym, dim3(nbl(nn)), dim3(256), 0, st, Y, n, cc, Z);
    bool pd = false;
    LRN_TRY(chol_ok(c, Z, n, w, &pd));
    if (!pd) { *ok = false; break; }
    // Y = Z^-1 X  (X and Z commute, so this is also X Z^-1)
    hipLaunchKernelGGL(k_copy, dim3(nbl(nn)), dim3(256), 0, st, X, nn, Y);
    LRN_TRY(trsm_left_lower(st, Z, n, n, nullptr, false, Y, n, n, w.trw));
    LRN_TRY(trsm_left_lower(st, Z, n, n, nullptr, true, Y, n, n, w.trw));
    // X <- sym((b/c) X + (a - b/c) Y), written through Z (free after the solves)
    hipLaunchKernelGGL(k_axpby_sym, dim3(nbl(nn)), dim3(256), 0, st, X, Y, n, b / cc, a - b / cc, Z);
    hipLaunchKernelGGL(k_copy, dim3(nbl(nn)), dim3(256), 0, st, Z, nn, X);
    ++it;
    l = l * (a + b * l2) / (1.0 + cc * l2);
    if (l > 1.0) l = 1.0;
    if (l_start < 1.0 && 1.0 - l < 1e-9) break;
  }
  if (its_out) *its_out = it;
  return LRN_OK;
}

// Qt (kk x n, ld kk) -> orthonormal rows, twice (CholeskyQR2); Gm: kk^2 scratch
int cholqr2_rows(lrn_ctx* c, double* Qt, int kk, int n, double* Gm, const Work& w, bool* ok) {
  hipStream_t st = c->stream;
  *ok = true;
  for (int pass = 0; pass < 2; ++pass) {
    LRN_TRY(GM(st, kk, kk, n, Qt, 1, kk, Qt, kk, 1, Gm, 1, kk, 1.0, 0.0, GEMM_TRI_LOWER));
    bool pd = false;
    LRN_TRY(chol_ok(c, Gm, kk, w, &pd));
    if (!pd) { *ok = false; return LRN_OK; }
    LRN_TRY(trsm_left_lower(st, Gm, kk, kk, nullptr, false, Qt, n, kk, w.trw));
  }
  return LRN_OK;
}

int leaf_jacobi(lrn_ctx* c, double* K, int n, double* V, const Work& w) {
  int sw = 0;
  LRN_TRY(jacobi_svd(c, K, V, w.vec, n, &sw, false));
  c->counts["sdc_leaf_sweeps"] += sw;
  c->counts["sdc_leaves"] += 1;
  return LRN_OK;
}

// K (n x n symmetric, destroyed) -> V (n x n, orthogonal, approximately diagonalising K)
int sdc_rec(lrn_ctx* c, double* K, int n, double* V, Bump& bump, const Work& w, int depth) {
  hipStream_t st = c->stream;
  if (n <= c->opt.sdc_leaf || depth >= 8) return leaf_jacobi(c, K, n, V, w);
  const size_t mark = bump.off;
  const long nn = (long)n * n;
  // shift: median of the diagonal; scale: 1-norm bound of K - sigma I
  std::vector<double> hd(2 * (size_t)n);
  hipLaunchKernelGGL(k_diag_colabs, dim3(n), dim3(256), 0, st, K, n, w.vec, w.vec + n);
  LRN_TRY(copy_out(c, hd.data(), w.vec, (size_t)2 * n * 8));
  std::vector<double> dg(hd.begin(), hd.begin() + n);
  std::nth_element(dg.begin(), dg.begin() + n / 2, dg.end());
  const double sigma = dg[n / 2];
  double alpha = 0.0;
  for (int j = 0; j < n; ++j) alpha = std::max(alpha, hd[n + j] + std::fabs(sigma));
  if (!(alpha > 0.0) || !std::isfinite(alpha)) return leaf_jacobi(c, K, n, V, w);
  {
    // the 1-norm overestimates ||K - sigma I||_2 by up to sqrt(n) -- on the clustered spectra of centred
    // iterates by 50x --, which costs QDWH steps: 12 power iterations give the 2-norm (x1.2 for safety;
    // the 1-norm stays the hard upper bound)
    double* xv = w.vec;
    double* yv = w.vec + n;
    hipLaunchKernelGGL(k_rand, dim3(nbl(n)), dim3(256), 0, st, yv, (long)n, (unsigned)(7 + depth));
    double nrm = 0.0;
    for (int it = 0; it < 12; ++it) {
      hipLaunchKernelGGL(k_normalize, dim3(1), dim3(1024), 0, st, yv, n, xv, w.vec + 2 * (size_t)n);
      hipLaunchKernelGGL(k_symv_shift, dim3((n + 63) / 64), dim3(256), 0, st, K, n, sigma, xv, yv);
    }
    hipLaunchKernelGGL(k_normalize, dim3(1), dim3(1024), 0, st, yv, n, xv, w.vec + 2 * (size_t)n);
    LRN_TRY(copy_out(c, &nrm, w.vec + 2 * (size_t)n, 8));
    if (nrm > 0.0 && std::isfinite(nrm)) alpha = std::min(alpha, 1.2 * nrm);
  }
  // persistent over the recursion: the two bases, the children's matrices and eigenvectors;
  // transient (released before recursing): X, Z, Y
  double* Qt = bump.take((size_t)nn + 64);
  double* Kc = bump.take((size_t)nn + 64);
  double* Vc = bump.take((size_t)nn + 64);
  const size_t keep = bump.off;
  double* X = bump.take((size_t)nn);
  double* Z = bump.take((size_t)nn);
  double* Y = bump.take((size_t)nn);
  if (!Qt || !Kc || !Vc || !X || !Z || !Y) { bump.off = mark; return leaf_jacobi(c, K, n, V, w); }
  hipLaunchKernelGGL(k_shift_scale, dim3(nbl(nn)), dim3(256), 0, st, K, n, sigma, 1.0 / alpha, X);
  int its = 0;
  bool ok = false;
  LRN_TRY(qdwh_sign(c, X, n, Z, Y, w, &its, &ok, c->opt.sdc_l0, 12));
  double tr = 0.0;
  int k = 0;
  for (int extra = 0; ok; ++extra) {
    // trace((sign + I)/2) must be an integer; an eigenvalue closer to the shift than l0 * alpha leaves it
    // fractional -> a few more (Halley) steps; the split below tolerates what remains
    hipLaunchKernelGGL(k_trace, dim3(1), dim3(256), 0, st, X, n, w.vec);
    LRN_TRY(copy_out(c, &tr, w.vec, 8));
    tr = 0.5 * (tr + n);
    k = (int)std::lround(tr);
    if (std::fabs(tr - k) < 1e-7 || extra >= 3) break;
    int it2 = 0;
    LRN_TRY(qdwh_sign(c, X, n, Z, Y, w, &it2, &ok, 1.0, 2));
    its += it2;
  }
  c->counts["sdc_qdwh_its"] += its;
  if (!ok) { bump.off = mark; c->counts["sdc_fallbacks"] += 1; return leaf_jacobi(c, K, n, V, w); }
  hipLaunchKernelGGL(k_proj, dim3(nbl(nn)), dim3(256), 0, st, X, n);          // X = P
  if (getenv("LRN_SDC_TRACE")) fprintf(stderr, "[sdc depth %d] n=%d sigma=%.4g alpha=%.4g qdwh its=%d trace(P)=%.6f k=%d\n", depth, n, sigma, alpha, its, tr, k);
  if (k < n / 16 || k > n - n / 16) {       // no useful split at this shift
    bump.off = mark;
    c->counts["sdc_fallbacks"] += 1;
    return leaf_jacobi(c, K, n, V, w);
  }
  const int k2 = n - k;
  double* Qt1 = Qt;                          // k  x n, ld k   : rows = basis of range(P)
  double* Qt2 = Qt + (((size_t)k * n + 31) & ~size_t(31));   // k2 x n, ld k2: basis of range(I - P) (256-byte aligned)
  // R random (n x n, in Z).  Qt1 = orth(R1' P): basis of range(P).  Qt2 = orth(R2' (I - Q1 Q1')): the exact
  // orthogonal complement of Q1 -- [Q1 Q2] is orthogonal whatever the quality of P; an unconverged
  // direction only leaves a small coupling Q1' K Q2 for the caller's Jacobi sweeps
  hipLaunchKernelGGL(k_rand, dim3(nbl(nn)), dim3(256), 0, st, Z, nn, (unsigned)(depth * 131 + n));
  LRN_TRY(GM(st, k, n, n, Z, n, 1, X, 1, n, Qt1, 1, k));
  bool ok1 = false, ok2 = false;
  LRN_TRY(cholqr2_rows(c, Qt1, k, n, Y, w, &ok1));
  if (ok1) {
    // Y (k2 x k) = R2' Q1 = R2' Qt1' ;  Qt2 = R2' - Y Qt1
    LRN_TRY(GM(st, k2, k, n, Z + (size_t)k * n, n, 1, Qt1, k, 1, Y, 1, k2));
    LRN_TRY(GM(st, k2, n, k, Y, 1, k2, Qt1, 1, k, Qt2, 1, k2));
    hipLaunchKernelGGL(k_rt_minus, dim3(nbl((long)k2 * n)), dim3(256), 0, st, Z, n, k, k2, Qt2);
    LRN_TRY(cholqr2_rows(c, Qt2, k2, n, Y, w, &ok2));
    if (ok2) {     // one more projection pass: CholeskyQR2 keeps Qt2 orthonormal, this keeps it orthogonal to Qt1
      LRN_TRY(GM(st, k2, k, n, Qt2, 1, k2, Qt1, k, 1, Y, 1, k2));
      LRN_TRY(GM(st, k2, n, k, Y, 1, k2, Qt1, 1, k, Qt2, 1, k2, -1.0, 1.0));
      LRN_TRY(cholqr2_rows(c, Qt2, k2, n, Y, w, &ok2));
    }
  }
  if (!ok1 || !ok2) { bump.off = mark; c->counts["sdc_fallbacks"] += 1; return leaf_jacobi(c, K, n, V, w); }
  // children: K1 = Qt1 K Qt1', K2 = Qt2 K Qt2'  (lower tiles, mirrored: exactly symmetric)
  double* K1 = Kc;
  double* K2 = Kc + (((size_t)k * k + 31) & ~size_t(31));
  LRN_TRY(GM(st, k, n, n, Qt1, 1, k, K, 1, n, Y, 1, k));
  LRN_TRY(GM(st, k, k, n, Y, 1, k, Qt1, k, 1, K1, 1, k, 1.0, 0.0, GEMM_TRI_LOWER));
  hipLaunchKernelGGL(k_mirror_lower, dim3(nbl((long)k * k)), dim3(256), 0, st, K1, k);
  LRN_TRY(GM(st, k2, n, n, Qt2, 1, k2, K, 1, n, Y, 1, k2));
  LRN_TRY(GM(st, k2, k2, n, Y, 1, k2, Qt2, k2, 1, K2, 1, k2, 1.0, 0.0, GEMM_TRI_LOWER));
  hipLaunchKernelGGL(k_mirror_lower, dim3(nbl((long)k2 * k2)), dim3(256), 0, st, K2, k2);
  bump.off = keep;                           // X, Z, Y are free again
  double* V1 = Vc;
  double* V2 = Vc + (((size_t)k * k + 31) & ~size_t(31));
  LRN_TRY(sdc_rec(c, K1, k, V1, bump, w, depth + 1));
  LRN_TRY(sdc_rec(c, K2, k2, V2, bump, w, depth + 1));
  // V = [Qt1' V1 | Qt2' V2]
  LRN_TRY(GM(st, n, k, k, Qt1, k, 1, V1, 1, k, V, 1, n));
  LRN_TRY(GM(st, n, k2, k2, Qt2, k2, 1, V2, 1, k2, V + (size_t)k * n, 1, n));
  c->counts["sdc_splits"] += 1;
  bump.off = mark;
  return LRN_OK;
}

}  // namespace

// K (n x n, symmetric positive definite, destroyed) -> V (n x n): orthogonal, V' K V nearly diagonal
int sdc_eig(lrn_ctx* c, double* K, int n, double* V) {
  const size_t nn = (size_t)n * n;
  const size_t work = (size_t)2 * n * CHOL_NB + 2 * (size_t)n + 512;
  LRN_TRY(ensure(c, c->sdcbuf, (8 * nn + work + 65536) * 8));
  LRN_TRY(ensure(c, c->info_dev, 64));
  double* base = c->sdcbuf.as<double>();
  Work w;
  w.chw = base;
  w.trw = w.chw + (size_t)n * CHOL_NB;
  w.vec = w.trw + (size_t)n * CHOL_NB;
  w.info = c->info_dev.as<int>() + 12;
  Bump bump{base + ((work + 31) & ~size_t(31)), 8 * nn + 32768, 0};
  for (const char* key : {"sdc_qdwh_its", "sdc_fallbacks", "sdc_splits", "sdc_leaves", "sdc_leaf_sweeps"}) c->counts[key] = 0;
  return sdc_rec(c, K, n, V, bump, w, 0);
}

}  // namespace lrn

extern "C" int lrn_dbg_sdc(lrn_ctx* c, int n, const double* K, double* V) {
  if (!c || n <= 0 || !K || !V) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  lrn::DBuf dk, dv;
  LRN_TRY(lrn::ensure(c, dk, (size_t)n * n * 8));
  LRN_TRY(lrn::ensure(c, dv, (size_t)n * n * 8));
  LRN_TRY(lrn::copy_in(c, dk.p, K, (size_t)n * n * 8));
  int rc = lrn::sdc_eig(c, dk.as<double>(), n, dv.as<double>());
  if (rc == LRN_OK) rc = lrn::copy_out(c, V, dv.p, (size_t)n * n * 8);
  lrn::release(dk);
  lrn::release(dv);
  return rc;
}
