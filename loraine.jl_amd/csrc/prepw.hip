// NT scaling on the device: prepare_W (reference src/prepare_W.jl:28-94).
//   X = L_X L_X', S = L_S L_S'  (blocked Cholesky, chol.hip)        :33-34
//   CC = L_S' L_X                 (MFMA GEMM)                        :39
//   CC V = U diag(D)              (one-sided Jacobi SVD, jacobi.hip) :42
//   G  = L_X V D^-1/2             (column scale + MFMA GEMM)         :60
//   Gi = D^1/2 V' L_X^-1          (blocked TRSM instead of inv(G))   :63
//   W  = G G'                     (MFMA GEMM, lower tiles + mirror)  :64
//   Si = L_S^-T L_S^-1            (TRSM with I, then GEMM)           :68
//   DDsi = 1/sqrt(diag(G' S G))   (GEMM S G + column dots)           :71-74
#include "../../include/loraine_hip.h"
#include "ctx.h"
#include "jacobi.h"
#include "ops.h"

namespace lrn {

__global__ void tril_kernel(double* __restrict__ A, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    if (i < j) A[e] = 0.0;
  }
}

__global__ void eye_kernel(double* __restrict__ V, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    V[e] = (e % n == e / n) ? 1.0 : 0.0;
}

__global__ void mirror_lower_kernel(double* __restrict__ A, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    if (i < j) A[e] = A[(long)j + (long)i * n];
  }
}

// B[:,j] = A[:,j] * f(d[j]);  mode 0: d^-1/2, 1: d^1/2, 2: d^-1
__global__ void scale_cols_kernel(const double* __restrict__ A, const double* __restrict__ d, int n, int mode,
                                  double* __restrict__ B) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int j = (int)(e / n);
    double s = mode == 2 ? d[j] : sqrt(d[j]);
    B[e] = mode == 1 ? A[e] * s : A[e] / s;
  }
}

__global__ void transpose_kernel(const double* __restrict__ A, int n, double* __restrict__ B) {
  __shared__ double tile[32][33];
  int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    int i = bx + threadIdx.x, j = by + r;
    if (i < n && j < n) tile[r][threadIdx.x] = A[(long)i + (long)j * n];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    int i = by + threadIdx.x, j = bx + r;      // B[i][j] = A[j][i]
    if (i < n && j < n) B[(long)i + (long)j * n] = tile[threadIdx.x][r];
  }
}

__global__ __launch_bounds__(256) void coldot_rsqrt_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                           int n, double* __restrict__ out) {
  __shared__ double sh[4];
  const double* a = A + (long)blockIdx.x * n;
  const double* b = B + (long)blockIdx.x * n;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += a[i] * b[i];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) sh[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = 1.0 / sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
}

static inline unsigned nb2(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

static int gemm_nn(hipStream_t st, int n, const double* A, bool tA, const double* B, bool tB, double* C, int flags = 0) {
  GemmDesc g;
  g.A = A; g.B = B; g.C = C;
  g.M = g.N = g.K = n;
  if (!tA) { g.sAm = 1; g.sAk = n; } else { g.sAm = n; g.sAk = 1; }
  if (!tB) { g.sBk = 1; g.sBn = n; } else { g.sBk = n; g.sBn = 1; }
  g.sCm = 1; g.sCn = n;
  g.flags = flags;
  return gemm(st, g);
}


int prepare_w_block(lrn_ctx* c, LmiBlock& b, int* info) {
  const int n = b.msz;
  const size_t mm = (size_t)n * n * 8;
  hipStream_t st = c->stream;
  // workspace: LX, LS, CC/tmp, V, Y, Y2  (6 n^2) + Linv blocks x2 + chol / trsm work x2
  size_t linv = chol_linv_doubles(n);
  size_t need = (6 * (size_t)n * n + 2 * linv + 2 * ((size_t)n * CHOL_NB + (size_t)CHOL_NB * n) + 4 * (size_t)n) * 8;
  LRN_TRY(ensure(c, c->scratch, need));
  double* LX = c->scratch.as<double>();
  double* LS = LX + (size_t)n * n;
  double* CC = LS + (size_t)n * n;
  double* V = CC + (size_t)n * n;
  double* Y = V + (size_t)n * n;
  double* Y2 = Y + (size_t)n * n;
  double* LinvX = Y2 + (size_t)n * n;
  double* LinvS = LinvX + linv;
  double* cw = LinvS + linv;              // n*NB
  double* tw = cw + (size_t)n * CHOL_NB;  // NB*n
  double* cw2 = tw + (size_t)CHOL_NB * n;
  double* tw2 = cw2 + (size_t)n * CHOL_NB;
  LRN_TRY(ensure(c, c->info_dev, 64));
  int* dinfo = c->info_dev.as<int>();
  int* dinfoS = dinfo + 12;
  *info = 0;
  // Two streams (option "prepw_streams"): the S side -- cholesky(S), then Si = LS^-T LS^-1, which nothing before the
  // end needs -- runs beside cholesky(X) and the Jacobi SVD, whose launches leave most of the chip idle; after the SVD
  // the triangular solve for Gi runs beside the GEMMs for G, W and DDsi.
  const bool two = c->opt.prepw_streams != 0;
  hipStream_t s2 = st;
  if (two) {
    if (!c->stream2) LRN_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    if (!c->evA) {
      LRN_HIP(c, hipEventCreateWithFlags(&c->evA, hipEventDisableTiming));
      LRN_HIP(c, hipEventCreateWithFlags(&c->evB, hipEventDisableTiming));
    }
    s2 = c->stream2;
    LRN_HIP(c, hipEventRecord(c->evA, st));                   // X, S (and the workspace) are ready
    LRN_HIP(c, hipStreamWaitEvent(s2, c->evA, 0));
  }
  // Cholesky of X (stream) and S (second stream)
  LRN_HIP(c, hipMemcpyAsync(LX, b.X.p, mm, hipMemcpyDeviceToDevice, st));
  LRN_HIP(c, hipMemsetAsync(dinfo, 0, 4, st));
  LRN_TRY(potrf_lower(st, LX, n, n, LinvX, cw, dinfo));
  hipLaunchKernelGGL(tril_kernel, dim3(nb2((long)n * n)), dim3(256), 0, st, LX, n);
  LRN_HIP(c, hipMemcpyAsync(LS, b.S.p, mm, hipMemcpyDeviceToDevice, s2));
  LRN_HIP(c, hipMemsetAsync(dinfoS, 0, 4, s2));
  LRN_TRY(potrf_lower(s2, LS, n, n, LinvS, two ? cw2 : cw, dinfoS));
  hipLaunchKernelGGL(tril_kernel, dim3(nb2((long)n * n)), dim3(256), 0, s2, LS, n);
  if (two) LRN_HIP(c, hipEventRecord(c->evB, s2));            // LS is final
  // the verdicts first, in the reference's order: X (prepare_W.jl:33), then S (:34) -- nothing else is queued on a failed
  // factor (no output is touched, every pass of the host's +1e-5 I loop costs the two factorisations only), and both
  // streams are idle when this returns
  int h[2] = {0, 0};
  LRN_HIP(c, hipMemcpyAsync(&h[0], dinfo, 4, hipMemcpyDeviceToHost, st));
  LRN_HIP(c, hipMemcpyAsync(&h[1], dinfoS, 4, hipMemcpyDeviceToHost, s2));
  LRN_HIP(c, hipStreamSynchronize(st));
  if (two) LRN_HIP(c, hipStreamSynchronize(s2));
  if (h[0] != 0 || h[1] != 0) {
    *info = h[0] != 0 ? 1 : 2;
    return LRN_OK;
  }
  // from here on an error return must not leave the second stream writing the shared scratch
  struct S2Guard { hipStream_t s; bool on; ~S2Guard() { if (on) (void)hipStreamSynchronize(s); } } guard{s2, two};
  // Si = LS^-T LS^-1   (on the second stream: overlaps everything up to the end of this function)
  hipLaunchKernelGGL(eye_kernel, dim3(nb2((long)n * n)), dim3(256), 0, s2, Y2, n);
  LRN_TRY(trsm_left_lower(s2, LS, n, n, LinvS, false, Y2, n, n, two ? tw2 : tw));
  LRN_TRY(gemm_nn(s2, n, Y2, true, Y2, false, b.Si.as<double>(), GEMM_TRI_LOWER));
  hipLaunchKernelGGL(mirror_lower_kernel, dim3(nb2((long)n * n)), dim3(256), 0, s2, b.Si.as<double>(), n);
  // SVD of CC = LS' LX = U D V' by one-sided Jacobi on CC' = LX' LS: its columns are rotated by
  // U and converge to V D, so V = (columns / D) needs no accumulation of rotations -- the rounds
  // are bandwidth-bound (every round streams the whole matrix), this removes the V half of it.
  int sweeps = 0;
  tic(c);
  LRN_TRY(gemm_nn(st, n, LX, true, LS, false, CC));
  // warm start: the left singular vectors of the previous IP iterate nearly orthogonalise the
  // columns of the new CC'
  bool warm = b.have_Vprev && c->opt.jacobi_warm;
  if (warm) {
    LRN_TRY(gemm_nn(st, n, CC, false, b.Vprev.as<double>(), false, Y));
    LRN_HIP(c, hipMemcpyAsync(CC, Y, mm, hipMemcpyDeviceToDevice, st));
  }
  toc(c, "prepw_gemm");
  tic(c);
  LRN_TRY(jacobi_svd(c, CC, nullptr, b.D.as<double>(), n, &sweeps, false));
  hipLaunchKernelGGL(scale_cols_kernel, dim3(nb2((long)n * n)), dim3(256), 0, st, CC, b.D.as<double>(), n, 2, V);
  c->counts["svd_sweeps"] = sweeps;
  toc(c, "prepw_svd");
  tic(c);
  // Gi' = LX^-T (V D^1/2)   (second stream: beside the GEMMs below; V, D and Y are free from here on)
  if (two) {
    LRN_HIP(c, hipEventRecord(c->evA, st));
    LRN_HIP(c, hipStreamWaitEvent(s2, c->evA, 0));
  }
  hipLaunchKernelGGL(scale_cols_kernel, dim3(nb2((long)n * n)), dim3(256), 0, s2, V, b.D.as<double>(), n, 1, Y);
  LRN_TRY(trsm_left_lower(s2, LX, n, n, LinvX, true, Y, n, n, two ? tw2 : tw));
  hipLaunchKernelGGL(transpose_kernel, dim3((n + 31) / 32, (n + 31) / 32), dim3(32, 8), 0, s2, Y, n, b.Gi.as<double>());
  // G = LX (V D^-1/2)
  hipLaunchKernelGGL(scale_cols_kernel, dim3(nb2((long)n * n)), dim3(256), 0, st, V, b.D.as<double>(), n, 0, CC);
  LRN_TRY(gemm_nn(st, n, LX, false, CC, false, b.G.as<double>()));
  if (c->opt.jacobi_warm) {          // U = LS' G D^-1/2  (G = LS^-T U D^1/2) for the next warm start
    LRN_TRY(ensure(c, b.Vprev, mm));
    LRN_TRY(gemm_nn(st, n, LS, true, b.G.as<double>(), false, CC));
    hipLaunchKernelGGL(scale_cols_kernel, dim3(nb2((long)n * n)), dim3(256), 0, st, CC, b.D.as<double>(), n, 0,
                       b.Vprev.as<double>());
    b.have_Vprev = true;
  }
  // W = G G'
  LRN_TRY(gemm_nn(st, n, b.G.as<double>(), false, b.G.as<double>(), true, b.W.as<double>(), GEMM_TRI_LOWER));
  hipLaunchKernelGGL(mirror_lower_kernel, dim3(nb2((long)n * n)), dim3(256), 0, st, b.W.as<double>(), n);
  // DDsi = 1/sqrt(diag(G' S G))
  LRN_TRY(gemm_nn(st, n, b.S.as<double>(), false, b.G.as<double>(), false, CC));
  hipLaunchKernelGGL(coldot_rsqrt_kernel, dim3(n), dim3(256), 0, st, b.G.as<double>(), CC, n, b.DDsi.as<double>());
  if (two) {                                                  // join: Si and Gi are complete when this call returns
    LRN_HIP(c, hipEventRecord(c->evB, s2));
    LRN_HIP(c, hipStreamWaitEvent(st, c->evB, 0));
  }
  toc(c, "prepw_gemm");
  LRN_HIP(c, hipGetLastError());
  guard.on = false;                                           // (joined through evB above)
  b.have_W = b.have_G = true;
  b.nt_free = false;
  c->scal_version += 1;
  return LRN_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Eigen-free NT scaling (round 3).  The reference takes an SVD of L_S'L_X (prepare_W.jl:39-64) to get G, Gi, D; what
// the iteration consumes of them can be written with K = L_X' S L_X = V D^2 V' alone:
//   W = G G' = L_X K^-1/2 L_X'                                   (prepare_W.jl:64)
//   with B = L_X' dS L_X, T = K^-1/2 B K^-1/2, TX = L_X^-1 dX L_X^-T = -I - T (+ sm K^-1 + R in the corrector):
//   dX = L_X TX L_X' (predictor_corrector.jl:255,257); the scaled directions of the step-length rule (:263-285) are
//   orthogonally similar to TX and T (DDsi.*(Gi dX Gi').*DDsi = V' TX V, DDsi.*(G' dS G).*DDsi = V' T V), same eigmin;
//   G RNT G' (:186, :257, :309) = L_X R L_X' with K^1/2 R + R K^1/2 = -(N K^-1/2 + K^-1/2 N'), N = L_X^-1 dX dS L_X = TX B
//   G (G'RdG + D - sm/D - RNT) G' (:186) = W Rd W + X - sm Si - L_X R L_X'
// Nothing is inverted: products with an explicit L_X^-1 lose the directions through cancellation once cond(X) passes
// 1e10 (control1 left the trajectory at iteration 21 that way); in the L_X basis every term is O(1).
// K is well conditioned along a solve (D^2 = eig(XS) stays near the central path: cond(K) <~ 1e3 where cond(X) reaches
// 1e12), so Y = (K/c)^1/2 and Z = (K/c)^-1/2 come from the coupled Newton-Schulz iteration -- products only, on the MFMA:
//   P = Z Y, T = a (3 I - a^2 P)/2, Y <- Y T, Z <- T Z,
// taken LITERALLY: every iterate is symmetric in exact arithmetic, but the iteration is only stable as written -- with
// any product replaced by its transposed twin (Z Y' for Z Y), or with the iterates symmetrised after each step, the
// rounding-level commutator grows by a factor ~ cond(K)^1/2 per step once the residual is at its floor (measured on
// thetaG11: 2e-7 -> 1e+90 in eleven steps; /tmp-free NumPy reproduction in DESIGN.md).  The direct-to-LDS GEMM computes
// A Bm', so every product also stores its transpose (GemmDesc::C2) for the next one to read.
// With the one-sided scaling a = sqrt(3/(1 + l + l^2)), l <- a l (3 - a^2 l^2)/2 for spec(P)^1/2 in [l, 1]: the map never
// leaves (0, 1], so a wrong guess of l costs steps, never correctness.  c = min(||K||_1, ||K||_F) >= lambda_max(K).
// The steps are queued without host round trips; ||I - P||_F of every step comes back in one copy.
// tools/nt_eigenfree_proto.py emulates the whole route in NumPy against the SVD route of the oracle.
__global__ __launch_bounds__(256) void colnorm_kernel(const double* __restrict__ K, int n, double* __restrict__ colsum,
                                                      double* __restrict__ colsq) {
  __shared__ double sh[8];
  const double* a = K + (long)blockIdx.x * n;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { double v = a[i]; s += fabs(v); q += v * v; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[w] = s; sh[4 + w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    colsum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    colsq[blockIdx.x] = sh[4] + sh[5] + sh[6] + sh[7];
  }
}

// sc[0] = c = min(max_j colsum_j, sqrt(sum_j colsq_j)), sc[1] = 1/c   (one workgroup, fixed order)
__global__ __launch_bounds__(256) void normc_kernel(const double* __restrict__ colsum, const double* __restrict__ colsq,
                                                    int n, double* __restrict__ sc) {
  __shared__ double shm[4], shq[4];
  double mx = 0.0, q = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) { mx = fmax(mx, colsum[j]); q += colsq[j]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mx = fmax(mx, __shfl_down(mx, off, 64)); q += __shfl_down(q, off, 64); }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { shm[w] = mx; shq[w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m1 = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
    double fr = sqrt(shq[0] + shq[1] + shq[2] + shq[3]);
    double cc = fmin(m1, fr);
    if (!(cc > 0.0)) cc = 1.0;
    sc[0] = cc;
    sc[1] = 1.0 / cc;
  }
}

__global__ void set_sc_kernel(double* __restrict__ sc, double cval) {
  sc[0] = cval;
  sc[1] = 1.0 / cval;
}

__global__ void scale_dev_kernel(const double* __restrict__ K, const double* __restrict__ sc, long total,
                                 double* __restrict__ Y) {
  const double f = sc[1];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) Y[e] = K[e] * f;
}

// T = a (3 I - a^2 P) / 2;  part[block] = sum (delta_ij - P_ij)^2 over the block's elements
__global__ __launch_bounds__(256) void ns_t_kernel(const double* __restrict__ P, int n, double a, double* __restrict__ T,
                                                   double* __restrict__ part) {
  __shared__ double sh[4];
  const long total = (long)n * n;
  const double a3 = 0.5 * a * a * a, a1 = 1.5 * a;
  double s = 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const double p = P[e];
    const bool dg = (e % n) == (e / n);
    const double r = (dg ? 1.0 : 0.0) - p;
    s += r * r;
    T[e] = (dg ? a1 : 0.0) - a3 * p;
  }
  if (!part) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void sum_sqrt_kernel(const double* __restrict__ part, int np, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int e = threadIdx.x; e < np; e += 256) s += part[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
}

// C = alpha A Bm'  (all n x n column-major): both operands contiguous along the result's dimensions -> direct-to-LDS kernel
int gemm_nt(hipStream_t st, int n, const double* A, const double* Bm, double* C, int flags, double alpha, double* Ct) {
  GemmDesc g;
  g.A = A; g.sAm = 1; g.sAk = n;
  g.B = Bm; g.sBk = n; g.sBn = 1;
  g.C = C; g.sCm = 1; g.sCn = n;
  g.C2 = Ct;
  g.M = g.N = g.K = n;
  g.alpha = alpha;
  g.flags = flags;
  return gemm(st, g);
}

int gemm_nt_slabs(hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha, SlabSrc* src) {
  GemmDesc g;
  g.A = A; g.sAm = 1; g.sAk = n;
  g.B = Bm; g.sBk = n; g.sBn = 1;
  g.C = C; g.sCm = 1; g.sCn = n;
  g.M = g.N = g.K = n;
  g.alpha = alpha;
  return gemm_slabs(st, g, src);
}

// C = sum of the slabs, Ct = its transpose: 32 x 32 tiles through LDS (round 4: the slab addition of a mid-size product and
// the transpose pass that followed it were two launches and two trips through memory)
__global__ __launch_bounds__(256) void slabs_transpose_kernel(SlabSrc src, int n, double* __restrict__ C, double* __restrict__ Ct) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bx + tx, j = by + r;
    if (i < n && j < n) {
      const long e = (long)i + (long)j * n;
      const double v = slab_sum(src, e);
      tile[r][tx] = v;
      if (src.n > 1 || src.p != C) C[e] = v;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = by + tx, j = bx + r;      // Ct[i][j] = C[j][i]
    if (i < n && j < n) Ct[(long)i + (long)j * n] = tile[tx][r];
  }
}

void slabs_to_c_and_ct(hipStream_t st, const SlabSrc& src, int n, double* C, double* Ct) {
  hipLaunchKernelGGL(slabs_transpose_kernel, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, st, src, n, C, Ct);
}

// C = (S + S') / 2 for S = the sum of the slabs (or C itself, in place): the tile pair (bi, bj), (bj, bi) by one workgroup.
// T != null: the Newton-Schulz pass on P = C in the same sweep -- T = a (3 I - a^2 P) / 2 and this workgroup's share of
// ||I - P||_F^2 in part[blockIdx.x] (ns_t_kernel's work; C may then be null: nobody reads P itself)
__global__ __launch_bounds__(256) void slabs_sym_kernel(SlabSrc src, int n, double* __restrict__ C, double a, double* __restrict__ T,
                                                        double* __restrict__ part) {
  __shared__ double ta[32][33], tb[32][33];
  __shared__ double sh[4];
  const int nt = (n + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // pair index -> (bi <= bj)
  int bj = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long)(bj + 1) * (bj + 2) / 2 <= (long)blockIdx.x) ++bj;
  while ((long)bj * (bj + 1) / 2 > (long)blockIdx.x) --bj;
  const int bi = (int)(blockIdx.x - (long)bj * (bj + 1) / 2);
  if (bj >= nt) return;
  const int oi = bi * 32, oj = bj * 32;
  for (int r = ty; r < 32; r += 8) {
    const int i = oi + tx, j = oj + r;       // tile (bi, bj): element (i, j)
    ta[r][tx] = (i < n && j < n) ? slab_sum(src, (long)i + (long)j * n) : 0.0;
    const int i2 = oj + tx, j2 = oi + r;     // tile (bj, bi): element (i2, j2)
    tb[r][tx] = (i2 < n && j2 < n) ? slab_sum(src, (long)i2 + (long)j2 * n) : 0.0;
  }
  __syncthreads();
  const double a3 = 0.5 * a * a * a, a1 = 1.5 * a;
  double acc = 0.0;
  for (int r = ty; r < 32; r += 8) {
    const int i = oi + tx, j = oj + r;
    if (i < n && j < n) {
      const double v = 0.5 * (ta[r][tx] + tb[tx][r]);
      if (C) C[(long)i + (long)j * n] = v;
      if (T) {
        const double rr = (i == j ? 1.0 : 0.0) - v;
        acc += rr * rr;
        T[(long)i + (long)j * n] = (i == j ? a1 : 0.0) - a3 * v;
      }
    }
    const int i2 = oj + tx, j2 = oi + r;
    if (bi != bj && i2 < n && j2 < n) {
      const double v = 0.5 * (tb[r][tx] + ta[tx][r]);
      if (C) C[(long)i2 + (long)j2 * n] = v;
      if (T) {
        acc += v * v;                                  // (off the diagonal: the residual entry is -v)
        T[(long)i2 + (long)j2 * n] = -a3 * v;
      }
    }
  }
  if (!T || !part) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// The same from slabs that hold only the LOWER 64-tiles of the product (rows i, columns j with i / 64 >= j / 64; gemm_slabs
// with GEMM_TRI_LOWER: half the MFMA work).  The pair of 32-tiles (bi <= bj) is filled from the lower one, (bj, bi): C over
// there = the sum, C over here = its transpose; inside a diagonal 32-tile the lower triangle is mirrored.  The result is
// the product's lower triangle mirrored -- exactly symmetric, NOT the average of the two triangles that slabs_sym_kernel
// forms (the form the products of msz >= 1500 have had since round 3: lower tiles + mirror).
__global__ __launch_bounds__(256) void slabs_symlow_kernel(SlabSrc src, int n, double* __restrict__ C, double a, double* __restrict__ T,
                                                           double* __restrict__ part) {
  __shared__ double tl[32][33];
  __shared__ double sh[4];
  const int nt = (n + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  int bj = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long)(bj + 1) * (bj + 2) / 2 <= (long)blockIdx.x) ++bj;
  while ((long)bj * (bj + 1) / 2 > (long)blockIdx.x) --bj;
  const int bi = (int)(blockIdx.x - (long)bj * (bj + 1) / 2);
  if (bj >= nt) return;
  const int oi = bi * 32, oj = bj * 32;          // lower tile: rows oj .., columns oi ..
  for (int r = ty; r < 32; r += 8) {
    const int i = oj + tx, j = oi + r;           // element (i, j) of the lower tile, i fastest
    tl[r][tx] = (i < n && j < n) ? slab_sum(src, (long)i + (long)j * n) : 0.0;
  }
  __syncthreads();
  const double a3 = 0.5 * a * a * a, a1 = 1.5 * a;
  double acc = 0.0;
  for (int r = ty; r < 32; r += 8) {
    {   // the lower tile itself: element (oj + tx, oi + r)
      const int i = oj + tx, j = oi + r;
      if (i < n && j < n) {
        const double v = (bi == bj && tx < r) ? tl[tx][r] : tl[r][tx];       // diagonal tile: (i, j) above the diagonal <- (j, i)
        if (C) C[(long)i + (long)j * n] = v;
        if (T) {
          const double rr = (i == j ? 1.0 : 0.0) - v;
          acc += rr * rr;
          T[(long)i + (long)j * n] = (i == j ? a1 : 0.0) - a3 * v;
        }
      }
    }
    if (bi != bj) {   // its mirror image: element (oi + tx, oj + r) = lower (oj + r, oi + tx)
      const int i = oi + tx, j = oj + r;
      if (i < n && j < n) {
        const double v = tl[tx][r];
        if (C) C[(long)i + (long)j * n] = v;
        if (T) {
          acc += v * v;
          T[(long)i + (long)j * n] = -a3 * v;
        }
      }
    }
  }
  if (!T || !part) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// the lower 64-tiles of C = alpha A Bm' as split-K slabs (src->n == 0: not available for this size -- take the full product)
static int gemm_nt_lower_slabs(hipStream_t st, int n, const double* A, const double* Bm, double alpha, SlabSrc* src) {
  static const bool off = getenv("LRN_SYM_LOWER_OFF") != nullptr;      // (measurement knob)
  src->p = nullptr; src->stride = 0; src->n = 0;
  if (off) return LRN_OK;
  GemmDesc g;
  g.A = A; g.sAm = 1; g.sAk = n;
  g.B = Bm; g.sBk = n; g.sBn = 1;
  g.C = nullptr; g.sCm = 1; g.sCn = n;
  g.M = g.N = g.K = n;
  g.alpha = alpha;
  g.flags = GEMM_TRI_LOWER;
  return gemm_slabs(st, g, src);
}

// P = A Bm' symmetrised (not stored) -> T = a (3 I - a^2 P) / 2, partial sums of ||I - P||_F^2 in part[0 .. *npart): the
// Newton-Schulz step's first product with its element-wise pass folded into the slab addition (msz < 1500, one rank)
int gemm_nt_sym_ns(hipStream_t st, int n, const double* A, const double* Bm, double* scratchC, double a, double* T, double* part,
                   int* npart) {
  SlabSrc src;
  const long nt = (n + 31) / 32;
  *npart = (int)(nt * (nt + 1) / 2);
  LRN_TRY(gemm_nt_lower_slabs(st, n, A, Bm, 1.0, &src));
  if (src.n > 0) {
    hipLaunchKernelGGL(slabs_symlow_kernel, dim3((unsigned)*npart), dim3(256), 0, st, src, n, (double*)nullptr, a, T, part);
    return LRN_OK;
  }
  LRN_TRY(gemm_nt_slabs(st, n, A, Bm, scratchC, 1.0, &src));
  hipLaunchKernelGGL(slabs_sym_kernel, dim3((unsigned)*npart), dim3(256), 0, st, src, n, (double*)nullptr, a, T, part);
  return LRN_OK;
}

// (M + M')/2 in place
__global__ void sym_inplace_kernel(double* __restrict__ M, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    if (i < j) {
      const long f = (long)j + (long)i * n;
      const double v = 0.5 * (M[e] + M[f]);
      M[e] = v;
      M[f] = v;
    }
  }
}

// C = alpha A Bm' for a product that is symmetric in exact arithmetic, returned exactly symmetric: lower tiles + mirror on
// the 128-tile direct-to-LDS kernel where that fills the chip, the plain product and a symmetrising pass below (at msz 800
// the 28 lower tiles of 128 take 121 us, the full product on 64-tiles 35 us)
// upper := lower inside the 128 x 128 diagonal tiles (GEMM_C_MIRROR mirrors the tiles below the diagonal only)
__global__ __launch_bounds__(256) void mirror_diag_tiles_kernel(double* __restrict__ C, int n) {
  const int t0 = blockIdx.x * 128;
  for (int e = threadIdx.x; e < 128 * 128; e += 256) {
    const int i = t0 + (e & 127), j = t0 + (e >> 7);
    if (i < n && j < n && i < j) C[(long)i + (long)j * n] = C[(long)j + (long)i * n];
  }
}

int gemm_nt_sym(hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha, int tri) {
  if (n >= 1500) {
    LRN_TRY(gemm_nt(st, n, A, Bm, C, GEMM_TRI_LOWER | GEMM_C_MIRROR | ((n & 1) ? 0 : tri), alpha));
    hipLaunchKernelGGL(mirror_diag_tiles_kernel, dim3((n + 127) / 128), dim3(256), 0, st, C, n);
    return LRN_OK;
  }
  SlabSrc src;
  const long nt = (n + 31) / 32;
  LRN_TRY(gemm_nt_lower_slabs(st, n, A, Bm, alpha, &src));
  if (src.n > 0) {      // (round 4: lower 64-tiles only + mirror, as the products of msz >= 1500)
    hipLaunchKernelGGL(slabs_symlow_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, src, n, C, 0.0, (double*)nullptr,
                       (double*)nullptr);
    return LRN_OK;
  }
  LRN_TRY(gemm_nt_slabs(st, n, A, Bm, C, alpha, &src));
  hipLaunchKernelGGL(slabs_sym_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, src, n, C, 0.0, (double*)nullptr,
                     (double*)nullptr);
  return LRN_OK;
}

bool products_sharded(const lrn_ctx* c, hipStream_t st, int n) {
  return c->comm && c->world > 1 && c->opt.shard_products != 0 && st == c->stream && n >= c->opt.shard_products_min;
}

static int shard_cols(const lrn_ctx* c, int n, int* c0, int* c1) {
  const int cb = (((n + c->world - 1) / c->world) + 15) & ~15;      // 16-column granularity: aligned operand pointers
  *c0 = std::min(n, c->rank * cb);
  *c1 = std::min(n, *c0 + cb);
  return cb;
}

// `tri`: which operand is triangular with explicit zeros in its other triangle (GEMM_KFROM_M / _N: zero for k < m / k < n,
// GEMM_KTO_M / _N: zero for k > m / k > n; one flag) -- the K loop of every tile then covers only the range where that
// operand is not zero: half the work of the products with L_X, L_X', L_S^-T (bitwise the same sums: the skipped terms
// are exact zeros).  Taken where the 128-tile kernel runs (n >= 1500, even); the sharded product ignores it.
static inline int tri_hint(int n, int tri) { return (n >= 1500 && (n & 1) == 0) ? tri : 0; }

int pgemm_nt(lrn_ctx* c, hipStream_t st, int n, const double* A, const double* Bm, double* C, int tri, double alpha,
             double* Ct) {
  if (!products_sharded(c, st, n)) return gemm_nt(st, n, A, Bm, C, tri_hint(n, tri), alpha, Ct);
  int c0, c1;
  const int cb = shard_cols(c, n, &c0, &c1);
  if (c1 > c0) {
    GemmDesc g;                      // C[:, c0:c1] = alpha A Bm[c0:c1, :]'
    g.A = A; g.sAm = 1; g.sAk = n;
    g.B = Bm + c0; g.sBk = n; g.sBn = 1;
    g.C = C + (long)c0 * n; g.sCm = 1; g.sCn = n;
    g.M = n; g.N = c1 - c0; g.K = n;
    g.alpha = alpha;
    LRN_TRY(gemm(st, g));
  }
  LRN_TRY(comm_allgather_cols(c, C, n, cb));
  c->counts["pgemm_sharded"] += 1;
  if (Ct) hipLaunchKernelGGL(transpose_kernel, dim3((n + 31) / 32, (n + 31) / 32), dim3(32, 8), 0, st, C, n, Ct);
  return LRN_OK;
}

int pgemm_nt_sym(lrn_ctx* c, hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha, int tri) {
  if (!products_sharded(c, st, n)) return gemm_nt_sym(st, n, A, Bm, C, alpha, tri);
  LRN_TRY(pgemm_nt(c, st, n, A, Bm, C, 0, alpha, nullptr));
  hipLaunchKernelGGL(sym_inplace_kernel, dim3(nb2((long)n * n)), dim3(256), 0, st, C, n);
  return LRN_OK;
}

// smallest pivot of a Cholesky factor: out = min_i L_ii^2
__global__ __launch_bounds__(256) void min_pivot_kernel(const double* __restrict__ L, int n, double* __restrict__ out) {
  __shared__ double sh[4];
  double m = 1e300;
  for (int i = threadIdx.x; i < n; i += 256) { const double v = L[(long)i * n + i]; m = fmin(m, v * v); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
}

int nt_factor(lrn_ctx* c, LmiBlock& b, int* info, double* minpiv) {
  const int n = b.msz;
  const size_t nn = (size_t)n * n, mm = nn * 8;
  hipStream_t st = c->stream;
  *info = 0;
  b.chol_valid = false;
  LRN_TRY(ensure(c, b.LXf, mm));
  LRN_TRY(ensure(c, b.LSf, mm));
  const size_t linv = chol_linv_doubles(n);
  LRN_TRY(ensure(c, c->lxbuf, (2 * linv + 2 * (size_t)n * CHOL_NB + 16) * 8));
  double* LinvX = c->lxbuf.as<double>();
  double* LinvS = LinvX + linv;
  double* cw = LinvS + linv;
  double* cw2 = cw + (size_t)n * CHOL_NB;
  double* piv = cw2 + (size_t)n * CHOL_NB;
  LRN_TRY(ensure(c, c->info_dev, 64));
  int* dinfo = c->info_dev.as<int>();
  int* dinfoS = dinfo + 12;
  if (!c->stream2) LRN_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  if (!c->evA) {
    LRN_HIP(c, hipEventCreateWithFlags(&c->evA, hipEventDisableTiming));
    LRN_HIP(c, hipEventCreateWithFlags(&c->evB, hipEventDisableTiming));
  }
  hipStream_t s2 = c->opt.prepw_streams ? c->stream2 : st;
  const bool two = s2 != st;
  const unsigned ge = nb2((long)nn);
  double *LX = b.LXf.as<double>(), *LS = b.LSf.as<double>();
  if (two) {
    LRN_HIP(c, hipEventRecord(c->evA, st));
    LRN_HIP(c, hipStreamWaitEvent(s2, c->evA, 0));
  }
  tic(c);
  LRN_HIP(c, hipMemcpyAsync(LX, b.X.p, mm, hipMemcpyDeviceToDevice, st));
  LRN_HIP(c, hipMemsetAsync(dinfo, 0, 4, st));
  LRN_TRY(potrf_lower(st, LX, n, n, LinvX, cw, dinfo));
  hipLaunchKernelGGL(tril_kernel, dim3(ge), dim3(256), 0, st, LX, n);
  hipLaunchKernelGGL(min_pivot_kernel, dim3(1), dim3(256), 0, st, LX, n, piv);
  LRN_HIP(c, hipMemcpyAsync(LS, b.S.p, mm, hipMemcpyDeviceToDevice, s2));
  LRN_HIP(c, hipMemsetAsync(dinfoS, 0, 4, s2));
  LRN_TRY(potrf_lower(s2, LS, n, n, LinvS, two ? cw2 : cw, dinfoS));
  hipLaunchKernelGGL(tril_kernel, dim3(ge), dim3(256), 0, s2, LS, n);
  hipLaunchKernelGGL(min_pivot_kernel, dim3(1), dim3(256), 0, s2, LS, n, piv + 1);
  // the verdicts, in the reference's order (prepare_W.jl:33-34)
  int h[2] = {0, 0};
  double hp[2] = {0.0, 0.0};
  LRN_HIP(c, hipMemcpyAsync(&h[0], dinfo, 4, hipMemcpyDeviceToHost, st));
  LRN_HIP(c, hipMemcpyAsync(&hp[0], piv, 8, hipMemcpyDeviceToHost, st));
  LRN_HIP(c, hipMemcpyAsync(&h[1], dinfoS, 4, hipMemcpyDeviceToHost, s2));
  LRN_HIP(c, hipMemcpyAsync(&hp[1], piv + 1, 8, hipMemcpyDeviceToHost, s2));
  LRN_HIP(c, hipStreamSynchronize(st));
  if (two) LRN_HIP(c, hipStreamSynchronize(s2));
  toc(c, "prepw_chol");
  if (h[0] != 0 || h[1] != 0) {
    *info = h[0] != 0 ? 1 : 2;
    return LRN_OK;
  }
  if (minpiv) { minpiv[0] = hp[0]; minpiv[1] = hp[1]; }
  b.chol_valid = true;
  return LRN_OK;
}

int prepare_w_ns(lrn_ctx* c, LmiBlock& b, int* info, bool* converged) {
  const int n = b.msz;
  const size_t nn = (size_t)n * n, mm = nn * 8;
  hipStream_t st = c->stream;
  *info = 0;
  *converged = false;
  b.nt_free = false;
  if (!b.chol_valid) {
    LRN_TRY(nt_factor(c, b, info, nullptr));
    if (*info != 0) return LRN_OK;
  }
  b.chol_valid = false;            // (consumed: whoever changes X, S afterwards need not remember to reset it)
  for (DBuf* d : {&b.LXt, &b.Yh, &b.Zh, &b.Ki, &b.Bs, &b.TX, &b.Qm}) LRN_TRY(ensure(c, *d, mm));
  const int npart = (int)std::min<size_t>(1024, (nn + 255) / 256);
  const int npart_alloc = std::max(npart, (int)((((long)n + 31) / 32) * (((long)n + 31) / 32 + 1) / 2));      // (gemm_nt_sym_ns: one per tile pair)
  const int maxit = std::max(4, std::min(c->opt.ns_maxit, 120));
  // scratch: P, T, Y' Z' of the resident set, a second set Ya Ya' Za Za', L_S^-1, L_S^-T (10 n^2), trsm work,
  // column norms, partial sums, residuals
  size_t need = (10 * nn + (size_t)CHOL_NB * n + 2 * (size_t)n + npart_alloc + maxit + 64) * 8;
  LRN_TRY(ensure(c, c->scratch, need));
  double* LXt = b.LXt.as<double>();
  double* Pm = c->scratch.as<double>();
  double* Tm = Pm + nn;
  double* Yt0 = Tm + nn;
  double* Zt0 = Yt0 + nn;
  double* Ya = Zt0 + nn;
  double* Yta = Ya + nn;
  double* Za = Yta + nn;
  double* Zta = Za + nn;
  double* LSi = Zta + nn;
  double* LSit = LSi + nn;
  double* tw2 = LSit + nn;
  double* colsum = tw2 + (size_t)CHOL_NB * n;
  double* colsq = colsum + n;
  double* part = colsq + n;
  double* res = part + npart_alloc;  // res[0..maxit), then sc[0..1]
  double* sc = res + maxit;
  double *LX = b.LXf.as<double>(), *LS = b.LSf.as<double>();
  // (sharded products carry collectives: everything on the context's stream then, the order of the calls is the order of
  // the collectives on every rank)
  const bool shp = products_sharded(c, st, n);
  hipStream_t s2 = (c->opt.prepw_streams && !shp) ? c->stream2 : st;
  const bool two = s2 != st;
  const unsigned ge = nb2((long)nn);
  if (two) {
    LRN_HIP(c, hipEventRecord(c->evA, st));
    LRN_HIP(c, hipStreamWaitEvent(s2, c->evA, 0));
  }
  tic(c);
  const dim3 tg((n + 31) / 32, (n + 31) / 32), tb(32, 8);
  double* LSt = Zta;                                 // (free until the first Newton-Schulz step, which st orders after its reader)
  hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, s2, LS, n, LSt);
  if (two) LRN_HIP(c, hipEventRecord(c->evB, s2));
  // (round 4: S^-1 no longer through L_S^-1 -- a triangular solve with msz right-hand sides, 436 ms of a 2 s iteration at
  // msz 10^4 -- but from what the iteration produces anyway: K = L_X' S L_X  =>  S^-1 = L_X K^-1 L_X', see the end)
  hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, st, LX, n, LXt);
  // K = CC' CC with CC = L_S' L_X (prepare_W.jl:39) -- NOT L_X' S L_X: with cond(X), cond(S) at 1e10 the entries of
  // |L_X'| |S| |L_X| are 1e10 times those of K and the explicit product has no correct digit left (measured: the
  // Newton-Schulz iteration then diverges); through the factors the error is that of CC itself, which the SVD shares
  double* Y = b.Yh.as<double>();
  double* Z = b.Zh.as<double>();
  if (two) LRN_HIP(c, hipStreamWaitEvent(st, c->evB, 0));
  LRN_TRY(pgemm_nt(c, st, n, LXt, LSt, Pm, GEMM_KFROM_M, 1.0));                           // CC' = L_X' L_S (L_X' upper)
  LRN_TRY(pgemm_nt_sym(c, st, n, Pm, Pm, Tm, 1.0));                                        // K = CC' CC
  hipLaunchKernelGGL(colnorm_kernel, dim3(n), dim3(256), 0, st, Tm, n, colsum, colsq);
  hipLaunchKernelGGL(normc_kernel, dim3(1), dim3(256), 0, st, colsum, colsq, n, sc);
  // Round 4: where a step of the iteration is expensive (three n^3 products; msz >= ns_lanczos_min) the scale c and the
  // lower end l of the schedule come from a short Lanczos run on K instead of norm bounds and a fixed guess: the norm bound
  // min(||K||_1, ||K||_F) is 1.4-7 x lambda_max on the iterates of theta1 / control1 / maxG11 (more as msz grows) and the
  // assumed l^2 = 2e-3 four to forty times below lambda_min / lambda_max -- two to three steps of eight to ten.  c keeps a
  // margin over the Ritz value (an eigenvalue above c would turn negative under the scaled map: the residual test then
  // sends the iteration to the SVD route); l may be wrong in either direction (costs steps, never correctness).
  double ell2 = std::min(std::max(c->opt.ns_l0, 1e-12), 0.25);
  if (c->opt.ns_lanczos != 0 && n >= c->opt.ns_lanczos_min) {
    double lo = 0.0, hi = 0.0, rh = 0.0;
    LRN_TRY(lanczos_ends(c, Tm, n, 24, &lo, &hi, &rh));
    double cn[2] = {0.0, 0.0};
    LRN_TRY(copy_out(c, cn, sc, 16));                              // the norm bound (always valid)
    const double cl = 1.05 * (hi + rh);
    if (hi > 0.0 && cl > 0.0 && cl < cn[0]) {
      hipLaunchKernelGGL(set_sc_kernel, dim3(1), dim3(1), 0, st, sc, cl);
      if (lo > 0.0) ell2 = std::min(0.25, std::max(ell2, 0.5 * lo / cl));
      c->counts["ns_lanczos_scaled"] += 1;
    }
  }
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ge), dim3(256), 0, st, Tm, sc, (long)nn, Y);
  toc(c, "prepw_gemm");
  tic(c);
  // Newton-Schulz: the planned steps are the scaled ones plus the plain steps of the quadratic phase
  double ell = std::sqrt(ell2);
  double *Yc = Y, *Ytc = Yt0, *Zc = Z, *Ztc = Zt0;        // current set (Y, Y', Z, Z')
  double *Yn = Ya, *Ytn = Yta, *Zn = Za, *Ztn = Zta;      // next set
  hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, st, Yc, n, Ytc);
  bool z_is_eye = true;
  int k = 0;
  // the transposed twin of a product: stored by the GEMM epilogue (8-byte scattered stores: 61 against 35 us at msz 800 on
  // the 64-tile kernel) or by a transpose pass of its own (8 us there)
  const bool dual = c->opt.ns_dual == 1 || (c->opt.ns_dual < 0 && n >= 1500);
  auto prod = [&](hipStream_t sx, const double* A, const double* Bm, double* C, double* Ct) -> int {
    if (dual) return pgemm_nt(c, sx, n, A, Bm, C, 0, 1.0, Ct);
    if (!products_sharded(c, sx, n)) {
      SlabSrc src;                                    // (a mid-size product: its slabs are added by the transpose pass)
      LRN_TRY(gemm_nt_slabs(sx, n, A, Bm, C, 1.0, &src));
      slabs_to_c_and_ct(sx, src, n, C, Ct);
      return LRN_OK;
    }
    LRN_TRY(pgemm_nt(c, sx, n, A, Bm, C, 0, 1.0));
    hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, sx, C, n, Ct);
    return LRN_OK;
  };
  // Y T and T Z are independent: the second one on a third stream, so that two workgroups share every CU where one
  // product alone (256 tiles of 128 at msz 2000) leaves each CU a single workgroup
  hipStream_t s3 = st;
  static const int s3_min = getenv("LRN_NS_S3_MIN") ? atoi(getenv("LRN_NS_S3_MIN")) : 1024;      // (measurement knob)
  if (c->opt.prepw_streams && n >= s3_min && !shp) {
    if (!c->stream3) {
      LRN_HIP(c, hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
      LRN_HIP(c, hipEventCreateWithFlags(&c->evC, hipEventDisableTiming));
      LRN_HIP(c, hipEventCreateWithFlags(&c->evD, hipEventDisableTiming));
    }
    s3 = c->stream3;
  }
  auto one_step = [&](double a) -> int {
    // P = Z Y is the one product that may be symmetrised (lower tiles + mirror where that halves its cost): with T exactly
    // symmetric and Y T, T Z taken literally the iteration stays stable (residual floor 1e-12 instead of 1e-13, NumPy and
    // device) -- and T is its own transposed twin
    const double* Pk = Yc;                                // Z = I: P = Y (exactly symmetric)
    int np_k = npart;
    if (!z_is_eye && n < 1500 && !products_sharded(c, st, n)) {
      // (mid sizes: symmetrisation, T and the residual in the pass that adds the product's slabs)
      LRN_TRY(gemm_nt_sym_ns(st, n, Zc, Ytc, Pm, a, Tm, part, &np_k));
    } else {
      if (!z_is_eye) { LRN_TRY(pgemm_nt_sym(c, st, n, Zc, Ytc, Pm, 1.0)); Pk = Pm; }
      hipLaunchKernelGGL(ns_t_kernel, dim3(npart), dim3(256), 0, st, Pk, n, a, Tm, part);
    }
    hipLaunchKernelGGL(sum_sqrt_kernel, dim3(1), dim3(256), 0, st, part, np_k, res + k);
    double* const Tt = Tm;
    if (!z_is_eye && s3 != st) {
      LRN_HIP(c, hipEventRecord(c->evC, st));                                          // T, T' are final
      LRN_HIP(c, hipStreamWaitEvent(s3, c->evC, 0));
      LRN_TRY(prod(s3, Tm, Ztc, Zn, Ztn));                                             // T Z
      LRN_HIP(c, hipEventRecord(c->evD, s3));
    }
    LRN_TRY(prod(st, Yc, Tt, Yn, Ytn));                                                // Y T
    if (z_is_eye) {
      LRN_HIP(c, hipMemcpyAsync(Zn, Tm, mm, hipMemcpyDeviceToDevice, st));
      LRN_HIP(c, hipMemcpyAsync(Ztn, Tt, mm, hipMemcpyDeviceToDevice, st));
      z_is_eye = false;
    } else if (s3 != st) {
      LRN_HIP(c, hipStreamWaitEvent(st, c->evD, 0));
    } else {
      LRN_TRY(prod(st, Tm, Ztc, Zn, Ztn));                                             // T Z
    }
    std::swap(Yc, Yn); std::swap(Ytc, Ytn); std::swap(Zc, Zn); std::swap(Ztc, Ztn);
    ++k;
    return LRN_OK;
  };
  // scaled steps until the assumed interval is [1 - 5e-4, 1], then two plain ones: 3 (5e-4)^2 -> 1e-12
  while (k < maxit && 1.0 - ell > 5e-4) {
    const double a = std::sqrt(3.0 / (1.0 + ell + ell * ell));
    ell = 0.5 * a * ell * (3.0 - a * a * ell * ell);
    LRN_TRY(one_step(a));
  }
  for (int e = 0; e < 2 && k < maxit; ++e) LRN_TRY(one_step(1.0));
  std::vector<double> hres(maxit + 2, 0.0);
  bool ok = false;
  for (;;) {
    LRN_HIP(c, hipMemcpyAsync(hres.data(), res, (size_t)(maxit + 2) * 8, hipMemcpyDeviceToHost, st));
    LRN_HIP(c, hipStreamSynchronize(st));
    const double rl = hres[k - 1];           // ||I - P|| at the START of the last step: the step squares it
    if (!(rl == rl) || rl > 1e30) break;     // NaN / overflow: not a matrix this iteration handles
    if (rl <= 1e-6) { ok = true; break; }    // (the literal iteration is quadratic down to 1e-13: the last step leaves <= 1e-12)
    if (k + 1 > maxit) break;
    LRN_TRY(one_step(1.0));
    if (rl > 1e-2 && k + 1 <= maxit) LRN_TRY(one_step(1.0));
  }
  c->counts["ns_steps"] = k;
  toc(c, "prepw_ns");
  static const bool trace = getenv("LRN_NS_TRACE") != nullptr;
  if (trace) {
    fprintf(stderr, "[ns n=%d] ok=%d c=%.3e res:", n, (int)ok, hres[maxit]);
    for (int i = 0; i < k; ++i) fprintf(stderr, " %.2e", hres[i]);
    fprintf(stderr, "\n");
  }
  if (!ok) {
    if (two) LRN_HIP(c, hipStreamSynchronize(s2));
    c->counts["ns_fallback"] += 1;
    return LRN_OK;
  }
  tic(c);
  b.ns_c = hres[maxit];
  if (Yc != Y) LRN_HIP(c, hipMemcpyAsync(Y, Yc, mm, hipMemcpyDeviceToDevice, st));
  if (Zc != Z) LRN_HIP(c, hipMemcpyAsync(Z, Zc, mm, hipMemcpyDeviceToDevice, st));
  // W = L_X K^-1/2 L_X' = L_X Z L_X' / sqrt(c)                                          (prepare_W.jl:64)
  LRN_TRY(pgemm_nt(c, st, n, LX, Ztc, Pm, GEMM_KTO_M, 1.0));                             // (L_X lower triangular)
  LRN_TRY(pgemm_nt_sym(c, st, n, Pm, LX, b.W.as<double>(), 1.0 / std::sqrt(b.ns_c), GEMM_KTO_N));
  // (K/c)^-1 = Zh^2: the sigma_mu S^-1 term of the corrector in the L_X basis
  if (n >= 1500) LRN_TRY(pgemm_nt_sym(c, st, n, Zc, Ztc, b.Ki.as<double>(), 1.0));       // symmetric: lower tiles + mirror
  else LRN_TRY(pgemm_nt(c, st, n, Zc, Ztc, b.Ki.as<double>(), 0, 1.0));
  // Si = S^-1 = L_X K^-1 L_X' = L_X Ki L_X' / c  (prepare_W.jl:68): two products with the triangular factor, nothing inverted
  LRN_TRY(pgemm_nt(c, st, n, LX, b.Ki.as<double>(), Pm, GEMM_KTO_M, 1.0));               // L_X Ki'  (Ki symmetric)
  LRN_TRY(pgemm_nt_sym(c, st, n, Pm, LX, b.Si.as<double>(), 1.0 / b.ns_c, GEMM_KTO_N));
  if (two) {                                                  // join (the transposes on the second stream)
    LRN_HIP(c, hipEventRecord(c->evB, s2));
    LRN_HIP(c, hipStreamWaitEvent(st, c->evB, 0));
  }
  toc(c, "prepw_gemm");
  LRN_HIP(c, hipGetLastError());
  b.have_W = true;
  b.have_G = false;
  b.nt_free = true;
  c->scal_version += 1;
  *converged = true;
  return LRN_OK;
}

}  // namespace lrn

using namespace lrn;

extern "C" int lrn_prepare_w(lrn_ctx* c, int il, const double* X, const double* S, double* D, double* G,
                             double* Gi, double* W, double* Si, double* DDsi, int* info) {
  if (!c || il < 0 || il >= c->nlmi || !X || !S || !info) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LmiBlock& b = c->lmi[il];
  const size_t mm = (size_t)b.msz * b.msz * 8, mv = (size_t)b.msz * 8;
  LRN_TRY(copy_in(c, b.X.p, X, mm));
  LRN_TRY(copy_in(c, b.S.p, S, mm));
  hipEvent_t a0, a1;
  if (c->profile) { (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventRecord(a0, c->stream); }
  LRN_TRY(prepare_w_block(c, b, info));
  if (c->profile) {
    (void)hipEventRecord(a1, c->stream); (void)hipEventSynchronize(a1);
    float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
    c->timing["prepare_w"] += ms; c->counts["prepare_w"] += 1;
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
  }
  if (*info != 0) return LRN_OK;
  if (D) LRN_TRY(copy_out(c, D, b.D.p, mv));
  if (G) LRN_TRY(copy_out(c, G, b.G.p, mm));
  if (Gi) LRN_TRY(copy_out(c, Gi, b.Gi.p, mm));
  if (W) LRN_TRY(copy_out(c, W, b.W.p, mm));
  if (Si) LRN_TRY(copy_out(c, Si, b.Si.p, mm));
  if (DDsi) LRN_TRY(copy_out(c, DDsi, b.DDsi.p, mv));
  return LRN_OK;
}

extern "C" int lrn_dbg_svd_jacobi(lrn_ctx* c, int n, const double* A, double* U_sigma, double* V, double* sigma,
                                  int* sweeps) {
  if (!c || n <= 0 || !A || !U_sigma || !V || !sigma) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf dA, dV, dS;
  size_t mm = (size_t)n * n * 8;
  LRN_TRY(ensure(c, dA, mm));
  LRN_TRY(ensure(c, dV, mm));
  LRN_TRY(ensure(c, dS, (size_t)n * 8));
  LRN_TRY(copy_in(c, dA.p, A, mm));
  int sw = 0;
  LRN_TRY(jacobi_svd(c, dA.as<double>(), dV.as<double>(), dS.as<double>(), n, &sw));
  if (sweeps) *sweeps = sw;
  LRN_TRY(copy_out(c, U_sigma, dA.p, mm));
  LRN_TRY(copy_out(c, V, dV.p, mm));
  int rc = copy_out(c, sigma, dS.p, (size_t)n * 8);
  release(dA); release(dV); release(dS);
  return rc;
}
