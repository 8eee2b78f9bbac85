// Device-resident interior-point step: everything of the reference's per-iteration host work
// that touches msz x msz matrices:
//   residual Rd, right-hand sides (makeRHS, corrector my_kron term), find_step with its two
//   extreme-eigenvalue problems per block, the predictor point / RNT, the iterate update and
//   the matrix parts of find_mu / check_convergence.
// Reference: src/predictor_corrector.jl:8-16,186,248-326 ; src/Solvers.jl:480-511 ;
// src/kron_etc.jl.  nvar-/nlin-vectors and scalars stay with the host driver.
//
// eigmin (predictor_corrector.jl:272,285; Solvers.jl:503,505) is a Lanczos iteration on the
// device: y = M q on all CUs (bandwidth-bound symmetric mat-vec), one fused single-workgroup
// kernel per step for alpha, the three-term update and beta, no host sync inside a batch of
// steps; the host only bisects the tiny tridiagonal matrix.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/loraine_hip.h"
#include "ops.h"
#include "tridiag.h"

namespace lrn {

static inline unsigned nbk(long n, long cap = 4096) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// out = a*A + b*B + c*C (null pointers skipped), n elements
__global__ void lin3_kernel(double* __restrict__ out, double a, const double* __restrict__ A, double b,
                            const double* __restrict__ B, double c, const double* __restrict__ C, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    double v = 0.0;
    if (A) v += a * A[e];
    if (B) v += b * B[e];
    if (C) v += c * C[e];
    out[e] = v;
  }
}

// out = (M + M')/2 (out may alias M only through the symmetric access pattern -> use a separate out)
__global__ void sym_kernel(const double* __restrict__ M, double* __restrict__ out, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    out[e] = 0.5 * (M[e] + M[(long)j + (long)i * n]);
  }
}

// Q = sym( dd_j * M[i,j] * dd_i )   (predictor_corrector.jl:268-269)
__global__ void scale_sym_kernel(const double* __restrict__ M, const double* __restrict__ dd, double* __restrict__ out, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    out[e] = 0.5 * dd[i] * dd[j] * (M[e] + M[(long)j + (long)i * n]);
  }
}

// RNT = -(Mx + Mx') ./ (D_i + D_j)   (predictor_corrector.jl:308-309)
__global__ void rnt_kernel(const double* __restrict__ Mx, const double* __restrict__ D, double* __restrict__ out, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    out[e] = -(Mx[e] + Mx[(long)j + (long)i * n]) / (D[i] + D[j]);
  }
}

// inner = G'RdG + diag(D - sigma_mu/D) - RNT    (predictor_corrector.jl:186)
__global__ void corr_inner_kernel(const double* __restrict__ GRG, const double* __restrict__ D, const double* __restrict__ RNT,
                                  double sigma_mu, double* __restrict__ out, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    double v = GRG[e] - RNT[e];
    if (i == j) v += D[i] - sigma_mu / D[i];
    out[e] = v;
  }
}

// TX = -I - T (+ a Ki + R): L_X^-1 dX L_X^-T in the eigen-free scaling (a = sigma_mu / c, Ki = (K/c)^-1; R may be null)
__global__ void tx_kernel(const double* __restrict__ T, double a, const double* __restrict__ Ki, const double* __restrict__ R,
                          double* __restrict__ out, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    double v = -T[e];
    if ((int)(e % n) == (int)(e / n)) v -= 1.0;
    if (R) v += a * Ki[e] + R[e];
    out[e] = v;
  }
}

__global__ void add_diag_mat_kernel(double* __restrict__ M, int n, double eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) M[(long)i * n + i] += eps;
}

// two-stage reductions: partial[b] = sum A.*B (B may be null -> A.*A)
__global__ __launch_bounds__(256) void dot_part_kernel(const double* __restrict__ A, const double* __restrict__ B, long n,
                                                       double* __restrict__ part) {
  __shared__ double sh[4];
  double s = 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) s += A[e] * (B ? B[e] : A[e]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void dot_final_kernel(const double* __restrict__ part, int np, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int e = threadIdx.x; e < np; e += 256) s += part[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = sh[0] + sh[1] + sh[2] + sh[3];
}

static int mm(lrn_ctx* c, int n, const double* A, bool tA, const double* B, bool tB, double* C) {
  GemmDesc g;
  g.A = A; g.B = B; g.C = C;
  g.M = g.N = g.K = n;
  if (!tA) { g.sAm = 1; g.sAk = n; } else { g.sAm = n; g.sAk = 1; }
  if (!tB) { g.sBk = 1; g.sBn = n; } else { g.sBk = n; g.sBn = 1; }
  g.sCm = 1; g.sCn = n;
  return gemm(c->stream, g);
}

static int dot_dev(lrn_ctx* c, const double* A, const double* B, long n, double* out_dev) {
  const int np = (int)std::min<long>(1024, (n + 255) / 256);
  LRN_TRY(ensure(c, c->redbuf, (size_t)(np + 64) * 8));
  hipLaunchKernelGGL(dot_part_kernel, dim3(np), dim3(256), 0, c->stream, A, B, n, c->redbuf.as<double>());
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, c->stream, c->redbuf.as<double>(), np, out_dev);
  return LRN_OK;
}

// ------------------------------------------------------------------ Lanczos eigmin
__global__ void lz_init_kernel(double* __restrict__ q, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u + 12345u;     // fixed pseudo-random start vector
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  q[i] = ((double)h / 4294967296.0) - 0.5;
}

// ypart[chunk][i] = sum_{j in chunk} M[i + j*n] q[j]
__global__ __launch_bounds__(256) void symv_part_kernel(const double* __restrict__ M, int n, int cper,
                                                        const double* __restrict__ q, double* __restrict__ ypart) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int c0 = blockIdx.y * cper, c1 = min(n, c0 + cper);
  double s = 0.0;
  for (int j = c0; j < c1; ++j) s += M[(long)i + (long)j * n] * q[j];
  ypart[(long)blockIdx.y * n + i] = s;
}

__device__ __forceinline__ double wg_sum1024b(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += sh[i];
  return s;
}

// one Lanczos step (no re-orthogonalisation): w = sum ypart; a = q.w; w -= a q + b_prev q_prev;
// b = ||w||; q_next = w / b.   j == -1: just normalise q in place (start vector).
__global__ __launch_bounds__(1024) void lz_step_kernel(const double* __restrict__ ypart, int nchunk, int n, int j,
                                                       double* __restrict__ q, double* __restrict__ qprev,
                                                       double* __restrict__ w, double* __restrict__ ab) {
  __shared__ double sh[16];
  const int t = threadIdx.x;
  if (j < 0) {
    double s = 0.0;
    for (int i = t; i < n; i += 1024) s += q[i] * q[i];
    s = wg_sum1024b(s, sh);
    double r = 1.0 / sqrt(s);
    for (int i = t; i < n; i += 1024) { q[i] *= r; qprev[i] = 0.0; }
    return;
  }
  const double bprev = j > 0 ? ab[2 * (j - 1) + 1] : 0.0;
  double a = 0.0;
  for (int i = t; i < n; i += 1024) {
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += ypart[(long)k * n + i];
    w[i] = s;
    a += q[i] * s;
  }
  a = wg_sum1024b(a, sh);
  double b2 = 0.0;
  for (int i = t; i < n; i += 1024) {
    double v = w[i] - a * q[i] - bprev * qprev[i];
    w[i] = v;
    b2 += v * v;
  }
  b2 = wg_sum1024b(b2, sh);
  double b = sqrt(b2);
  double r = b > 0.0 ? 1.0 / b : 0.0;
  for (int i = t; i < n; i += 1024) {
    double qi = q[i];
    qprev[i] = qi;
    q[i] = w[i] * r;
  }
  if (t == 0) { ab[2 * j] = a; ab[2 * j + 1] = b; }
}

// One Lanczos step in ONE launch (n <= LZ_FUSED_MAX; round 3).  The two-kernel step above costs two dependent launches
// (10 + 7 us at msz 800, rocprofv3) for a few microseconds of work.  Here every workgroup first finishes step j-1 by
// itself -- alpha_{j-1} from the partial dots of the previous launch, w = y_{j-1} - alpha q_{j-1} - beta_{j-2} q_{j-2},
// beta_{j-1} = ||w|| and q_j = w / beta_{j-1} for ALL n entries, redundantly (n <= 4096 flops per thread block, in LDS) --
// and then computes its 16 rows of y_j = M q_j and their share of q_j . y_j.  Vectors rotate through three (q) and two (y)
// buffers so that nothing a workgroup still reads is overwritten inside a launch.  `do_symv` = 0: only finish step j-1
// (last launch of a batch: the host needs alpha, beta of every step it reads).
static constexpr int LZ_FUSED_MAX = 16384;      // (round 4: 4096 -> 16384, q_j in up to 128 KB of LDS: at msz 10^4 the two-kernel
                                                // step costs 0.21 + 0.24 ms -- its single-workgroup half sums 64 partial vectors)
__device__ __forceinline__ void lz_fused_body(const double* __restrict__ M, int n, int nwg, int j, int do_symv, int qmod,
                                              double* Q3, double* Y2, double* PA2, double* ab, double* qs, double* sh) {
  const int t = threadIdx.x;
  double* qj = Q3 + (size_t)(j % qmod) * n;       // qmod = 3: rotating buffers; > number of steps: every q_j is kept
  if (j == 0) {
    for (int i = t; i < n; i += 256) qs[i] = qj[i];
  } else {
    const double* qm1 = Q3 + (size_t)((j - 1) % qmod) * n;  // q_{j-1}
    const double* qm2 = Q3 + (size_t)((j > 1 ? j - 2 : 0) % qmod) * n;   // q_{j-2}
    const double* ym1 = Y2 + (size_t)((j + 1) & 1) * n;     // y_{j-1}
    const double* pa = PA2 + (size_t)((j + 1) & 1) * nwg;
    double a = 0.0;
    for (int e = t; e < nwg; e += 256) a += pa[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((t & 63) == 0) sh[t >> 6] = a;
    __syncthreads();
    const double alpha = sh[0] + sh[1] + sh[2] + sh[3];
    const double bprev = j > 1 ? ab[2 * (j - 2) + 1] : 0.0;
    __syncthreads();
    double b2 = 0.0;
    for (int i = t; i < n; i += 256) {
      const double v = ym1[i] - alpha * qm1[i] - (j > 1 ? bprev * qm2[i] : 0.0);
      qs[i] = v;
      b2 += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) b2 += __shfl_down(b2, off, 64);
    if ((t & 63) == 0) sh[t >> 6] = b2;
    __syncthreads();
    const double beta = sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
    const double r = beta > 0.0 ? 1.0 / beta : 0.0;
    for (int i = t; i < n; i += 256) {
      const double v = qs[i] * r;
      qs[i] = v;
      if (blockIdx.x == 0) qj[i] = v;
    }
    if (blockIdx.x == 0 && t == 0) { ab[2 * (j - 1)] = alpha; ab[2 * (j - 1) + 1] = beta; }
  }
  __syncthreads();
  if (!do_symv) return;
  // rows [16 blockIdx.x, +16) of M q = the same COLUMNS of the symmetric M (contiguous): wave w takes four of them, its
  // lanes run down the columns with the four loads of a step in flight together (round 4; round 3 walked the rows with a
  // stride of n and four loads in flight per thread: 50 dependent rounds of L2 latency at msz 800)
  {
    const int lane = t & 63, w = t >> 6;
    const int c0 = blockIdx.x * 16 + 4 * w;
    const double* m0 = M + (size_t)min(c0 + 0, n - 1) * n;
    const double* m1 = M + (size_t)min(c0 + 1, n - 1) * n;
    const double* m2 = M + (size_t)min(c0 + 2, n - 1) * n;
    const double* m3 = M + (size_t)min(c0 + 3, n - 1) * n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
    for (int k = lane; k < n; k += 64) {
      const double q = qs[k];
      a0 += m0[k] * q; a1 += m1[k] * q; a2 += m2[k] * q; a3 += m3[k] * q;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a0 += __shfl_down(a0, off, 64); a1 += __shfl_down(a1, off, 64);
      a2 += __shfl_down(a2, off, 64); a3 += __shfl_down(a3, off, 64);
    }
    if (lane == 0) { sh[4 * w + 0] = a0; sh[4 * w + 1] = a1; sh[4 * w + 2] = a2; sh[4 * w + 3] = a3; }
  }
  __syncthreads();
  if (t < 16) {
    const double y = sh[t];
    const int ii = blockIdx.x * 16 + t;
    double d = 0.0;
    if (ii < n) {
      Y2[(size_t)(j & 1) * n + ii] = y;
      d = qs[ii] * y;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) d += __shfl_down(d, off, 16);
    if (t == 0) PA2[(size_t)(j & 1) * nwg + blockIdx.x] = d;
  }
}

__global__ __launch_bounds__(256) void lz_fused_kernel(const double* __restrict__ M, int n, int nwg, int j, int do_symv, int qmod,
                                                       double* Q3, double* Y2, double* PA2, double* ab) {
  extern __shared__ double qs[];            // q_j (n doubles)
  __shared__ double sh[16 * 16 + 8];
  lz_fused_body(M, n, nwg, j, do_symv, qmod, Q3, Y2, PA2, ab, qs, sh);
}

// The same step of TWO independent runs on matrices of one size in one launch: blockIdx.y picks the run (round 4, second
// session).  The two eigmin searches of a step-length computation used to live on two streams so that their launch chains
// overlap.  Under rocprofv3's kernel trace they do not (tools/lz_overlap.py on a maxG11 solve: 4 % of the kernel time of the
// two queues overlaps); without the profiler both forms take the same time (same box, profiles/r04_lanczos_pair_ab.txt:
// find_step 4.99 vs 4.96 ms at 480 steps, 0.84 vs 0.90 at 64) -- the chains did overlap, and what a step-length search costs
// is its LONGER chain at 8-10 us per step.  The paired form is the default all the same: half the launches for the host to
// issue, one stream, no events between streams.
struct LzPair {
  const double* M[2];
  double* Q3[2];
  double* Y2[2];
  double* PA2[2];
  double* ab[2];
};
__global__ __launch_bounds__(256) void lz_fused_pair_kernel(LzPair a, int n, int nwg, int j, int do_symv, int qmod) {
  extern __shared__ double qs[];            // q_j (n doubles)
  __shared__ double sh[16 * 16 + 8];
  const int r = blockIdx.y;                 // (uniform: the arrays of the argument block are read with scalar loads)
  lz_fused_body(a.M[r], n, nwg, j, do_symv, qmod, a.Q3[r], a.Y2[r], a.PA2[r], a.ab[r], qs, sh);
}

// more than 64 KB of dynamic LDS need the attribute (once per device); false: the two-kernel step is taken
static bool lz_big_lds_ok() {
  static bool done[64] = {}, ok[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!done[dev]) {
    ok[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(lz_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LZ_FUSED_MAX * 8) == hipSuccess &&
              hipFuncSetAttribute(reinterpret_cast<const void*>(lz_fused_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LZ_FUSED_MAX * 8) == hipSuccess;
    if (!ok[dev]) (void)hipGetLastError();
    done[dev] = true;
  }
  return ok[dev];
}

// ---- the same steps [j0, j1) in ONE launch (round 4; MEASURED SLOWER, kept behind LRN_LZ_PERSIST=1 for the record): a batch
// of Lanczos steps is a chain of launches of 7-8 us each for 2-3 us of work.  Here the nwg <= 256 workgroups stay
// resident and meet at a barrier after every step: a monotonic counter in global memory (release fence, one atomic add per
// workgroup, acquire fence; every workgroup executes the same number of barriers).  The kernel can NOT hang: a workgroup
// that waits longer than `limit` ticks of the 100 MHz wall clock (its peers were not scheduled -- a GPU shared with another
// process, an over-subscribed chip) raises flag[1], every workgroup leaves at its next barrier, and the host redoes the run
// with one launch per step (lz_fetch).  flag[0]: the counter, flag[1]: abort.
__device__ __forceinline__ bool lz_grid_barrier(unsigned* flag, unsigned target, long long limit, int* ok_s) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(flag, 1u);
    int ok = 1;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (wall_clock64() - t0 > limit) {
        __hip_atomic_store(flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
    *ok_s = ok;
  }
  __syncthreads();
  return *ok_s != 0;
}

__global__ __launch_bounds__(256) void lz_fused_multi_kernel(const double* __restrict__ M, int n, int nwg, int j0, int j1,
                                                             int qmod, double* Q3, double* Y2, double* PA2, double* ab,
                                                             unsigned* flag, unsigned base, long long limit) {
  extern __shared__ double qs[];
  __shared__ double sh[16 * 16 + 8];
  __shared__ int ok_s;
  unsigned target = base;
  for (int j = j0; j < j1; ++j) {
    lz_fused_body(M, n, nwg, j, 1, qmod, Q3, Y2, PA2, ab, qs, sh);
    target += (unsigned)nwg;
    if (!lz_grid_barrier(flag, target, limit, &ok_s)) return;
  }
  if (blockIdx.x == 0) lz_fused_body(M, n, nwg, j1, 0, qmod, Q3, Y2, PA2, ab, qs, sh);      // finish step j1 - 1
}

// ---- RESIDENT steps, second form (round 4, second session; option lz_resident).  What made the kernel above slower than
// one launch per step is not the barrier but its two device-scope fences (tools/lab/xcd_barrier.hip,
// profiles/r04_xcd_barrier.txt: a barrier of 51 workgroups with an 801-vector exchanged costs 4.7 us per step with
// __threadfence() on both sides and 2.3 us when counter AND payload travel as relaxed agent-scope atomics -- they bypass the
// L1s and meet at the coherent level, nothing has to be written back or invalidated), and that every step still re-read its
// 16 columns of M and three vectors from L2.  Here, for n <= 1024:
//  * a workgroup keeps its 16 columns of M in REGISTERS for the whole launch (wave w: columns 4 w .. 4 w + 3, lane l: rows
//    l, l + 64, ...: the order in which lz_fused_body sums them) and q_j, q_{j-1}, q_{j-2} in LDS;
//  * the only data other workgroups produce -- the 16 entries of y_j and the partial sum of q_j . y_j per workgroup -- are
//    written and read with relaxed agent-scope atomic stores / loads; beta_{j-1} stays in a register;
//  * there is no barrier at all: a word that has not been written yet holds a mark, and a reader polls the 17 words of every
//    workgroup until no mark is left (LZ_MARK_BITS below); every poll loop is bounded by the wall clock and an abort word.
// The arithmetic, operation by operation, is lz_fused_body's: same coefficients bit for bit
// (test_resident_lanczos_steps_are_the_launched_ones).  blockIdx.y: the run (two runs of a step-length search in lock-step).
struct LzRes {
  const double* M;
  double* Q3;
  double* Y3;            // three n-vectors: y_j in buffer j % 3
  double* PA3;           // three nwg-vectors: the workgroups' shares of q_j . y_j
  double* ab;
  unsigned* flag;        // flag[1]: abort word
};
struct LzResPair { LzRes r[2]; };

__device__ __forceinline__ double lz_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lz_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// "Not written yet": a quiet NaN with a payload no computation produces.  The exchange needs NO counter: a workgroup's
// 17 words of step j go to buffer j % 3, which it has filled with this value at step j - 1 -- at a time when every
// workgroup was done reading the buffer's previous content (that of step j - 3: read in the first phase of step j - 2, and a
// workgroup publishes its words of step j - 2 only after that phase; seeing all of those is what lets step j - 1 begin) --
// and whose reset it has seen acknowledged before it published step j - 1 (s_waitcnt vmcnt(0) between the two).  A reader
// therefore finds either the mark or the word of step j, never an older word, and polls until no mark is left.
static constexpr unsigned long long LZ_MARK_BITS = 0x7ff8a5a5deadbeefULL;
__device__ __forceinline__ bool lz_is_mark(double v) { return (unsigned long long)__double_as_longlong(v) == LZ_MARK_BITS; }

__global__ void lz_mark_kernel(double* __restrict__ p, int cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt) p[i] = __longlong_as_double((long long)LZ_MARK_BITS);
}

static constexpr int LZ_RES_MAX = 1024;       // 16 rows per lane and column
__global__ __launch_bounds__(256) void lz_resident_kernel(LzResPair args, int n, int nwg, int j0, int j1, int qmod, long long limit) {
  extern __shared__ double ql[];            // three n-vectors: q_j, q_{j-1}, q_{j-2} rotate through them
  __shared__ double sh[16 * 16 + 8];
  __shared__ int ok_s;
  const LzRes& R = args.r[blockIdx.y];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int c0 = blockIdx.x * 16 + 4 * w;
  const double mark = __longlong_as_double((long long)LZ_MARK_BITS);
  constexpr int U = LZ_RES_MAX / 64;
  double mreg[4][U];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const double* col = R.M + (size_t)min(c0 + cc, n - 1) * n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = lane + 64 * u;
      mreg[cc][u] = k < n ? col[k] : 0.0;
    }
  }
  // state of the recurrence at entry: q_{j0-1}, q_{j0-2} (the ring of the previous launches), beta_{j0-2}
  double bprev = 0.0;
  if (j0 > 0) {
    const double* g1 = R.Q3 + (size_t)((j0 - 1) % qmod) * n;
    double* l1 = ql + (size_t)((j0 - 1) % 3) * n;
    for (int i = t; i < n; i += 256) l1[i] = g1[i];
    if (j0 > 1) {
      const double* g2 = R.Q3 + (size_t)((j0 - 2) % qmod) * n;
      double* l2 = ql + (size_t)((j0 - 2) % 3) * n;
      for (int i = t; i < n; i += 256) l2[i] = g2[i];
      bprev = R.ab[2 * (j0 - 2) + 1];
    }
  }
  if (t == 0) ok_s = 1;
  __syncthreads();
  double alpha_out = 0.0;
  for (int j = j0; j <= j1; ++j) {
    const bool finish = j == j1;            // (workgroup 0 only: alpha, beta of step j1 - 1 and q_{j1} for the host and the next launch)
    if (finish && blockIdx.x != 0) break;
    double* qs = ql + (size_t)(j % 3) * n;
    double* qj = R.Q3 + (size_t)(j % qmod) * n;
    bool polled = true;
    if (j == 0) {
      for (int i = t; i < n; i += 256) qs[i] = qj[i];
    } else {
      const double* qm1 = ql + (size_t)((j - 1) % 3) * n;
      const double* qm2 = ql + (size_t)((j > 1 ? j - 2 : 0) % 3) * n;
      const double* ym1 = R.Y3 + (size_t)((j - 1) % 3) * n;
      const double* pa = R.PA3 + (size_t)((j - 1) % 3) * nwg;
      // the words of step j - 1 of every workgroup: polled until none is the mark (all requests of a poll in flight together)
      double yv[LZ_RES_MAX / 256];
      double a = 0.0;
      long long t0 = 0;
      for (int tries = 0;; ++tries) {
        int bad = 0;
#pragma unroll
        for (int u = 0; u < LZ_RES_MAX / 256; ++u) {
          const int i = t + 256 * u;
          yv[u] = i < n ? lz_ld(ym1 + i) : 0.0;
        }
        a = t < nwg ? lz_ld(pa + t) : 0.0;            // (nwg <= 64)
#pragma unroll
        for (int u = 0; u < LZ_RES_MAX / 256; ++u) bad |= lz_is_mark(yv[u]) ? 1 : 0;
        bad |= lz_is_mark(a) ? 1 : 0;
        if (tries > 0 && t == 0) {                    // bounded: the wall clock, and the other workgroups' verdict
          if (tries == 1) t0 = wall_clock64();
          if (__hip_atomic_load(R.flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ok_s = 0;
          else if (wall_clock64() - t0 > limit) {
            __hip_atomic_store(R.flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok_s = 0;
          }
        }
        if (!__syncthreads_or(bad)) break;
        if (!ok_s) { polled = false; break; }
      }
      if (!polled) return;
      // this workgroup's words of buffer (j + 1) % 3 back to the mark (see LZ_MARK_BITS): issued first thing -- everybody has
      // published step j - 1, so nobody reads that buffer's old content any more --, acknowledged by the time step j is published
      if (!finish && t < 16) {
        const int ii = blockIdx.x * 16 + t;
        if (ii < n) lz_st(R.Y3 + (size_t)((j + 1) % 3) * n + ii, mark);
        if (t == 0) lz_st(R.PA3 + (size_t)((j + 1) % 3) * nwg + blockIdx.x, mark);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
      if ((t & 63) == 0) sh[t >> 6] = a;
      __syncthreads();
      const double alpha = sh[0] + sh[1] + sh[2] + sh[3];
      __syncthreads();
      double b2 = 0.0;
#pragma unroll
      for (int u = 0; u < LZ_RES_MAX / 256; ++u) {
        const int i = t + 256 * u;
        if (i < n) {
          const double v = yv[u] - alpha * qm1[i] - (j > 1 ? bprev * qm2[i] : 0.0);
          qs[i] = v;
          b2 += v * v;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) b2 += __shfl_down(b2, off, 64);
      if ((t & 63) == 0) sh[t >> 6] = b2;
      __syncthreads();
      const double beta = sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
      const double r = beta > 0.0 ? 1.0 / beta : 0.0;
      for (int i = t; i < n; i += 256) qs[i] *= r;
      alpha_out = alpha;
      bprev = beta;
    }
    if (j == 0 && !finish && t < 16) {      // (step 0: the marks of buffer 1; lz_resident_prepare has set them already, kept for symmetry)
      const int ii = blockIdx.x * 16 + t;
      if (ii < n) lz_st(R.Y3 + (size_t)n + ii, mark);
      if (t == 0) lz_st(R.PA3 + (size_t)nwg + blockIdx.x, mark);
    }
    __syncthreads();
    if (finish) {      // (workgroup 0)
      if (j > 0) {
        for (int i = t; i < n; i += 256) qj[i] = qs[i];
        if (t == 0) { R.ab[2 * (j - 1)] = alpha_out; R.ab[2 * (j - 1) + 1] = bprev; }
      }
      break;
    }
    {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = lane + 64 * u;
        if (k < n) {
          const double q = qs[k];
          a0 += mreg[0][u] * q; a1 += mreg[1][u] * q; a2 += mreg[2][u] * q; a3 += mreg[3][u] * q;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off, 64); a1 += __shfl_down(a1, off, 64);
        a2 += __shfl_down(a2, off, 64); a3 += __shfl_down(a3, off, 64);
      }
      if (lane == 0) { sh[4 * w + 0] = a0; sh[4 * w + 1] = a1; sh[4 * w + 2] = a2; sh[4 * w + 3] = a3; }
    }
    __syncthreads();
    if (t < 16) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the marks above have been acknowledged
      const double y = sh[t];
      const int ii = blockIdx.x * 16 + t;
      double d = 0.0;
      if (ii < n) {
        lz_st(R.Y3 + (size_t)(j % 3) * n + ii, y);
        d = qs[ii] * y;
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) d += __shfl_down(d, off, 16);
      if (t == 0) lz_st(R.PA3 + (size_t)(j % 3) * nwg + blockIdx.x, d);
    }
    // q_j, alpha_{j-1}, beta_{j-1} for the host and the next launch: plain stores of workgroup 0, BEHIND the publication
    // (the wait for the marks' acknowledgement would otherwise wait for them as well, every step, with everybody waiting)
    if (blockIdx.x == 0 && j > 0) {
      for (int i = t; i < n; i += 256) qj[i] = qs[i];
      if (t == 0) { R.ab[2 * (j - 1)] = alpha_out; R.ab[2 * (j - 1) + 1] = bprev; }
    }
    __syncthreads();      // (sh and the q ring are rewritten by the next step)
  }
}

// steps [j0, j1) of the single-launch Lanczos recurrence with every q_j kept (Q: (j1 + 1) x n doubles, q_0 = unit start
// vector in Q[0..n)), then the finishing launch: alpha_j, beta_j of all steps < j1 are in ab, q_{j1} in Q.  For
// lanczos.hip (preconditioner setup); n <= LZ_FUSED_MAX.
int lz_fused_steps(hipStream_t st, const double* M, int n, int j0, int j1, int qcap, double* Q, double* Y2, double* PA2,
                   double* ab) {
  if (n > LZ_FUSED_LIMIT || j1 + 1 > qcap) return LRN_ERR_ARG;
  const int nwg = (n + 15) / 16;
  const size_t lds = (size_t)n * 8;
  for (int j = j0; j < j1; ++j)
    hipLaunchKernelGGL(lz_fused_kernel, dim3(nwg), dim3(256), lds, st, M, n, nwg, j, 1, qcap, Q, Y2, PA2, ab);
  hipLaunchKernelGGL(lz_fused_kernel, dim3(1), dim3(256), lds, st, M, n, nwg, j1, 0, qcap, Q, Y2, PA2, ab);
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// the same steps as ONE resident launch (lz_resident_kernel; n <= LZ_RES_MAX): Y3 / PA3 = three n- / nwg-vectors and flag =
// two words of the caller's workspace (flag[1]: abort word), all prepared by lz_resident_prepare before the run's first step
bool lz_resident_ok(const lrn_ctx* c, int n) {
  return c->opt.lz_resident != 0 && !c->lz_no_persist && n <= LZ_RES_MAX && n >= 32;
}
// before the first step of a run: the three exchange buffers hold the mark, the abort word is clear
void lz_resident_prepare(hipStream_t st, int n, double* Y3, double* PA3, unsigned* flag) {
  const int nwg = (n + 15) / 16;
  hipLaunchKernelGGL(lz_mark_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, st, Y3, 3 * n);
  hipLaunchKernelGGL(lz_mark_kernel, dim3((3 * nwg + 255) / 256), dim3(256), 0, st, PA3, 3 * nwg);
  (void)hipMemsetAsync(flag, 0, 16, st);
}
int lz_resident_steps(hipStream_t st, const double* M, int n, int j0, int j1, int qcap, double* Q, double* Y3, double* PA3,
                      double* ab, unsigned* flag) {
  if (n > LZ_RES_MAX || j1 + 1 > qcap || qcap < 3) return LRN_ERR_ARG;
  const int nwg = (n + 15) / 16;
  LzResPair a;
  a.r[0] = LzRes{M, Q, Y3, PA3, ab, flag};
  a.r[1] = a.r[0];
  hipLaunchKernelGGL(lz_resident_kernel, dim3(nwg, 1), dim3(256), (size_t)3 * n * 8, st, a, n, nwg, j0, j1, qcap, 2000000LL);
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// (the eigenvalues of the tridiagonal matrices: tridiag.h -- bisection on a division-free Sturm count, bracket from the
// previous batch's value)

// |beta_m * s_m| for the Ritz pair (theta, s) of T_m: the residual norm ||M v - theta v|| of the
// Ritz vector, a rigorous bound on the distance from theta to the spectrum.  s by two steps of
// inverse iteration on the tridiagonal matrix (Thomas algorithm with a tiny shift).
static double ritz_residual(const std::vector<double>& a, const std::vector<double>& b, int m, double theta) {
  if (m <= 1) return 0.0;
  std::vector<double> s(m, 1.0 / std::sqrt((double)m)), d(m), u(m), y(m);
  double scale = 0.0;
  for (int i = 0; i < m; ++i) scale = std::max(scale, std::fabs(a[i]) + (i < m - 1 ? std::fabs(b[i]) : 0.0));
  const double shift = theta - 1e-13 * std::max(scale, 1e-300) - 1e-300;
  for (int it = 0; it < 3; ++it) {
    // solve (T - shift I) y = s  (T - shift I is positive definite up to rounding)
    d[0] = a[0] - shift;
    if (d[0] == 0.0) d[0] = 1e-300;
    u[0] = s[0];
    for (int i = 1; i < m; ++i) {
      double l = b[i - 1] / d[i - 1];
      d[i] = a[i] - shift - l * b[i - 1];
      if (d[i] == 0.0) d[i] = 1e-300;
      u[i] = s[i] - l * u[i - 1];
    }
    y[m - 1] = u[m - 1] / d[m - 1];
    for (int i = m - 2; i >= 0; --i) y[i] = (u[i] - b[i] * y[i + 1]) / d[i];
    double nrm = 0.0;
    for (int i = 0; i < m; ++i) nrm += y[i] * y[i];
    nrm = std::sqrt(nrm);
    if (!(nrm > 0.0) || !std::isfinite(nrm)) return std::fabs(b[m - 1]);
    for (int i = 0; i < m; ++i) s[i] = y[i] / nrm;
  }
  return std::fabs(b[m - 1] * s[m - 1]);
}

// One Lanczos iteration as a resumable run: batches of 16 steps are queued on the run's stream, the (alpha, beta)
// pairs come back in one copy per batch and the host decides on the tridiagonal matrix.  The two eigmin calls of a
// step-length search run side by side: in one launch per pair of steps (eigmin_dev_pair_merged), or, above LZ_FUSED_MAX,
// as two launch chains on two streams (eigmin_dev_pair).
struct LzRun {
  const double* M = nullptr;
  int n = 0;
  hipStream_t st = nullptr;
  double *q = nullptr, *qprev = nullptr, *w = nullptr, *ypart = nullptr, *ab = nullptr;
  int nchunk = 1, cper = 1, mmax = 0, m = 0, m1 = 0;
  std::vector<double> a, b, hab;
  double theta = 0.0, theta_prev = 0.0, scale = 0.0;
  double last_move = 0.0;                   // |theta - theta_prev| of the last collect: how far the next one is expected to move
  int mc = 0;                               // steps whose coefficients the host holds (lz_fetch); a batch [mc, m1) may be queued ahead
  bool ahead = false;
  double err_prev = 0.0, err_last = 0.0;    // residual / (1e-3 x its scale) at the last two looks (0: none): lz_queue_ahead
  bool have_prev = false, conv = false, done = false;
  bool fused = false;                       // lz_fused_kernel: Q3 = q (3 n), Y2 = w (2 n), PA2 = ypart (2 nwg)
  int nwg = 0;
  bool persist = false;                     // lz_fused_multi_kernel: a batch of steps per launch
  bool resident = false;                    // lz_resident_kernel: the same, M in registers, relaxed-atomic exchange (option lz_resident)
  double *y3 = nullptr, *pa3 = nullptr;     // its exchange buffers (3 n, 3 nwg doubles)
  unsigned* flag = nullptr;                 // its barrier counter and abort word
  unsigned bar_base = 0;                    // barriers passed so far x nwg
};

static int lz_begin(lrn_ctx* c, LzRun& r, const double* M, int n, hipStream_t st, DBuf& buf) {
  r.M = M; r.n = n; r.st = st;
  // without re-orthogonalisation the extreme Ritz value may need more than n steps
  r.mmax = std::min(1500, 4 * n + 40);
  // column chunks of the symmetric mat-vec: ~64 columns per thread keeps the step kernel's reduction short
  r.nchunk = std::max(1, std::min(64, n / 64));
  r.cper = (n + r.nchunk - 1) / r.nchunk;
  r.nchunk = (n + r.cper - 1) / r.cper;
  static const bool no_fused = getenv("LRN_LZ_UNFUSED") != nullptr;
  r.fused = !no_fused && n <= LZ_FUSED_MAX && n >= 32 && ((size_t)n * 8 <= 60 * 1024 || lz_big_lds_ok());
  r.nwg = (n + 15) / 16;
  LRN_TRY(ensure(c, buf, ((size_t)5 * n + (size_t)std::max(r.nchunk * n, 2 * r.nwg) + 2 * (size_t)r.mmax + 64 + 3 * (size_t)n + 3 * (size_t)r.nwg) * 8));
  r.q = buf.as<double>();                 // fused: Q3 = q[0..3n)
  r.qprev = r.q + n;
  r.w = r.q + 3 * (size_t)n;              // fused: Y2 = w[0..2n)
  r.ypart = r.w + 2 * (size_t)n;          // fused: PA2
  r.ab = r.ypart + (size_t)std::max(r.nchunk * n, 2 * r.nwg);
  // (measurement knob, off: with two runs interleaved on two streams the per-step launches are hidden already and the
  // barrier -- device-scope release / acquire across eight L2s -- costs more than a launch: maxG11 find_step 1.0 -> 1.25 ms)
  static const bool persist_on = getenv("LRN_LZ_PERSIST") && atoi(getenv("LRN_LZ_PERSIST")) != 0;
  r.persist = r.fused && persist_on && !c->lz_no_persist && r.nwg <= 256 && (size_t)n * 8 <= 60 * 1024;
  r.resident = r.fused && !r.persist && c->opt.lz_resident != 0 && !c->lz_no_persist && n <= LZ_RES_MAX;
  r.flag = reinterpret_cast<unsigned*>(r.ab + 2 * (size_t)r.mmax + 8);      // (inside the 64 doubles of slack)
  r.y3 = r.ab + 2 * (size_t)r.mmax + 64;
  r.pa3 = r.y3 + 3 * (size_t)n;
  r.bar_base = 0;
  return LRN_OK;
}

// start vector (after lz_begin; a fresh workspace is zeroed on c->stream, which r.st must have waited for)
static void lz_start(LzRun& r) {
  if (r.persist) (void)hipMemsetAsync(r.flag, 0, 16, r.st);
  if (r.resident) lz_resident_prepare(r.st, r.n, r.y3, r.pa3, r.flag);
  r.bar_base = 0;
  hipLaunchKernelGGL(lz_init_kernel, dim3((r.n + 255) / 256), dim3(256), 0, r.st, r.q, r.n);
  hipLaunchKernelGGL(lz_step_kernel, dim3(1), dim3(1024), 0, r.st, r.ypart, r.nchunk, r.n, -1, r.q, r.qprev, r.w, r.ab);
}

static void lz_launch(LzRun& r) {
  const int batch = r.n <= 16 ? r.n : 16;
  r.m1 = std::min(r.mmax, r.m + batch);
  if (r.resident) {
    LzResPair a;
    a.r[0] = LzRes{r.M, r.q, r.y3, r.pa3, r.ab, r.flag};
    a.r[1] = a.r[0];
    hipLaunchKernelGGL(lz_resident_kernel, dim3(r.nwg, 1), dim3(256), (size_t)3 * r.n * 8, r.st, a, r.n, r.nwg, r.m, r.m1, 3, 2000000LL);
    return;
  }
  if (r.fused && r.persist) {
    hipLaunchKernelGGL(lz_fused_multi_kernel, dim3(r.nwg), dim3(256), (size_t)r.n * 8, r.st, r.M, r.n, r.nwg, r.m, r.m1, 3, r.q,
                       r.w, r.ypart, r.ab, r.flag, r.bar_base, 2000000LL);          // limit: 20 ms at 100 MHz
    r.bar_base += (unsigned)(r.m1 - r.m) * (unsigned)r.nwg;
    return;
  }
  if (r.fused) {
    const size_t lds = (size_t)r.n * 8;
    for (int j = r.m; j < r.m1; ++j)
      hipLaunchKernelGGL(lz_fused_kernel, dim3(r.nwg), dim3(256), lds, r.st, r.M, r.n, r.nwg, j, 1, 3, r.q, r.w, r.ypart, r.ab);
    hipLaunchKernelGGL(lz_fused_kernel, dim3(1), dim3(256), lds, r.st, r.M, r.n, r.nwg, r.m1, 0, 3, r.q, r.w, r.ypart, r.ab);
    return;
  }
  for (int j = r.m; j < r.m1; ++j) {
    hipLaunchKernelGGL(symv_part_kernel, dim3((r.n + 255) / 256, r.nchunk), dim3(256), 0, r.st, r.M, r.n, r.cper, r.q, r.ypart);
    hipLaunchKernelGGL(lz_step_kernel, dim3(1), dim3(1024), 0, r.st, r.ypart, r.nchunk, r.n, j, r.q, r.qprev, r.w, r.ab);
  }
}

// waits for the batch in flight and brings its coefficients: r.mc steps are on the host afterwards
static int lz_fetch(lrn_ctx* c, LzRun& r) {
  const int m1 = r.m1;
  r.hab.resize(2 * (size_t)m1);
  LRN_HIP(c, hipMemcpyAsync(r.hab.data(), r.ab, (size_t)2 * m1 * 8, hipMemcpyDeviceToHost, r.st));
  unsigned fl[2] = {0u, 0u};
  if (r.persist || r.resident) LRN_HIP(c, hipMemcpyAsync(fl, r.flag, 8, hipMemcpyDeviceToHost, r.st));
  LRN_HIP(c, hipStreamSynchronize(r.st));
  if ((r.persist || r.resident) && fl[1] != 0u) {
    // the resident workgroups did not all meet in time (see lz_fused_multi_kernel): from now on one launch per step on
    // this context, and this run again from its start vector
    c->lz_no_persist = true;
    c->counts["lz_persist_abort"] += 1;
    r.persist = false;
    r.resident = false;
    r.m = 0; r.have_prev = false; r.scale = 0.0;
    r.err_prev = r.err_last = 0.0;
    lz_start(r);
    lz_launch(r);
    return lz_fetch(c, r);
  }
  r.mc = m1;
  r.m = m1;
  r.ahead = false;
  return LRN_OK;
}

// A run that is alone on the GPU (its partner of eigmin_dev_pair has ended, or eigmin_dev) leaves the stream empty while
// the host looks at T: a synchronisation, a bisection, an inverse iteration and the first launch of the next batch, ~50 us
// per 117 us batch (maxG11: 7.3 us per step in runs of 16-30 steps, 12 in runs of 120).  Between lz_fetch and lz_decide:
// when the last two looks say that the coming one cannot end the run -- the residual, extrapolated geometrically, stays
// above 1e-2 of its scale, out of reach of every rule of lz_decide (the Kato-Temple rule needs 1e-3, the plain one 1e-11;
// the sign-class rule is excluded by theta < 0) -- the next batch is queued before that look.  Timing only: the looks and
// their verdicts are the same; a batch queued in vain is ignored (eigmin_dev_pair makes c->stream wait for it).
static bool lz_cannot_end_at_next_look(const LzRun& r) {
  static const bool off = getenv("LRN_LZ_NOAHEAD") != nullptr;      // (measurement knob)
  if (off || r.persist || r.ahead || r.mc >= r.mmax || !(r.err_prev > 0.0) || !(r.err_last > 0.0)) return false;
  if (!(r.theta_prev < 0.0)) return false;
  const double next = r.err_last * std::min(1.0, r.err_last / r.err_prev);
  return next > 10.0;
}

static void lz_queue_ahead(lrn_ctx* c, LzRun& r) {
  if (!lz_cannot_end_at_next_look(r)) return;
  lz_launch(r);                       // (r.m == r.mc: the batch [mc, m1))
  r.ahead = true;
  c->counts["lanczos_ahead"] += 1;
}

// the look at T of the fetched steps: r.done when converged, settled or out of steps
static int lz_decide(lrn_ctx* c, LzRun& r) {
  const int m1 = r.mc;
  r.a.resize(m1); r.b.resize(m1);
  int mm_ = m1;
  for (int j = 0; j < m1; ++j) {
    r.a[j] = r.hab[2 * j]; r.b[j] = r.hab[2 * j + 1];
    r.scale = std::max(r.scale, std::fabs(r.a[j]) + std::fabs(r.b[j]));
    if (!(r.b[j] > 1e-14 * r.scale) && j + 1 < m1) { mm_ = j + 1; break; }      // invariant subspace
  }
  // T of the previous batch is a leading block of this one: its smallest eigenvalue bounds this one from above
  r.theta = tri_eig_kth(r.a, r.b, mm_, 0, r.have_prev ? &r.theta_prev : nullptr, r.last_move);
  if (mm_ < m1) { r.conv = true; r.done = true; return LRN_OK; }
  // stop on the rigorous residual bound; for a clearly non-negative spectrum (theta > 0 is an
  // upper bound of lambda_min) the callers only need the sign class once theta has settled
  const double res = ritz_residual(r.a, r.b, mm_, r.theta);
  r.err_prev = r.err_last;
  r.err_last = res / (1e-3 * std::max(std::max(std::fabs(r.theta), 1e-4 * r.scale), 1e-300));
  if (res <= 1e-11 * std::max(std::fabs(r.theta), 1e-4 * r.scale)) { r.conv = true; r.done = true; return LRN_OK; }
  // Kato-Temple: theta - lambda_min <= res^2 / (lambda_2 - theta).  lambda_2 is bounded below through the second Ritz
  // pair (an eigenvalue lies within res2 of theta2; if that eigenvalue is lambda_min itself -- a ghost copy -- the gap
  // below is <= 0 and the rule does not fire).  The step-length rule consumes lambda_min to ~1e-10 relative.
  static const double kt_tol = getenv("LRN_EIGMIN_KT") ? atof(getenv("LRN_EIGMIN_KT")) : 1e-10;
  if (kt_tol > 0.0 && mm_ >= 8 && r.theta <= -1e-6 && res <= 1e-3 * std::max(std::fabs(r.theta), 1e-4 * r.scale)) {
    const double th2 = tri_eig_kth(r.a, r.b, mm_, 1);
    const double res2 = ritz_residual(r.a, r.b, mm_, th2);
    const double gap = (th2 - res2) - r.theta;
    if (gap > 0.0 && res < 0.25 * gap && res * res / gap <= kt_tol * std::fabs(r.theta)) {
      r.conv = true; r.done = true;
      return LRN_OK;
    }
  }
  if (r.have_prev && r.theta > 0.0 && std::fabs(r.theta - r.theta_prev) <= 1e-3 * r.theta && r.mc >= 64) { r.done = true; return LRN_OK; }
  r.last_move = r.have_prev ? std::fabs(r.theta - r.theta_prev) : 0.0;
  r.theta_prev = r.theta;
  r.have_prev = true;
  if (r.mc >= r.mmax) r.done = true;
  return LRN_OK;
}

// Both ends of the spectrum of a symmetric matrix from `nsteps` plain Lanczos steps: lo = smallest Ritz value (an UPPER
// bound of lambda_min), hi = largest Ritz value, res_hi = residual norm of its Ritz pair (an eigenvalue lies within res_hi
// of hi).  For the scaling of the Newton-Schulz iteration (prepw.hip), where a wrong value costs steps, not correctness.
int lanczos_ends(lrn_ctx* c, const double* M, int n, int nsteps, double* lo, double* hi, double* res_hi) {
  LzRun r;
  LRN_TRY(lz_begin(c, r, M, n, c->stream, c->lzbuf));
  r.mmax = std::min(r.mmax, std::max(4, nsteps));
  lz_start(r);
  std::vector<double> hab;
  while (r.m < r.mmax) {
    lz_launch(r);
    r.m = r.m1;
  }
  const int m = r.m;
  hab.resize(2 * (size_t)m);
  LRN_TRY(copy_out(c, hab.data(), r.ab, (size_t)2 * m * 8));
  std::vector<double> a(m), b(m), an(m);
  int mm_ = m;
  double scale = 0.0;
  for (int j = 0; j < m; ++j) {
    a[j] = hab[2 * j]; b[j] = hab[2 * j + 1]; an[j] = -a[j];
    scale = std::max(scale, std::fabs(a[j]) + std::fabs(b[j]));
    if (!(b[j] > 1e-14 * scale) && j + 1 < m) { mm_ = j + 1; break; }      // invariant subspace: the Ritz values are exact
  }
  if (!(scale == scale) || mm_ < 1) return set_error(c, LRN_ERR_STATE, "lanczos_ends: not a finite matrix");
  *lo = tri_eig_kth(a, b, mm_, 0);
  const double top = -tri_eig_kth(an, b, mm_, 0);    // largest eigenvalue of T = - smallest of -T (same off-diagonal)
  *hi = top;
  *res_hi = mm_ < m ? 0.0 : ritz_residual(an, b, mm_, -top);
  c->counts["lanczos_ends_steps"] += m;
  return LRN_OK;
}

int eigmin_dev(lrn_ctx* c, const double* M, int n, double* lam, int* steps_out, bool* converged = nullptr,
               double* scale_out = nullptr) {
  if (converged) *converged = true;
  if (scale_out) *scale_out = 0.0;
  if (n == 1) {
    LRN_TRY(copy_out(c, lam, M, 8));
    if (steps_out) *steps_out = 1;
    return LRN_OK;
  }
  LzRun r;
  LRN_TRY(lz_begin(c, r, M, n, c->stream, c->lzbuf));
  lz_start(r);
  while (!r.done) {
    if (!r.ahead) lz_launch(r);
    LRN_TRY(lz_fetch(c, r));
    lz_queue_ahead(c, r);                    // (alone on the GPU: see there)
    LRN_TRY(lz_decide(c, r));
  }
  *lam = r.theta;
  c->counts["lanczos_steps"] += r.m;
  c->counts["lanczos_runs"] += 1;
  if (steps_out) *steps_out = r.m;
  if (converged) *converged = r.conv;
  if (scale_out) *scale_out = r.scale;
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// ---- two runs in lock-step, one launch per pair of steps (lz_fused_pair_kernel), both on c->stream
static void lz_launch_pair(LzRun* r) {
  const int n = r[0].n;
  const int batch = n <= 16 ? n : 16;
  const int m0 = r[0].m, m1 = std::min(r[0].mmax, m0 + batch);
  LzPair a;
  for (int k = 0; k < 2; ++k) { a.M[k] = r[k].M; a.Q3[k] = r[k].q; a.Y2[k] = r[k].w; a.PA2[k] = r[k].ypart; a.ab[k] = r[k].ab; }
  if (r[0].resident && r[1].resident) {
    LzResPair ra;
    for (int k = 0; k < 2; ++k) ra.r[k] = LzRes{r[k].M, r[k].q, r[k].y3, r[k].pa3, r[k].ab, r[k].flag};
    hipLaunchKernelGGL(lz_resident_kernel, dim3(r[0].nwg, 2), dim3(256), (size_t)3 * n * 8, r[0].st, ra, n, r[0].nwg, m0, m1, 3, 2000000LL);
    r[0].m1 = r[1].m1 = m1;
    return;
  }
  const size_t lds = (size_t)n * 8;
  for (int j = m0; j < m1; ++j)
    hipLaunchKernelGGL(lz_fused_pair_kernel, dim3(r[0].nwg, 2), dim3(256), lds, r[0].st, a, n, r[0].nwg, j, 1, 3);
  hipLaunchKernelGGL(lz_fused_pair_kernel, dim3(1, 2), dim3(256), lds, r[0].st, a, n, r[0].nwg, m1, 0, 3);
  r[0].m1 = r[1].m1 = m1;
}

static int lz_fetch_pair(lrn_ctx* c, LzRun* r, bool* aborted) {
  const int m1 = r[0].m1;
  unsigned fl[2][2] = {{0u, 0u}, {0u, 0u}};
  for (int k = 0; k < 2; ++k) {
    r[k].hab.resize(2 * (size_t)m1);
    LRN_HIP(c, hipMemcpyAsync(r[k].hab.data(), r[k].ab, (size_t)2 * m1 * 8, hipMemcpyDeviceToHost, r[0].st));
    if (r[k].resident) LRN_HIP(c, hipMemcpyAsync(fl[k], r[k].flag, 8, hipMemcpyDeviceToHost, r[0].st));
  }
  LRN_HIP(c, hipStreamSynchronize(r[0].st));
  *aborted = fl[0][1] != 0u || fl[1][1] != 0u;      // a resident launch gave up at a barrier: the caller starts over, launched
  if (*aborted) return LRN_OK;
  for (int k = 0; k < 2; ++k) { r[k].mc = m1; r[k].m = m1; r[k].ahead = false; }
  return LRN_OK;
}

static int eigmin_dev_pair_merged(lrn_ctx* c, const double* M1, const double* M2, int n, double lam[2], bool conv[2],
                                  double scale[2], bool* taken) {
  LzRun r[2];
  LRN_TRY(lz_begin(c, r[0], M1, n, c->stream, c->lzbuf));
  LRN_TRY(lz_begin(c, r[1], M2, n, c->stream, c->lzbuf2));
  *taken = r[0].fused && r[1].fused && !r[0].persist && !r[1].persist && r[0].mmax == r[1].mmax;
  if (!*taken) return LRN_OK;                // (n > LZ_FUSED_MAX or < 32: the two-stream form below)
  lz_start(r[0]);
  lz_start(r[1]);
  while (!r[0].done && !r[1].done) {
    if (!r[0].ahead) lz_launch_pair(r);
    bool aborted = false;
    LRN_TRY(lz_fetch_pair(c, r, &aborted));
    if (aborted) {
      c->lz_no_persist = true;               // (lz_begin: no resident launches on this context from now on)
      c->counts["lz_persist_abort"] += 1;
      return eigmin_dev_pair_merged(c, M1, M2, n, lam, conv, scale, taken);
    }
    if (lz_cannot_end_at_next_look(r[0]) && lz_cannot_end_at_next_look(r[1])) {
      lz_launch_pair(r);                     // (queued before the host looks at the batch it has just fetched: lz_queue_ahead)
      r[0].ahead = r[1].ahead = true;
      c->counts["lanczos_ahead"] += 1;
    }
    LRN_TRY(lz_decide(c, r[0]));
    LRN_TRY(lz_decide(c, r[1]));
    c->counts["lanczos_pair_batches"] += 1;
    if (r[0].resident && r[1].resident) c->counts["lanczos_resident_batches"] += 1;
  }
  // the longer run goes on alone (a pair batch queued ahead carries its steps [mc, m1) already)
  for (int k = 0; k < 2; ++k) {
    while (!r[k].done) {
      if (!r[k].ahead) lz_launch(r[k]);
      LRN_TRY(lz_fetch(c, r[k]));
      lz_queue_ahead(c, r[k]);
      LRN_TRY(lz_decide(c, r[k]));
    }
  }
  for (int k = 0; k < 2; ++k) {
    lam[k] = r[k].theta; conv[k] = r[k].conv; scale[k] = r[k].scale;
    c->counts["lanczos_steps"] += r[k].m;
    c->counts["lanczos_runs"] += 1;
  }
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// The Lanczos runs of two matrices of the same size side by side (second one on c->stream2, which first waits for
// everything queued on c->stream): results as from two eigmin_dev calls.  Option eigmin_pair = 2 (default): both runs in one
// launch per step instead (eigmin_dev_pair_merged); 1: this two-stream form.
static int eigmin_dev_pair(lrn_ctx* c, const double* M1, const double* M2, int n, double lam[2], bool conv[2],
                           double scale[2]) {
  if (c->opt.eigmin_pair >= 2) {
    bool taken = false;
    LRN_TRY(eigmin_dev_pair_merged(c, M1, M2, n, lam, conv, scale, &taken));
    if (taken) return LRN_OK;
  }
  if (!c->stream2) LRN_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  LzRun r[2];
  LRN_TRY(lz_begin(c, r[0], M1, n, c->stream, c->lzbuf));
  LRN_TRY(lz_begin(c, r[1], M2, n, c->stream2, c->lzbuf2));
  LRN_HIP(c, hipEventRecord(c->ev1, c->stream));             // the matrices, and the zeroing of a fresh workspace
  LRN_HIP(c, hipStreamWaitEvent(c->stream2, c->ev1, 0));
  lz_start(r[0]);
  lz_start(r[1]);
  lz_launch(r[0]);
  lz_launch(r[1]);
  while (!r[0].done || !r[1].done) {
    for (int k = 0; k < 2; ++k) {
      if (r[k].done) continue;
      LRN_TRY(lz_fetch(c, r[k]));
      if (r[1 - k].done) lz_queue_ahead(c, r[k]);      // (otherwise the other run's batch keeps the GPU busy meanwhile)
      LRN_TRY(lz_decide(c, r[k]));
      if (!r[k].done && !r[k].ahead) lz_launch(r[k]);
    }
  }
  if (r[1].ahead) {                          // a batch queued in vain on stream2 still reads M2 and its workspace
    LRN_HIP(c, hipEventRecord(c->ev1, c->stream2));
    LRN_HIP(c, hipStreamWaitEvent(c->stream, c->ev1, 0));
  }
  for (int k = 0; k < 2; ++k) {
    lam[k] = r[k].theta; conv[k] = r[k].conv; scale[k] = r[k].scale;
    c->counts["lanczos_steps"] += r[k].m;
    c->counts["lanczos_runs"] += 1;
  }
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// ---- certified smallest eigenvalue
// A Ritz value is only an UPPER bound of lambda_min, and plain Lanczos resolves the spectrum relative to
// its spread: lambda_min = -1 next to eigenvalues of 1e6..1e10 (a poor direction after a regularised
// Schur solve) comes back as -0.94 or even +9 -- the step-length rule (predictor_corrector.jl:272-291)
// would then leave the cone.  Every estimate is therefore certified by one Cholesky test
//   M - (theta - delta) I  positive definite  <=>  lambda_min > theta - delta,
// and if the test fails lambda_min is bracketed by bisection over such tests (the reference calls a
// dense `eigmin`; a positive-definiteness test is its GEMM-rich equivalent on this hardware).
static int chol_shift_is_pd(lrn_ctx* c, const double* M, int n, double shift, bool* pd) {
  hipStream_t st = c->stream;
  LRN_TRY(ensure(c, c->info_dev, 64));
  const size_t nn = (size_t)n * n;
  LRN_TRY(ensure(c, c->ezbuf, (nn + (size_t)n * CHOL_NB + chol_linv_doubles(n) + 64) * 8));
  double* F = c->ezbuf.as<double>();
  double* work = F + nn;
  double* linv = work + (size_t)n * CHOL_NB;
  int* info = c->info_dev.as<int>() + 8;
  LRN_HIP(c, hipMemcpyAsync(F, M, nn * 8, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(add_diag_mat_kernel, dim3((n + 255) / 256), dim3(256), 0, st, F, n, shift);
  LRN_HIP(c, hipMemsetAsync(info, 0, 4, st));
  LRN_TRY(potrf_lower(st, F, n, n, linv, work, info));
  int h = 0;
  LRN_TRY(copy_out(c, &h, info, 4));
  *pd = h == 0;
  c->counts["eigmin_chol_tests"] += 1;
  return LRN_OK;
}

static int eigmin_certify(lrn_ctx* c, const double* M, int n, double theta, bool conv, double scale, double* lam);

int eigmin_certified(lrn_ctx* c, const double* M, int n, double* lam) {
  double theta = 0.0, scale = 0.0;
  bool conv = false;
  LRN_TRY(eigmin_dev(c, M, n, &theta, nullptr, &conv, &scale));
  return eigmin_certify(c, M, n, theta, conv, scale, lam);
}

// eigmin_certified of two matrices of the same size: the two Lanczos runs interleaved, then the certificates
int eigmin_certified_pair(lrn_ctx* c, const double* M1, const double* M2, int n, double* lam1, double* lam2) {
  if (n == 1 || !c->opt.eigmin_pair) {
    LRN_TRY(eigmin_certified(c, M1, n, lam1));
    return eigmin_certified(c, M2, n, lam2);
  }
  double th[2], sc[2];
  bool cv[2];
  LRN_TRY(eigmin_dev_pair(c, M1, M2, n, th, cv, sc));
  LRN_TRY(eigmin_certify(c, M1, n, th[0], cv[0], sc[0], lam1));
  return eigmin_certify(c, M2, n, th[1], cv[1], sc[1], lam2);
}

static int eigmin_certify(lrn_ctx* c, const double* M, int n, double theta, bool conv, double scale, double* lam) {
  static const bool trace = getenv("LRN_EIGMIN_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[eigmin n=%d] theta=%.12g conv=%d scale=%.3g\n", n, theta, (int)conv, scale);
  if (n == 1) { *lam = theta; return LRN_OK; }
  // a Ritz value converged to 1e-11 on a spectrum of moderate spread (the usual O(1) scaled directions)
  // needs no certificate: the failures are unconverged runs on spectra spanning 1e6 and more
  if (conv && theta <= -1e-6 && scale <= 1e3 * std::fabs(theta)) { *lam = theta; return LRN_OK; }
  bool pd = false;
  if (theta > -1e-6) {
    // callers only use the class "lambda_min > -1e-6" (step 0.99, DIMACS err2/err4 = 0)
    LRN_TRY(chol_shift_is_pd(c, M, n, 1e-6, &pd));
    if (pd) { *lam = theta; return LRN_OK; }
  } else {
    const double delta = 1e-7 * std::fabs(theta);
    LRN_TRY(chol_shift_is_pd(c, M, n, delta - theta, &pd));
    if (pd) { *lam = theta; return LRN_OK; }
  }
  // the Ritz value was not converged: bracket lambda_min in (lo, hi], hi = theta is an upper bound
  c->counts["eigmin_bisections"] += 1;
  double hi = theta, beta = std::max(2.0 * std::fabs(theta), 1.0);
  for (int it = 0; it < 200; ++it) {
    LRN_TRY(chol_shift_is_pd(c, M, n, beta, &pd));
    if (pd) break;
    hi = std::min(hi, -beta);
    beta *= 4.0;
  }
  if (!pd) return set_error(c, LRN_ERR_STATE, "eigmin: matrix has no finite lower bound (NaN/Inf entries?)");
  double lo = -beta;
  for (int it = 0; it < 100 && hi - lo > 1e-9 * std::max(std::fabs(lo), 1e-6); ++it) {
    const double mid = 0.5 * (lo + hi);
    LRN_TRY(chol_shift_is_pd(c, M, n, -mid, &pd));
    if (trace) fprintf(stderr, "   bisect mid=%.12g pd=%d\n", mid, (int)pd);
    if (pd) lo = mid; else hi = mid;
  }
  *lam = lo;          // the safe side: slightly too negative shortens the step
  return LRN_OK;
}

// ------------------------------------------------------------------ Lyapunov solve (eigen-free NT scaling)
// out = scale (T + T') by 32 x 32 tiles (both reads coalesced); with `dotp`: part[block] = sum dotp .* out over the block's
// tiles.  out is exactly symmetric: (i,j) and (j,i) add the same two numbers.
__global__ __launch_bounds__(256) void symadd_kernel(SlabSrc T, int n, double scale, double* __restrict__ out,
                                                     const double* __restrict__ dotp, double* __restrict__ part) {
  __shared__ double ta[32][33], tb[32][33];
  __shared__ double sh[4];
  const int nt = (n + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  double acc = 0.0;
  for (long t = blockIdx.x; t < (long)nt * nt; t += gridDim.x) {
    const int bi = (int)(t % nt) * 32, bj = (int)(t / nt) * 32;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi + tx, j = bj + r;       // tile (bi, bj): element (i, j)
      ta[r][tx] = (i < n && j < n) ? slab_sum(T, (long)i + (long)j * n) : 0.0;
      const int i2 = bj + tx, j2 = bi + r;     // tile (bj, bi): element (i2, j2)
      tb[r][tx] = (i2 < n && j2 < n) ? slab_sum(T, (long)i2 + (long)j2 * n) : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi + tx, j = bj + r;
      if (i < n && j < n) {
        const double v = scale * (ta[r][tx] + tb[tx][r]);      // T[i,j] + T[j,i]
        out[(long)i + (long)j * n] = v;
        if (dotp) acc += dotp[(long)i + (long)j * n] * v;
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ double block_sum_parts(const double* __restrict__ part, int np, double* sh) {
  double s = 0.0;
  for (int e = threadIdx.x; e < np; e += 256) s += part[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// out = (M + M') / 2: the tiled kernel above from msz 512 on (the element-wise one reads M' with stride n)
// C = alpha A Bm' for a consumer that can add split-K slabs while it reads: below msz 1500 on one rank the product may come
// back as slabs (src->n > 1, C untouched); otherwise it is in C
static int prod_slabs(lrn_ctx* c, hipStream_t st, int n, const double* A, const double* Bm, double* C, double alpha, int tri,
                      SlabSrc* src) {
  if (n >= 1500 || products_sharded(c, st, n)) {
    src->p = C; src->stride = 0; src->n = 1;
    return pgemm_nt(c, st, n, A, Bm, C, tri, alpha);
  }
  return gemm_nt_slabs(st, n, A, Bm, C, alpha, src);
}

static void sym_half(hipStream_t st, const double* M, double* out, int n) {
  if (n >= 512) {
    const long nt = (n + 31) / 32;
    hipLaunchKernelGGL(symadd_kernel, dim3((unsigned)std::min<long>(1024, nt * nt)), dim3(256), 0, st, SlabSrc{M, 0, 1}, n, 0.5, out,
                       (const double*)nullptr, (double*)nullptr);
  } else {
    hipLaunchKernelGGL(sym_kernel, dim3(nbk((long)n * n)), dim3(256), 0, st, M, out, n);
  }
}

// alpha = rr_k / <p, Ap> (every workgroup sums the same partials in the same order); R += alpha p; r -= alpha Ap;
// part2[block] = sum r^2
__global__ __launch_bounds__(256) void lyap_xr_kernel(const double* __restrict__ part1, int np1, const double* __restrict__ hist,
                                                      int k, const double* __restrict__ p, const double* __restrict__ Ap,
                                                      double* __restrict__ R, double* __restrict__ r, long total,
                                                      double* __restrict__ part2) {
  __shared__ double sh[4];
  const double pAp = block_sum_parts(part1, np1, sh);
  const double rr = hist[k];
  const double alpha = (pAp > 0.0 && rr > 0.0) ? rr / pAp : 0.0;
  double s = 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    R[e] += alpha * p[e];
    const double v = r[e] - alpha * Ap[e];
    r[e] = v;
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part2[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// beta = rr_{k+1} / rr_k; p = r + beta p; hist[k+1] = rr_{k+1}
__global__ __launch_bounds__(256) void lyap_p_kernel(const double* __restrict__ part2, int np2, double* __restrict__ hist, int k,
                                                     const double* __restrict__ r, double* __restrict__ p, long total) {
  __shared__ double sh[4];
  const double rn = block_sum_parts(part2, np2, sh);
  const double rr = hist[k];
  const double beta = rr > 0.0 ? rn / rr : 0.0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) p[e] = r[e] + beta * p[e];
  if (blockIdx.x == 0 && threadIdx.x == 0) hist[k + 1] = rn;
}

// R with Yh R + R Yh = Cm (Yh symmetric positive definite, Cm symmetric) by conjugate gradients in the Frobenius inner
// product: one product Yh p per step (the other half of the operator is its transpose), all scalars stay on the device,
// the host reads the residual history once per batch of steps.  Cm is used as the residual and destroyed.
//
// Round 4 -- the SAME solution from a better conditioned equation (option lyap_form = 1, default).  Zh = Yh^-1 is at hand
// (the coupled Newton-Schulz iteration produces both), and multiplying  Yh R + R Yh = C  by Zh from both sides gives
// Zh R + R Zh = Zh C Zh.  Any positive combination is again a Lyapunov equation for the same R:
//     M R + R M = C / s + s Zh C Zh,      M = Yh / s + s Zh,
// whose coefficient has the spectrum y / s + s / y: with s^2 = tr(Yh) / tr(Zh) (inside [y_min^2, y_max^2], free) its
// condition number is ~ sqrt(cond(Yh)) / 2 and never above cond(Yh) / 2.  Still one product per CG step, two more for the
// right-hand side; NumPy on spectra of cond(K) 1e2 / 1e3 / 1e4: 42 / 77 / 137 steps -> 14 / 21 / 29, same accuracy
// (C5: 32-48 steps of a 10^4-cube product each, 1.1-1.6 s of a 2.8 s iteration).
__global__ __launch_bounds__(256) void lyap_trace2_kernel(const double* __restrict__ Y, const double* __restrict__ Z, int n,
                                                          double* __restrict__ sc) {
  __shared__ double sh[8];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { a += Y[(size_t)i * n + i]; b += Z[(size_t)i * n + i]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ty = sh[0] + sh[1] + sh[2] + sh[3], tz = sh[4] + sh[5] + sh[6] + sh[7];
    double s = (ty > 0.0 && tz > 0.0) ? sqrt(ty / tz) : 1.0;
    if (!(s > 0.0) || isinf(s)) s = 1.0;
    sc[0] = s;
    sc[1] = 1.0 / s;
  }
}

// M = Y / s + s Z
__global__ void lyap_mop_kernel(const double* __restrict__ Y, const double* __restrict__ Z, const double* __restrict__ sc,
                                long total, double* __restrict__ Mop) {
  const double s = sc[0], si = sc[1];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    Mop[e] = Y[e] * si + s * Z[e];
}

// C <- C / s + s (T + T') / 2   (T = Zh C Zh up to rounding asymmetry), by 32 x 32 tiles as symadd_kernel
__global__ __launch_bounds__(256) void lyap_rhs_kernel(double* __restrict__ Cm, const double* __restrict__ T, int n,
                                                       const double* __restrict__ sc) {
  __shared__ double ta[32][33], tb[32][33];
  const double s = sc[0], si = sc[1];
  const int nt = (n + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (long t = blockIdx.x; t < (long)nt * nt; t += gridDim.x) {
    const int bi = (int)(t % nt) * 32, bj = (int)(t / nt) * 32;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi + tx, j = bj + r;
      ta[r][tx] = (i < n && j < n) ? T[(long)i + (long)j * n] : 0.0;
      const int i2 = bj + tx, j2 = bi + r;
      tb[r][tx] = (i2 < n && j2 < n) ? T[(long)i2 + (long)j2 * n] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi + tx, j = bj + r;
      if (i < n && j < n) {
        const long e = (long)i + (long)j * n;
        Cm[e] = Cm[e] * si + s * 0.5 * (ta[r][tx] + tb[tx][r]);
      }
    }
  }
}

static int lyap_solve(lrn_ctx* c, LmiBlock& b, double* Cm, double* R, double* work, bool* ok, int* steps) {
  const int n = b.msz;
  const long nn = (long)n * n;
  hipStream_t st = c->stream;
  const int maxit = std::max(2, c->opt.lyap_maxit);
  const int np = (int)std::min<long>(1024, (nn + 255) / 256);
  const bool combined = c->opt.lyap_form != 0;
  LRN_TRY(ensure(c, b.lyap, ((size_t)(combined ? 3 : 2) * nn + 2 * 1024 + maxit + 16) * 8));
  double* p = b.lyap.as<double>();
  double* Ap = p + nn;
  double* part1 = Ap + nn;
  double* part2 = part1 + 1024;
  double* hist = part2 + 1024;
  double* Mop = hist + maxit + 16;
  double* r = Cm;
  const double* Cop = b.Yh.as<double>();             // coefficient matrix of the equation that is iterated on
  const int ntile0 = (n + 31) / 32;
  if (combined) {
    double* sc = part2;                               // (two doubles, free until the first lyap_xr_kernel)
    hipLaunchKernelGGL(lyap_trace2_kernel, dim3(1), dim3(256), 0, st, b.Yh.as<double>(), b.Zh.as<double>(), n, sc);
    hipLaunchKernelGGL(lyap_mop_kernel, dim3(np), dim3(256), 0, st, b.Yh.as<double>(), b.Zh.as<double>(), sc, nn, Mop);
    LRN_TRY(pgemm_nt(c, st, n, b.Zh.as<double>(), Cm, work));                 // Zh C   (C symmetric)
    LRN_TRY(pgemm_nt(c, st, n, work, b.Zh.as<double>(), Ap));                 // Zh C Zh
    hipLaunchKernelGGL(lyap_rhs_kernel, dim3((unsigned)std::min<long>(1024, (long)ntile0 * ntile0)), dim3(256), 0, st, Cm, Ap, n, sc);
    Cop = Mop;
  }
  LRN_HIP(c, hipMemsetAsync(R, 0, (size_t)nn * 8, st));
  LRN_HIP(c, hipMemcpyAsync(p, r, (size_t)nn * 8, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(dot_part_kernel, dim3(np), dim3(256), 0, st, r, (const double*)nullptr, nn, part1);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, part1, np, hist);
  std::vector<double> h(maxit + 1, 0.0);
  const double tol2 = c->opt.lyap_tol * c->opt.lyap_tol;
  int k = 0;
  *ok = false;
  const int ntile = (n + 31) / 32;
  const int np1 = (int)std::min<long>(1024, (long)ntile * ntile);
  int batch = std::min(maxit, n >= 2000 ? 6 : 8);      // first look at the residual history
  while (k < maxit) {
    const int k1 = std::min(maxit, k + batch);
    for (; k < k1; ++k) {
      SlabSrc prod;                                  // (mid sizes: the slabs of the product are added by symadd_kernel)
      LRN_TRY(prod_slabs(c, st, n, Cop, p, work, 1.0, 0, &prod));
      hipLaunchKernelGGL(symadd_kernel, dim3(np1), dim3(256), 0, st, prod, n, 1.0, Ap, p, part1);
      hipLaunchKernelGGL(lyap_xr_kernel, dim3(np), dim3(256), 0, st, part1, np1, hist, k, p, Ap, R, r, nn, part2);
      hipLaunchKernelGGL(lyap_p_kernel, dim3(np), dim3(256), 0, st, part2, np, hist, k, r, p, nn);
    }
    LRN_HIP(c, hipMemcpyAsync(h.data(), hist, (size_t)(k + 1) * 8, hipMemcpyDeviceToHost, st));
    LRN_HIP(c, hipStreamSynchronize(st));
    if (!(h[k] == h[k])) break;                                   // NaN
    if (h[0] == 0.0 || h[k] <= tol2 * h[0]) { *ok = true; break; }
    // how many more steps at the rate of the last two (CG only gets faster): queue that many before the next look --
    // a step is one n^3 product (31 ms at msz 10^4: none to waste), a look is a host round trip (20 us: too many at 800)
    batch = n >= 2000 ? 2 : 4;
    if (k >= 2 && h[k] > 0.0 && h[k] < h[k - 2]) {
      const double f = 0.5 * std::log(h[k - 2] / h[k]);           // decrement of log ||r||^2 per step
      const double m = std::log(h[k] / (tol2 * h[0])) / f;
      const int lo_b = n >= 2000 ? 1 : 2;
      batch = std::max(lo_b, std::min(8, (int)std::floor(0.9 * m + 0.5)));
    }
  }
  if (steps) *steps = k;
  return LRN_OK;
}

// ------------------------------------------------------------------ resident step
static int ensure_resident(lrn_ctx* c, LmiBlock& b) {
  size_t mm_ = (size_t)b.msz * b.msz * 8;
  for (DBuf* d : {&b.Cd, &b.Rd, &b.delX, &b.delS, &b.Xn, &b.Sn, &b.RNT, &b.t0, &b.t1, &b.t2})
    if (d->bytes < mm_) LRN_TRY(ensure(c, *d, mm_, true));
  b.resident = true;
  return LRN_OK;
}

}  // namespace lrn

using namespace lrn;

#define BLK(il)                                                        \
  if (!c || (il) < 0 || (il) >= c->nlmi) return LRN_ERR_ARG;          \
  LRN_HIP(c, hipSetDevice(c->device));                                 \
  LmiBlock& b = c->lmi[(il)];                                          \
  LRN_TRY(ensure_resident(c, b));                                      \
  const size_t mm_ = (size_t)b.msz * b.msz * 8;                        \
  (void)mm_;

extern "C" int lrn_ip_set_c(lrn_ctx* c, int il, const double* C) {
  BLK(il);
  if (!C) return LRN_ERR_ARG;
  return copy_in(c, b.Cd.p, C, mm_);
}

extern "C" int lrn_ip_set_iterate(lrn_ctx* c, int il, const double* X, const double* S) {
  BLK(il);
  if (!X || !S) return LRN_ERR_ARG;
  LRN_TRY(copy_in(c, b.X.p, X, mm_));
  LRN_TRY(copy_in(c, b.S.p, S, mm_));
  LRN_HIP(c, hipMemsetAsync(b.RNT.p, 0, mm_, c->stream));
  b.have_Vprev = false;
  b.chol_valid = false;
  return LRN_OK;
}

extern "C" int lrn_ip_get_iterate(lrn_ctx* c, int il, double* X, double* S) {
  BLK(il);
  if (X) LRN_TRY(copy_out(c, X, b.X.p, mm_));
  if (S) LRN_TRY(copy_out(c, S, b.S.p, mm_));
  return LRN_OK;
}

extern "C" int lrn_ip_add_diag(lrn_ctx* c, int il, int which, double eps) {
  BLK(il);
  if (which != 1 && which != 2) return LRN_ERR_ARG;
  b.chol_valid = false;
  hipLaunchKernelGGL(add_diag_mat_kernel, dim3((b.msz + 255) / 256), dim3(256), 0, c->stream,
                     (which == 1 ? b.X : b.S).as<double>(), b.msz, eps);
  return LRN_OK;
}

extern "C" int lrn_ip_prepare_w(lrn_ctx* c, int il, int* info) {
  BLK(il);
  if (!info) return LRN_ERR_ARG;
  hipEvent_t a0, a1;
  if (c->profile) { (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventRecord(a0, c->stream); }
  bool conv = false;
  if (c->opt.nt_mode == 1 && b.msz > 1) LRN_TRY(prepare_w_ns(c, b, info, &conv));
  if (!conv && *info == 0) {
    b.nt_free = false;
    LRN_TRY(prepare_w_block(c, b, info));
  }
  if (c->profile) {
    (void)hipEventRecord(a1, c->stream); (void)hipEventSynchronize(a1);
    float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
    c->timing["prepare_w"] += ms; c->counts["prepare_w"] += 1;
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
  }
  return LRN_OK;
}

extern "C" int lrn_ip_aa_x(lrn_ctx* c, double* out) {
  if (!c || !out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_HIP(c, hipMemsetAsync(c->v1.p, 0, (size_t)n * 8, c->stream));
  for (auto& b : c->lmi) LRN_TRY(aa_times(c, b, b.X.as<double>(), c->v1.as<double>()));
  return copy_out(c, out, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_ip_residual_d(lrn_ctx* c, const double* y) {
  if (!c || !y) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LRN_TRY(copy_in(c, c->v0.p, y, (size_t)c->nvar * 8));
  tic(c);
  for (auto& b : c->lmi) {
    LRN_TRY(ensure_resident(c, b));
    const long mm_ = (long)b.msz * b.msz;
    LRN_TRY(aat_to_mat(c, b, c->v0.as<double>(), b.t0.as<double>()));
    hipLaunchKernelGGL(lin3_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.Rd.as<double>(), 1.0, b.Cd.as<double>(),
                       -1.0, b.S.as<double>(), -1.0, b.t0.as<double>(), mm_);
  }
  toc(c, "residual_d");
  return LRN_OK;
}

extern "C" int lrn_ip_rhs_pred(lrn_ctx* c, double* out) {
  if (!c || !out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_HIP(c, hipMemsetAsync(c->v1.p, 0, (size_t)n * 8, c->stream));
  for (auto& b : c->lmi) {
    LRN_TRY(ensure_resident(c, b));
    const long mm_ = (long)b.msz * b.msz;
    hipLaunchKernelGGL(lin3_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t0.as<double>(), 1.0, b.Rd.as<double>(),
                       1.0, b.S.as<double>(), 0.0, (const double*)nullptr, mm_);
    LRN_TRY(wmw(c, b, b.t0.as<double>(), b.t1.as<double>(), b.t2.as<double>()));
    LRN_TRY(aa_times(c, b, b.t2.as<double>(), c->v1.as<double>()));
  }
  return copy_out(c, out, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_ip_rhs_pred2(lrn_ctx* c, double* aax_out, double* out) {
  if (!c || !aax_out || !out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_TRY(ensure(c, c->v2, (size_t)(n + 64) * 8));
  tic(c);
  LRN_HIP(c, hipMemsetAsync(c->v1.p, 0, (size_t)n * 8, c->stream));
  LRN_HIP(c, hipMemsetAsync(c->v2.p, 0, (size_t)n * 8, c->stream));
  for (auto& b : c->lmi) {
    LRN_TRY(ensure_resident(c, b));
    const long mm_ = (long)b.msz * b.msz;
    hipLaunchKernelGGL(lin3_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t0.as<double>(), 1.0, b.Rd.as<double>(),
                       1.0, b.S.as<double>(), 0.0, (const double*)nullptr, mm_);
    if (wmw_pattern_ok(c, b)) {          // every constraint sparse: W M W only where AA reads it (one product instead of two)
      LRN_TRY(aa_times(c, b, b.X.as<double>(), c->v2.as<double>()));
      LRN_TRY(aa_times_wmw_pattern(c, b, b.t0.as<double>(), b.t1.as<double>(), c->v1.as<double>()));
      continue;
    }
    LRN_TRY(wmw(c, b, b.t0.as<double>(), b.t1.as<double>(), b.t2.as<double>()));
    LRN_TRY(aa_times2(c, b, b.X.as<double>(), c->v2.as<double>(), b.t2.as<double>(), c->v1.as<double>()));
  }
  toc(c, "rhs");
  LRN_TRY(copy_out(c, aax_out, c->v2.p, (size_t)n * 8));
  return copy_out(c, out, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_ip_rhs_corr(lrn_ctx* c, double sigma_mu, double* out) {
  if (!c || !out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  tic(c);
  LRN_HIP(c, hipMemsetAsync(c->v1.p, 0, (size_t)n * 8, c->stream));
  for (auto& b : c->lmi) {
    LRN_TRY(ensure_resident(c, b));
    const int m = b.msz;
    const long mm_ = (long)m * m;
    if (b.nt_free) {
      // G (G'RdG + D - sigma_mu/D - RNT) G' = W Rd W + X - sigma_mu Si - G RNT G'
      if (wmw_pattern_ok(c, b)) {        // AA vec(W Rd W) on the pattern, AA vec(X - sigma_mu Si - G RNT G') by itself
        LRN_TRY(aa_times_wmw_pattern(c, b, b.Rd.as<double>(), b.t1.as<double>(), c->v1.as<double>()));
        hipLaunchKernelGGL(lin3_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t0.as<double>(), 1.0, b.X.as<double>(), -sigma_mu,
                           b.Si.as<double>(), -1.0, b.Qm.as<double>(), mm_);
        LRN_TRY(aa_times(c, b, b.t0.as<double>(), c->v1.as<double>()));
        continue;
      }
      LRN_TRY(wmw(c, b, b.Rd.as<double>(), b.t1.as<double>(), b.t2.as<double>()));
      hipLaunchKernelGGL(lin3_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t0.as<double>(), 1.0, b.t2.as<double>(), 1.0,
                         b.X.as<double>(), -sigma_mu, b.Si.as<double>(), mm_);
      hipLaunchKernelGGL(lin3_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t2.as<double>(), 1.0, b.t0.as<double>(), -1.0,
                         b.Qm.as<double>(), 0.0, (const double*)nullptr, mm_);
      LRN_TRY(aa_times(c, b, b.t2.as<double>(), c->v1.as<double>()));
      continue;
    }
    double* G = b.G.as<double>();
    // t1 = G' Rd G
    LRN_TRY(mm(c, m, G, true, b.Rd.as<double>(), false, b.t0.as<double>()));
    LRN_TRY(mm(c, m, b.t0.as<double>(), false, G, false, b.t1.as<double>()));
    hipLaunchKernelGGL(corr_inner_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t1.as<double>(), b.D.as<double>(),
                       b.RNT.as<double>(), sigma_mu, b.t0.as<double>(), m);
    // my_kron(G,G,inner) = vec(G inner G')
    LRN_TRY(mm(c, m, G, false, b.t0.as<double>(), false, b.t1.as<double>()));
    LRN_TRY(mm(c, m, b.t1.as<double>(), false, G, true, b.t2.as<double>()));
    LRN_TRY(aa_times(c, b, b.t2.as<double>(), c->v1.as<double>()));
  }
  toc(c, "rhs");
  return copy_out(c, out, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_ip_find_step(lrn_ctx* c, int predict, double sigma_mu, double tau, const double* dely,
                                double* alpha, double* beta) {
  if (!c || !dely || !alpha || !beta) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LRN_TRY(copy_in(c, c->v0.p, dely, (size_t)c->nvar * 8));
  hipEvent_t a0, a1;
  if (c->profile) { (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventRecord(a0, c->stream); }
  for (int il = 0; il < c->nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    LRN_TRY(ensure_resident(c, b));
    const int m = b.msz;
    const long mm_ = (long)m * m;
    const unsigned g = nbk(mm_);
    double *t0 = b.t0.as<double>(), *t1 = b.t1.as<double>(), *t2 = b.t2.as<double>();
    double *G = b.G.as<double>(), *Gi = b.Gi.as<double>();
    // delS = Rd - mat(AA' dely)                                   (:252)
    LRN_TRY(aat_to_mat(c, b, c->v0.as<double>(), t0));
    hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, b.delS.as<double>(), 1.0, b.Rd.as<double>(), -1.0, t0,
                       0.0, (const double*)nullptr, mm_);
    double lamX = 0.0, lamS = 0.0;
    double* t3 = b.Xn.as<double>();          // free here: Xn / Sn are rebuilt by lrn_ip_update after the step lengths
    if (b.nt_free) {
      // everything in the L_X basis (prepw.hip::prepare_w_ns): Bs = L_X' dS L_X, T = Z Bs Z / c,
      // TX = L_X^-1 dX L_X^-T = -I - T (+ sigma_mu K^-1 + R), dX = L_X TX L_X'                  (:253-257)
      const unsigned gs = (unsigned)std::min<long>(1024, ((long)(m + 31) / 32) * ((m + 31) / 32));
      LRN_TRY(pgemm_nt(c, c->stream, m, b.LXt.as<double>(), b.delS.as<double>(), t0, GEMM_KFROM_M));    // L_X' upper triangular
      // (both second products are symmetric in exact arithmetic: from msz 1500 on the lower tiles + mirror, half the work;
      // below, the full product and the average of the two triangles)
      if (m >= 1500) {
        LRN_TRY(pgemm_nt_sym(c, c->stream, m, t0, b.LXt.as<double>(), b.Bs.as<double>(), 1.0, GEMM_KFROM_N));
      } else {
        SlabSrc prod;
        LRN_TRY(prod_slabs(c, c->stream, m, t0, b.LXt.as<double>(), t1, 1.0, GEMM_KFROM_N, &prod));
        hipLaunchKernelGGL(symadd_kernel, dim3(gs), dim3(256), 0, c->stream, prod, m, 0.5, b.Bs.as<double>(), (const double*)nullptr,
                           (double*)nullptr);
      }
      LRN_TRY(pgemm_nt(c, c->stream, m, b.Zh.as<double>(), b.Bs.as<double>(), t0));
      if (m >= 1500) {
        LRN_TRY(pgemm_nt_sym(c, c->stream, m, t0, b.Zh.as<double>(), t3, 1.0 / b.ns_c));
      } else {
        SlabSrc prod;
        LRN_TRY(prod_slabs(c, c->stream, m, t0, b.Zh.as<double>(), t1, 1.0 / b.ns_c, 0, &prod));
        hipLaunchKernelGGL(symadd_kernel, dim3(gs), dim3(256), 0, c->stream, prod, m, 0.5, t3, (const double*)nullptr, (double*)nullptr);
      }
      hipLaunchKernelGGL(tx_kernel, dim3(g), dim3(256), 0, c->stream, t3, sigma_mu / b.ns_c, b.Ki.as<double>(),
                         predict ? (const double*)nullptr : b.RNT.as<double>(), b.TX.as<double>(), m);
      LRN_TRY(pgemm_nt(c, c->stream, m, b.LXf.as<double>(), b.TX.as<double>(), t0, GEMM_KTO_M));        // L_X lower triangular
      LRN_TRY(pgemm_nt_sym(c, c->stream, m, t0, b.LXf.as<double>(), b.delX.as<double>(), 1.0, GEMM_KTO_N));
      // the scaled directions of the step-length rule are orthogonally similar to TX and T               (:263-285)
      LRN_TRY(eigmin_certified_pair(c, b.TX.as<double>(), t3, m, &lamX, &lamS));
      alpha[il] = lamX > -1e-6 ? 0.99 : std::min(1.0, -tau / lamX);
      beta[il] = lamS > -1e-6 ? 0.99 : std::min(1.0, -tau / lamS);
      continue;
    }
    // t2 = W delS W                                                (:253)
    LRN_TRY(wmw(c, b, b.delS.as<double>(), t1, t2));
    if (predict) {
      // delX = mat(-X - W delS W)                                  (:255)
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, t0, -1.0, b.X.as<double>(), -1.0, t2, 0.0,
                         (const double*)nullptr, mm_);
    } else {
      // delX = mat(sigma_mu Si - X - W delS W + G RNT G')          (:257)
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, t0, sigma_mu, b.Si.as<double>(), -1.0,
                         b.X.as<double>(), -1.0, t2, mm_);
      LRN_TRY(mm(c, m, G, false, b.RNT.as<double>(), false, t1));
      LRN_TRY(mm(c, m, t1, false, G, true, t2));
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, t0, 1.0, t0, 1.0, t2, 0.0, (const double*)nullptr, mm_);
    }
    sym_half(c->stream, t0, b.delX.as<double>(), m);
    // step lengths: eigmin of DDsi-scaled G' delS G and Gi delX Gi'   (:263-291)
    LRN_TRY(mm(c, m, Gi, false, b.delX.as<double>(), false, t0));
    LRN_TRY(mm(c, m, t0, false, Gi, true, t1));
    hipLaunchKernelGGL(scale_sym_kernel, dim3(g), dim3(256), 0, c->stream, t1, b.DDsi.as<double>(), t2, m);
    LRN_TRY(mm(c, m, G, true, b.delS.as<double>(), false, t0));
    LRN_TRY(mm(c, m, t0, false, G, false, t1));
    hipLaunchKernelGGL(scale_sym_kernel, dim3(g), dim3(256), 0, c->stream, t1, b.DDsi.as<double>(), t3, m);
    LRN_TRY(eigmin_certified_pair(c, t2, t3, m, &lamX, &lamS));
    alpha[il] = lamX > -1e-6 ? 0.99 : std::min(1.0, -tau / lamX);
    beta[il] = lamS > -1e-6 ? 0.99 : std::min(1.0, -tau / lamS);
  }
  if (c->profile) {
    (void)hipEventRecord(a1, c->stream); (void)hipEventSynchronize(a1);
    float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
    c->timing["find_step"] += ms; c->counts["find_step"] += 1;
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
  }
  return LRN_OK;
}

extern "C" int lrn_ip_update(lrn_ctx* c, int predict, const double* alpha, const double* beta, double* trXnSn) {
  if (!c || !alpha || !beta) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  std::vector<double> tr(c->nlmi, 0.0);
  LRN_TRY(ensure(c, c->redout, 64 * 8));
  for (int il = 0; il < c->nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    LRN_TRY(ensure_resident(c, b));
    const int m = b.msz;
    const long mm_ = (long)m * m;
    const unsigned g = nbk(mm_);
    double *t0 = b.t0.as<double>(), *t1 = b.t1.as<double>(), *t2 = b.t2.as<double>();
    if (predict) {
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, b.Xn.as<double>(), 1.0, b.X.as<double>(), alpha[il],
                         b.delX.as<double>(), 0.0, (const double*)nullptr, mm_);
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, b.Sn.as<double>(), 1.0, b.S.as<double>(), beta[il],
                         b.delS.as<double>(), 0.0, (const double*)nullptr, mm_);
      LRN_TRY(dot_dev(c, b.Xn.as<double>(), b.Sn.as<double>(), mm_, c->redout.as<double>() + il));
      if (b.nt_free) {
        // Qm = G RNT G' without the eigenvectors: N = L_X^-1 dX dS L_X = TX Bs, Yh R + R Yh = -(N Zh + Zh N') / c,
        // Qm = L_X R L_X'
        tic(c);
        LRN_TRY(pgemm_nt(c, c->stream, m, b.TX.as<double>(), b.Bs.as<double>(), t0));              // N
        SlabSrc prod;
        LRN_TRY(prod_slabs(c, c->stream, m, t0, b.Zh.as<double>(), t1, 1.0, 0, &prod));            // N Zh
        hipLaunchKernelGGL(symadd_kernel, dim3((unsigned)std::min<long>(1024, ((long)(m + 31) / 32) * ((m + 31) / 32))), dim3(256), 0,
                           c->stream, prod, m, -1.0 / b.ns_c, t2, (const double*)nullptr, (double*)nullptr);
        bool ok = false;
        int steps = 0;
        LRN_TRY(lyap_solve(c, b, t2, b.RNT.as<double>(), t0, &ok, &steps));
        c->counts["lyap_steps"] += steps;
        c->counts["lyap_solves"] += 1;
        if (ok) {
          LRN_TRY(pgemm_nt(c, c->stream, m, b.LXf.as<double>(), b.RNT.as<double>(), t0, GEMM_KTO_M));
          LRN_TRY(pgemm_nt_sym(c, c->stream, m, t0, b.LXf.as<double>(), b.Qm.as<double>(), 1.0, GEMM_KTO_N));
          toc(c, "lyap");
          continue;
        }
        // the Lyapunov iteration did not converge (K far from well conditioned): take the SVD for this iteration
        toc(c, "lyap");
        c->counts["lyap_fallback"] += 1;
        int info = 0;
        LRN_TRY(prepare_w_block(c, b, &info));
        if (info != 0) return set_error(c, LRN_ERR_STATE, "SVD fallback of the NT scaling failed (info %d)", info);
        b.nt_free = false;
      }
      // RNT = -(Gi delX delS G + its transpose) ./ (D_i + D_j)     (:308-309)
      LRN_TRY(mm(c, m, b.Gi.as<double>(), false, b.delX.as<double>(), false, t0));
      LRN_TRY(mm(c, m, t0, false, b.delS.as<double>(), false, t1));
      LRN_TRY(mm(c, m, t1, false, b.G.as<double>(), false, t2));
      hipLaunchKernelGGL(rnt_kernel, dim3(g), dim3(256), 0, c->stream, t2, b.D.as<double>(), b.RNT.as<double>(), m);
    } else {
      b.chol_valid = false;
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, t0, 1.0, b.X.as<double>(), alpha[0],
                         b.delX.as<double>(), 0.0, (const double*)nullptr, mm_);
      sym_half(c->stream, t0, b.X.as<double>(), m);
      hipLaunchKernelGGL(lin3_kernel, dim3(g), dim3(256), 0, c->stream, t0, 1.0, b.S.as<double>(), beta[0],
                         b.delS.as<double>(), 0.0, (const double*)nullptr, mm_);
      sym_half(c->stream, t0, b.S.as<double>(), m);
    }
  }
  if (predict && trXnSn && c->nlmi > 0) LRN_TRY(copy_out(c, trXnSn, c->redout.p, (size_t)c->nlmi * 8));
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

extern "C" int lrn_ip_stats(lrn_ctx* c, double* out5) {
  if (!c || !out5) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LRN_TRY(ensure(c, c->redout, (size_t)std::max(64, 5 * c->nlmi) * 8));
  for (int il = 0; il < c->nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    LRN_TRY(ensure_resident(c, b));
    const long mm_ = (long)b.msz * b.msz;
    double* o = c->redout.as<double>() + 5 * il;
    LRN_TRY(dot_dev(c, b.X.as<double>(), b.S.as<double>(), mm_, o + 0));
    LRN_TRY(dot_dev(c, b.Rd.as<double>(), nullptr, mm_, o + 3));
    LRN_TRY(dot_dev(c, b.Cd.as<double>(), b.X.as<double>(), mm_, o + 4));
  }
  LRN_TRY(copy_out(c, out5, c->redout.p, (size_t)5 * c->nlmi * 8));
  for (int il = 0; il < c->nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    double lx = 0, ls = 0;
    bool have = false;
    if (c->opt.nt_mode == 1 && b.msz > 1) {
      // the callers consume max(0, -eigmin) (Solvers.jl:503-511): a successful Cholesky factorisation IS the certificate
      // that both terms vanish, and the next prepare_W starts with exactly these two factorisations -- they are kept
      // (b.chol_valid).  Reported then: the smallest pivots (positive upper bounds of the smallest eigenvalues).
      int info = 0;
      double piv[2] = {0.0, 0.0};
      LRN_TRY(nt_factor(c, b, &info, piv));
      if (info == 0) { lx = piv[0]; ls = piv[1]; have = true; c->counts["stats_chol"] += 1; }
    }
    if (!have) LRN_TRY(eigmin_certified_pair(c, b.X.as<double>(), b.S.as<double>(), b.msz, &lx, &ls));
    out5[5 * il + 1] = lx;
    out5[5 * il + 2] = ls;
    out5[5 * il + 3] = std::sqrt(out5[5 * il + 3]);
  }
  return LRN_OK;
}

extern "C" int lrn_dbg_get_block(lrn_ctx* c, int il, const char* name, double* out, int* flag) {
  if (!c || il < 0 || il >= c->nlmi || !name || !out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LmiBlock& b = c->lmi[il];
  const size_t mm_ = (size_t)b.msz * b.msz * 8, mv = (size_t)b.msz * 8;
  struct { const char* n; DBuf* d; size_t bytes; } tab[] = {
      {"W", &b.W, mm_}, {"Si", &b.Si, mm_}, {"G", &b.G, mm_}, {"Gi", &b.Gi, mm_}, {"D", &b.D, mv}, {"DDsi", &b.DDsi, mv},
      {"X", &b.X, mm_}, {"S", &b.S, mm_}, {"delX", &b.delX, mm_}, {"delS", &b.delS, mm_}, {"RNT", &b.RNT, mm_},
      {"LX", &b.LXf, mm_}, {"LS", &b.LSf, mm_}, {"Bs", &b.Bs, mm_}, {"TX", &b.TX, mm_}, {"Ki", &b.Ki, mm_}, {"Yh", &b.Yh, mm_},
      {"Zh", &b.Zh, mm_}, {"Qm", &b.Qm, mm_}};
  if (flag) *flag = b.nt_free ? 1 : 0;
  c->timing["ns_c"] = b.ns_c;
  for (auto& t : tab)
    if (!strcmp(name, t.n)) {
      if (!t.d->p || t.d->bytes < t.bytes) return set_error(c, LRN_ERR_STATE, "block array %s not allocated", name);
      return copy_out(c, out, t.d->p, t.bytes);
    }
  return set_error(c, LRN_ERR_ARG, "unknown block array %s", name);
}

extern "C" int lrn_dbg_eigmin(lrn_ctx* c, int n, const double* M, double* lam, int* steps) {
  if (!c || n <= 0 || !M || !lam) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  DBuf d;
  LRN_TRY(ensure(c, d, (size_t)n * n * 8));
  LRN_TRY(copy_in(c, d.p, M, (size_t)n * n * 8));
  int rc = steps ? eigmin_dev(c, d.as<double>(), n, lam, steps) : eigmin_certified(c, d.as<double>(), n, lam);
  release(d);
  return rc;
}

// ---- builder-defined synthetic problem on top of lrn_synthetic_dense_model
namespace lrn {
__global__ void synth_x0_kernel(double* __restrict__ X, const double* __restrict__ Q, int n, int r) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    double s = 0.0;
    for (int k = 0; k < r; ++k) s += Q[i + (long)k * n] * Q[j + (long)k * n];
    X[e] = s / n + (i == j ? 1.0 : 0.0);
  }
}
__global__ void eye_add_kernel(double* __restrict__ out, const double* __restrict__ M, double sgn, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    out[e] = sgn * M[e] + ((int)(e % n) == (int)(e / n) ? 1.0 : 0.0);
}
}  // namespace lrn

extern "C" int lrn_synthetic_dense_problem(lrn_ctx* c, uint64_t seed, double* b_out, double* y0_out, double* normC) {
  if (!c || c->nlmi != 1 || !b_out || !y0_out) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  LmiBlock& b = c->lmi[0];
  LRN_TRY(ensure_resident(c, b));
  const int m = b.msz, n = c->nvar;
  const long mm_ = (long)m * m;
  // host-side small random factors (deterministic LCG-free: splitmix64 + Box-Muller)
  auto next = [&seed]() {
    seed += 0x9E3779B97F4A7C15ull;
    uint64_t z = seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto unif = [&]() { return ((double)(next() >> 11) + 0.5) / 9007199254740992.0; };
  auto gauss = [&]() { return std::sqrt(-2.0 * std::log(unif())) * std::cos(6.283185307179586 * unif()); };
  const int r = 8;
  std::vector<double> Q((size_t)m * r), y0(n);
  for (auto& v : Q) v = gauss();
  for (auto& v : y0) v = gauss() / std::sqrt((double)n);
  LRN_TRY(copy_in(c, b.t1.p, Q.data(), Q.size() * 8));
  hipLaunchKernelGGL(synth_x0_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.t0.as<double>(), b.t1.as<double>(), m, r);
  // b = AA vec(X0)
  LRN_HIP(c, hipMemsetAsync(c->v1.p, 0, (size_t)n * 8, c->stream));
  LRN_TRY(aa_times(c, b, b.t0.as<double>(), c->v1.as<double>()));
  LRN_TRY(copy_out(c, b_out, c->v1.p, (size_t)n * 8));
  // C = I + mat(AA' y0)
  LRN_TRY(copy_in(c, c->v0.p, y0.data(), (size_t)n * 8));
  LRN_TRY(aat_to_mat(c, b, c->v0.as<double>(), b.t0.as<double>()));
  hipLaunchKernelGGL(eye_add_kernel, dim3(nbk(mm_)), dim3(256), 0, c->stream, b.Cd.as<double>(), b.t0.as<double>(), 1.0, m);
  if (normC) {
    LRN_TRY(ensure(c, c->redout, 64 * 8));
    LRN_TRY(dot_dev(c, b.Cd.as<double>(), nullptr, mm_, c->redout.as<double>()));
    double s = 0;
    LRN_TRY(copy_out(c, &s, c->redout.p, 8));
    *normC = std::sqrt(s);
  }
  memcpy(y0_out, y0.data(), (size_t)n * 8);
  return LRN_OK;
}
