// One-sided (Hestenes) Jacobi SVD in FP64 for gfx950.
//
// Replaces  fsvd(CC)  in the NT scaling (reference src/prepare_W.jl:42) and  eigen(W)  in the
// preconditioner setup (src/Solvers.jl:642,706; W = G G' so eig(W) = svd(G)^2).
// A V = U Sigma: plane rotations orthogonalise the columns of A in place, V accumulates
// them; singular values are the final column norms.  High relative accuracy, no squaring
// of the condition number, and every round of the round-robin ordering is n/2 independent
// column pairs -> one workgroup per pair.
//   * n <= 96: one workgroup keeps A and V in LDS and runs all sweeps in one launch.
//   * larger n: one launch per round (n-1 rounds per sweep), columns streamed from L2/MALL.
#include "ctx.h"
#include "jacobi.h"

namespace lrn {

__device__ __forceinline__ void rr_pair(int np, int round, int k, int* p, int* q) {
  // round-robin tournament on np (even) players; pair k of round `round`
  int m = np - 1;
  int a, b;
  if (k == 0) { a = m; b = round; }
  else { a = (round + k) % m; b = (round - k + m) % m; }
  *p = a < b ? a : b;
  *q = a < b ? b : a;
}

__device__ __forceinline__ double block_sum256(double v, double* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// one workgroup per column pair of this round
__global__ __launch_bounds__(256) void jacobi_round_kernel(double* __restrict__ A, double* __restrict__ V,
                                                          int n, int np, int round, double tol,
                                                          int* __restrict__ nrot) {
  __shared__ double sh[4];
  int p, q;
  rr_pair(np, round, blockIdx.x, &p, &q);
  if (q >= n) return;            // padded player
  double* ap = A + (long)p * n;
  double* aq = A + (long)q * n;
  double al = 0.0, be = 0.0, ga = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    double x = ap[i], y = aq[i];
    al += x * x; be += y * y; ga += x * y;
  }
  al = block_sum256(al, sh);
  be = block_sum256(be, sh);
  ga = block_sum256(ga, sh);
  if (!(fabs(ga) > tol * sqrt(al * be))) return;     // also skips zero / NaN columns
  double zeta = (be - al) / (2.0 * ga);
  double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
  for (int i = threadIdx.x; i < n; i += 256) {
    double x = ap[i], y = aq[i];
    ap[i] = c * x - s * y;
    aq[i] = s * x + c * y;
  }
  if (V) {
    double* vp = V + (long)p * n;
    double* vq = V + (long)q * n;
    for (int i = threadIdx.x; i < n; i += 256) {
      double x = vp[i], y = vq[i];
      vp[i] = c * x - s * y;
      vq[i] = s * x + c * y;
    }
  }
  if (threadIdx.x == 0) atomicAdd(nrot, 1);
}

// whole SVD in one workgroup, n <= JAC_SMALL
__global__ __launch_bounds__(256) void jacobi_small_kernel(double* __restrict__ A, double* __restrict__ V,
                                                          int n, double tol, int max_sweeps,
                                                          int* __restrict__ sweeps_out) {
  extern __shared__ double smem[];
  double* a = smem;                 // n x n, column stride n+1
  double* v = smem + (size_t)n * (n + 1);
  __shared__ int rot;
  const int ld = n + 1;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int e = t; e < n * n; e += 256) {
    int i = e % n, j = e / n;
    a[i + j * ld] = A[e];
    if (V) v[i + j * ld] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int np = (n + 1) & ~1;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (t == 0) rot = 0;
    __syncthreads();
    for (int round = 0; round < np - 1; ++round) {
      for (int k = w; k < np / 2; k += 4) {
        int p, q;
        rr_pair(np, round, k, &p, &q);
        if (q >= n) continue;
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int i = lane; i < n; i += 64) {
          double x = a[i + p * ld], y = a[i + q * ld];
          al += x * x; be += y * y; ga += x * y;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          al += __shfl_xor(al, off, 64);
          be += __shfl_xor(be, off, 64);
          ga += __shfl_xor(ga, off, 64);
        }
        if (!(fabs(ga) > tol * sqrt(al * be))) continue;
        double zeta = (be - al) / (2.0 * ga);
        double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
        for (int i = lane; i < n; i += 64) {
          double x = a[i + p * ld], y = a[i + q * ld];
          a[i + p * ld] = c * x - s * y;
          a[i + q * ld] = s * x + c * y;
          if (V) {
            double vx = v[i + p * ld], vy = v[i + q * ld];
            v[i + p * ld] = c * vx - s * vy;
            v[i + q * ld] = s * vx + c * vy;
          }
        }
        if (lane == 0) rot = 1;
      }
      __syncthreads();
    }
    int r = rot;
    __syncthreads();
    if (!r) { ++sweep; break; }
  }
  for (int e = t; e < n * n; e += 256) {
    int i = e % n, j = e / n;
    A[e] = a[i + j * ld];
    if (V) V[e] = v[i + j * ld];
  }
  if (t == 0 && sweeps_out) *sweeps_out = sweep;
}

__global__ void set_identity_kernel(double* __restrict__ V, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    V[e] = (e % n == e / n) ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void colnorm_kernel(const double* __restrict__ A, int n, double* __restrict__ sig) {
  __shared__ double sh[4];
  const double* a = A + (long)blockIdx.x * n;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += a[i] * a[i];
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) sig[blockIdx.x] = sqrt(s);
}

int jacobi_svd(lrn_ctx* c, double* A, double* V, double* sigma, int n, int* sweeps_out) {
  hipStream_t st = c->stream;
  const double tol = 2.220446049250313e-16 * sqrt((double)(n > 4 ? n : 4));
  const int max_sweeps = 40;
  int sweeps = 0;
  LRN_TRY(ensure(c, c->info_dev, 64));
  int* cnt = c->info_dev.as<int>() + 4;
  if (n <= JAC_SMALL) {
    size_t sh = (size_t)2 * n * (n + 1) * 8;
    hipFuncSetAttribute((const void*)jacobi_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(jacobi_small_kernel, dim3(1), dim3(256), sh, st, A, V, n, tol, max_sweeps, cnt);
    LRN_HIP(c, hipMemcpyAsync(&sweeps, cnt, 4, hipMemcpyDeviceToHost, st));
    LRN_HIP(c, hipStreamSynchronize(st));
  } else {
    if (V) hipLaunchKernelGGL(set_identity_kernel, dim3(1024), dim3(256), 0, st, V, n);
    const int np = (n + 1) & ~1;
    for (; sweeps < max_sweeps;) {
      LRN_HIP(c, hipMemsetAsync(cnt, 0, 4, st));
      for (int round = 0; round < np - 1; ++round)
        hipLaunchKernelGGL(jacobi_round_kernel, dim3(np / 2), dim3(256), 0, st, A, V, n, np, round, tol, cnt);
      int h = 0;
      LRN_HIP(c, hipMemcpyAsync(&h, cnt, 4, hipMemcpyDeviceToHost, st));
      LRN_HIP(c, hipStreamSynchronize(st));
      ++sweeps;
      if (h == 0) break;
    }
  }
  hipLaunchKernelGGL(colnorm_kernel, dim3(n), dim3(256), 0, st, A, n, sigma);
  if (sweeps_out) *sweeps_out = sweeps;
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

}  // namespace lrn
