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
#include <algorithm>
#include <cstdio>
#include <cstdlib>

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
                                                          int* __restrict__ sweeps_out, int v_init) {
  extern __shared__ double smem[];
  double* a = smem;                 // n x n, column stride n+1
  double* v = smem + (size_t)n * (n + 1);
  __shared__ int rot;
  const int ld = n + 1;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int e = t; e < n * n; e += 256) {
    int i = e % n, j = e / n;
    a[i + j * ld] = A[e];
    if (V) v[i + j * ld] = v_init ? V[e] : ((i == j) ? 1.0 : 0.0);
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

// ------------------------------------------------------------------ blocked rounds
// Block one-sided Jacobi on 16-column blocks; a round = nbk/2 disjoint block pairs (I,J), and
// every pair's n x 32 panel P = [A_I A_J] is split into row chunks so that a round fills the
// chip.  Three launches per round:
//   jb_gram_kernel    partial Gram matrices P_chunk' P_chunk        (FP64 MFMA)
//   jb_rotate_kernel  G = sum of partials; one parallel-ordered cyclic sweep of plane rotations
//                     on the 32x32 Gram matrix in LDS (same rotation formula / threshold as
//                     the scalar kernel), accumulated in R
//   jb_apply_kernel   P <- P R and V_panel <- V_panel R              (FP64 MFMA)
// nbk-1 rounds make every pair of columns meet once per sweep.
static constexpr int JB = 16;

__global__ __launch_bounds__(256) void jb_gram_kernel(const double* __restrict__ A, int n, int nbk2, int round,
                                                      int RC, double* __restrict__ Gpart) {
  __shared__ double part[4][3][256];
  int I, J;
  rr_pair(nbk2, round, blockIdx.x, &I, &J);
  const int cI = I * JB, cJ = J * JB;
  if (cI >= n) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ci = lane & 15, kq = lane >> 4;
  v4f64 aII = {0, 0, 0, 0}, aIJ = aII, aJJ = aII;
  const bool okI = cI + ci < n, okJ = cJ + ci < n;
  const double* pI = A + (long)(cI + ci) * n;
  const double* pJ = A + (long)(cJ + ci) * n;
  const int rbeg = blockIdx.y * RC, rend = min(n, rbeg + RC);
  for (int r0 = rbeg + 4 * w; r0 < rend; r0 += 16) {
    int row = r0 + kq;
    double xI = (okI && row < rend) ? pI[row] : 0.0;
    double xJ = (okJ && row < rend) ? pJ[row] : 0.0;
    aII = __builtin_amdgcn_mfma_f64_16x16x4f64(xI, xI, aII, 0, 0, 0);
    aIJ = __builtin_amdgcn_mfma_f64_16x16x4f64(xI, xJ, aIJ, 0, 0, 0);
    aJJ = __builtin_amdgcn_mfma_f64_16x16x4f64(xJ, xJ, aJJ, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int i = kq + 4 * r;                         // f64 C/D map: row = (lane>>4) + 4*reg, col = lane&15
    part[w][0][i * 16 + ci] = aII[r];
    part[w][1][i * 16 + ci] = aIJ[r];
    part[w][2][i * 16 + ci] = aJJ[r];
  }
  __syncthreads();
  double* out = Gpart + ((long)blockIdx.x * gridDim.y + blockIdx.y) * 768;
  for (int e = t; e < 768; e += 256) {
    int tile = e >> 8, o = e & 255;
    out[e] = part[0][tile][o] + part[1][tile][o] + part[2][tile][o] + part[3][tile][o];
  }
}

// Plane-rotation parameters that annihilate the (p,q) Gram entry (Hestenes / same formula as the
// scalar kernel); returns false when the pair is already orthogonal to tolerance.
// The rotation parameters sit on the critical path of every step of the Gram sweep (31 or 63 dependent
// steps per round): v_rsq_f64 seeds with two Newton steps (full double accuracy for the
// normal-range arguments that occur here) instead of the IEEE sqrt / divide expansions.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}

__device__ __forceinline__ bool jrot(double al, double be, double ga, double tol, double* c, double* s) {
  *c = 1.0; *s = 0.0;
  if (!(ga * ga > tol * tol * al * be)) return false;      // |ga| > tol sqrt(al be), also rejects NaN
  // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (be - al) / (2 ga), written as
  //   t = 2 ga / (d + sign(d) sqrt(d^2 + 4 ga^2)),  d = be - al
  //   c^2 = (1 + |d| / r) / 2,  s = sign(d) ga / (r c),  r = sqrt(d^2 + 4 ga^2)  -- the same (small-angle) rotation
  // with two reciprocal square roots on the dependent chain and no reciprocal
  double d = be - al, g2 = 2.0 * ga;
  double r2 = d * d + g2 * g2;
  double ir = fast_rsqrt(r2);
  double c2 = 0.5 + 0.5 * fabs(d) * ir;
  double ic = fast_rsqrt(c2);
  *c = c2 * ic;
  *s = (d >= 0.0 ? ga : -ga) * ir * ic;
  return true;
}

// One cyclic sweep over the 32x32 Gram matrix: 31 steps of 16 disjoint rotations.  With
// J = diag of the step's 2x2 rotations, G' = J'GJ decomposes into 256 independent 2x2 blocks
// (pair k1 x pair k2) -> one thread per block, every thread recomputes the two rotations it
// needs from the diagonal blocks, G is double-buffered: ONE barrier per step.
// `full` = 0: only the 256 CROSS pairs (p in block I, q in block J) in 16 steps, pair k of step s = (k, 16 + (k + s) % 16).
// The within-block pairs were made orthogonal when the block last met a partner with a full sweep; the driver asks
// for the full schedule in round 0 of every sweep, where each block occurs once -- so a sweep still visits every
// column pair once, at about half the dependent steps (the latency of this kernel is the round's time at msz ~ 1000).
__global__ __launch_bounds__(256) void jb_rotate_kernel(const double* __restrict__ Gpart, int nchunk, int n, int nbk2,
                                                        int round, double tol, double big2, int inner, int full,
                                                        double* __restrict__ Rbuf,
                                                        int* __restrict__ flags, int* __restrict__ nrot) {
  __shared__ double G[2][32][33];
  __shared__ double R[32][33];
  __shared__ unsigned char sched[31][16][2];   // round-robin schedule of the 32 local columns
  __shared__ int anyrot, anybig;
  int I, J;
  rr_pair(nbk2, round, blockIdx.x, &I, &J);
  const int t = threadIdx.x;
  if (I * JB >= n) {
    if (t == 0) flags[blockIdx.x] = 0;
    return;
  }
  if (t == 0) { anyrot = 0; anybig = 0; }
  const int nsteps = full ? 31 : 16;
  for (int e = t; e < nsteps * 16; e += 256) {
    int p, q;
    if (full) rr_pair(32, e >> 4, e & 15, &p, &q);
    else { p = e & 15; q = 16 + (((e & 15) + (e >> 4)) & 15); }
    sched[e >> 4][e & 15][0] = (unsigned char)p;
    sched[e >> 4][e & 15][1] = (unsigned char)q;
  }
  const double* gp = Gpart + (long)blockIdx.x * nchunk * 768;
  for (int e = t; e < 1024; e += 256) {
    int i = e >> 5, j = e & 31;
    int tile, ii, jj;
    if (i < 16 && j < 16) { tile = 0; ii = i; jj = j; }
    else if (i < 16) { tile = 1; ii = i; jj = j - 16; }
    else if (j < 16) { tile = 1; ii = j; jj = i - 16; }      // G_JI = G_IJ'
    else { tile = 2; ii = i - 16; jj = j - 16; }
    int o = tile * 256 + ii * 16 + jj;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += gp[(long)c * 768 + o];
    G[0][i][j] = s;
    R[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int k1 = t >> 4, k2 = t & 15;
  int cur = 0;
  bool rotated = false, big = false;
  // R <- R J of a step is applied one step late: its LDS round trip then runs under the next step's rotation
  // parameters (one wave per SIMD: every latency of the chain is exposed) -- (pc, ps, pp, pq) is the pending rotation
  double pc = 1.0, ps = 0.0;
  int pp = 0, pq = 0;
  const int src = (threadIdx.x & 0x30) | k1;
  // inner > 1: further sweeps on the same Gram matrix (cheap next to streaming the panels at large n)
  for (int istep = 0; istep < nsteps * inner; ++istep) {
    const int step = istep % nsteps;
    int p1, q1, p2, q2;
    if (full) {
      p1 = sched[step][k1][0]; q1 = sched[step][k1][1];
      p2 = sched[step][k2][0]; q2 = sched[step][k2][1];
    } else {
      p1 = k1; q1 = 16 + ((k1 + step) & 15);
      p2 = k2; q2 = 16 + ((k2 + step) & 15);
    }
    // every LDS read of the step is issued before the dependent FP64 chain of the rotation parameters
    const double al = G[cur][p2][p2], be = G[cur][q2][q2], ga = G[cur][p2][q2];
    const double b00 = G[cur][p1][p2], b01 = G[cur][p1][q2], b10 = G[cur][q1][p2], b11 = G[cur][q1][q2];
    double rx[2], ry[2];
    if (istep > 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) { rx[h] = R[k1 + 16 * h][pp]; ry[h] = R[k1 + 16 * h][pq]; }
    }
    // one rotation per thread (pair k2); the row pair's (k1) comes from the lane of this wave whose
    // k2 equals my k1 -- halves the dependent FP64 chain of a step
    double c1, s1, c2, s2;
    rotated |= jrot(al, be, ga, tol, &c2, &s2);
    // "above the early-stop level": the columns were further from orthogonal than `early`, OR the rotation that was applied
    // has a large angle -- inside a cluster of singular values (relative width < early) the cosines stay below the
    // level while the angles are O(1), and such rotations re-perturb pairs annihilated earlier to first order
    big |= ga * ga > big2 * al * be || s2 * s2 > 1e-4;
    c1 = __shfl(c2, src, 64);
    s1 = __shfl(s2, src, 64);
    // 2x2 block (pair k1 rows, pair k2 columns):  B' = J1' B J2
    double t00 = c2 * b00 - s2 * b01, t01 = s2 * b00 + c2 * b01;
    double t10 = c2 * b10 - s2 * b11, t11 = s2 * b10 + c2 * b11;
    G[cur ^ 1][p1][p2] = c1 * t00 - s1 * t10;
    G[cur ^ 1][p1][q2] = c1 * t01 - s1 * t11;
    G[cur ^ 1][q1][p2] = s1 * t00 + c1 * t10;
    G[cur ^ 1][q1][q2] = s1 * t01 + c1 * t11;
    // R <- R J of the PREVIOUS step: rows k1 and k1+16, its column pair k2 (each element owned by one thread)
    if (istep > 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        R[k1 + 16 * h][pp] = pc * rx[h] - ps * ry[h];
        R[k1 + 16 * h][pq] = ps * rx[h] + pc * ry[h];
      }
    }
    pc = c2; ps = s2; pp = p2; pq = q2;
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {           // the last step's rotation
    double x = R[k1 + 16 * h][pp], y = R[k1 + 16 * h][pq];
    R[k1 + 16 * h][pp] = pc * x - ps * y;
    R[k1 + 16 * h][pq] = ps * x + pc * y;
  }
  if (rotated) anyrot = 1;
  if (big) anybig = 1;
  __syncthreads();
  double* ro = Rbuf + (long)blockIdx.x * 1024;
  for (int e = t; e < 1024; e += 256) ro[e] = R[e >> 5][e & 31];
  if (t == 0) {
    flags[blockIdx.x] = anyrot;
    if (anyrot) atomicAdd(nrot, 1);
    if (anybig) atomicAdd(nrot + 1, 1);           // a rotation above the early-stop level (jacobi_svd)
  }
}

// (P R)' = R' P': output rows land on consecutive lanes (coalesced loads and stores)
__global__ __launch_bounds__(256) void jb_apply_kernel(double* __restrict__ A, double* __restrict__ V, int n, int nbk2,
                                                       int round, int RC, const double* __restrict__ Rbuf,
                                                       const int* __restrict__ flags) {
  __shared__ double R[32][33];
  if (!flags[blockIdx.x]) return;
  double* Mx = blockIdx.z == 0 ? A : V;
  int I, J;
  rr_pair(nbk2, round, blockIdx.x, &I, &J);
  const int cI = I * JB, cJ = J * JB;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ci = lane & 15, kq = lane >> 4;
  const double* ri = Rbuf + (long)blockIdx.x * 1024;
  for (int e = t; e < 1024; e += 256) R[e >> 5][e & 31] = ri[e];
  __syncthreads();
  const int rbeg = blockIdx.y * RC, rend = min(n, rbeg + RC);
  for (int r0 = rbeg + 16 * w; r0 < rend; r0 += 64) {
    const int row = r0 + ci;
    double pk[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      int k = 4 * ks + kq;
      int col = k < 16 ? cI + k : cJ + k - 16;
      pk[ks] = (row < rend && col < n) ? Mx[(long)row + (long)col * n] : 0.0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      v4f64 acc = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(R[4 * ks + kq][16 * h + ci], pk[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = kq + 4 * r;                      // output column within the half
        int col = h == 0 ? cI + m : cJ + m;
        if (row < rend && col < n) Mx[(long)row + (long)col * n] = acc[r];
      }
    }
  }
}


// ------------------------------------------------------------------ 32-column blocks (large n)
// Every round streams the whole matrix (Gram: read, apply: read + write), so at large n a sweep is
// bound by HBM traffic = rounds x 3 x 8 n^2 bytes.  Doubling the block width halves the rounds;
// the 64 x 64 Gram sweep (63 steps of 32 rotations, 1024 independent 2x2 blocks per step) costs more
// per round but is amortised once n >= ~5000.  Same structure as the 16-column kernels above.
static constexpr int JB2 = 32;
__device__ __constant__ signed char jb2_tile_of[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};

__global__ __launch_bounds__(256) void jb2_gram_kernel(const double* __restrict__ A, int n, int nbk2, int round, int RC,
                                                       double* __restrict__ Gpart) {
  __shared__ double part[10][256];
  int I, J;
  rr_pair(nbk2, round, blockIdx.x, &I, &J);
  const int cI = I * JB2, cJ = J * JB2;
  if (cI >= n) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ci = lane & 15, kq = lane >> 4;
  v4f64 acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = v4f64{0, 0, 0, 0};
  const double* pc[4];
  bool ok[4];
#pragma unroll
  for (int sb = 0; sb < 4; ++sb) {
    int col = (sb < 2 ? cI : cJ) + 16 * (sb & 1) + ci;
    ok[sb] = col < n;
    pc[sb] = A + (long)(ok[sb] ? col : 0) * n;
  }
  const int rbeg = blockIdx.y * RC, rend = min(n, rbeg + RC);
  // lane (ci, kq) holds rows r0 + 4 kq .. +3 of its column: 16 lanes cover 128 contiguous bytes.
  // The next slab is loaded while the 40 MFMAs of the current one issue (few waves per SIMD).
  const bool vec = (n & 1) == 0;            // 16-byte alignment of (col * n + row), row % 4 == 0
  auto load = [&](int r0, double (&x)[4][4]) {
    const int row = r0 + 4 * kq;
    if (vec && row + 3 < rend) {
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        if (ok[sb]) {
          const double2 lo = *reinterpret_cast<const double2*>(pc[sb] + row);
          const double2 hi = *reinterpret_cast<const double2*>(pc[sb] + row + 2);
          x[sb][0] = lo.x; x[sb][1] = lo.y; x[sb][2] = hi.x; x[sb][3] = hi.y;
        } else {
          x[sb][0] = x[sb][1] = x[sb][2] = x[sb][3] = 0.0;
        }
      }
    } else {
#pragma unroll
      for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[sb][j] = (ok[sb] && row + j < rend) ? pc[sb][row + j] : 0.0;
    }
  };
  double xa[4][4], xb[4][4];
  int r0 = rbeg + 16 * w;
  if (r0 < rend) load(r0, xa);
  for (; r0 < rend; r0 += 64) {
    const bool more = r0 + 64 < rend;
    if (more) load(r0 + 64, xb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int ti = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) {
          acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a][j], xa[b][j], acc[ti], 0, 0, 0);
          ++ti;
        }
    }
    if (more) {
#pragma unroll
      for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[sb][j] = xb[sb][j];
    }
  }
  // reduce the four waves in turn through one 20 KB image
  for (int ph = 0; ph < 4; ++ph) {
    if (w == ph) {
#pragma unroll
      for (int ti = 0; ti < 10; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int o = (kq + 4 * r) * 16 + ci;          // f64 C/D map: row = (lane>>4) + 4*reg, col = lane&15
          part[ti][o] = (ph == 0 ? 0.0 : part[ti][o]) + acc[ti][r];
        }
    }
    __syncthreads();
  }
  double* out = Gpart + ((long)blockIdx.x * gridDim.y + blockIdx.y) * 2560;
  for (int e = t; e < 2560; e += 256) out[e] = part[e >> 8][e & 255];
}

// One (or `inner`) cyclic sweep(s) over the 64 x 64 Gram matrix: 63 steps of 32 disjoint rotations,
// one thread per 2x2 block (pair k1 x pair k2), double-buffered, one barrier per step.
__global__ __launch_bounds__(1024) void jb2_rotate_kernel(const double* __restrict__ Gpart, int nchunk, int n, int nbk2,
                                                          int round, double tol, double big2, int inner, int full,
                                                          double* __restrict__ Rbuf,
                                                          int* __restrict__ flags, int* __restrict__ nrot) {
  extern __shared__ double jb2_sh[];
  double (*G)[64][65] = reinterpret_cast<double (*)[64][65]>(jb2_sh);                 // [2][64][65]
  double (*R)[65] = reinterpret_cast<double (*)[65]>(jb2_sh + 2 * 64 * 65);           // [64][65]
  unsigned char (*sched)[32][2] = reinterpret_cast<unsigned char (*)[32][2]>(jb2_sh + 3 * 64 * 65);   // [63][32][2]
  __shared__ int anyrot, anybig;
  int I, J;
  rr_pair(nbk2, round, blockIdx.x, &I, &J);
  const int t = threadIdx.x;
  if (I * JB2 >= n) {
    if (t == 0) flags[blockIdx.x] = 0;
    return;
  }
  if (t == 0) { anyrot = 0; anybig = 0; }
  // `full` = 0: the 1024 cross pairs only, in 32 steps (as jb_rotate_kernel; the driver asks for all 63 in round 0)
  const int nsteps = full ? 63 : 32;
  for (int e = t; e < nsteps * 32; e += 1024) {
    int p, q;
    if (full) rr_pair(64, e >> 5, e & 31, &p, &q);
    else { p = e & 31; q = 32 + (((e & 31) + (e >> 5)) & 31); }
    sched[e >> 5][e & 31][0] = (unsigned char)p;
    sched[e >> 5][e & 31][1] = (unsigned char)q;
  }
  const double* gp = Gpart + (long)blockIdx.x * nchunk * 2560;
  for (int e = t; e < 4096; e += 1024) {
    int i = e >> 6, j = e & 63;
    int a = i >> 4, b = j >> 4;
    int o = a <= b ? jb2_tile_of[a][b] * 256 + (i & 15) * 16 + (j & 15) : jb2_tile_of[b][a] * 256 + (j & 15) * 16 + (i & 15);
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += gp[(long)c * 2560 + o];
    G[0][i][j] = s;
    R[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int k1 = t >> 5, k2 = t & 31;
  int cur = 0;
  bool rotated = false, big = false;
  double pc = 1.0, ps = 0.0;                          // R <- R J one step late, as in jb_rotate_kernel
  int pp = 0, pq = 0;
  const int src = (threadIdx.x & 0x20) | k1;          // lane of this wave with k2 == my k1
  for (int istep = 0; istep < nsteps * inner; ++istep) {
    const int step = istep % nsteps;
    const int p1 = sched[step][k1][0], q1 = sched[step][k1][1];
    const int p2 = sched[step][k2][0], q2 = sched[step][k2][1];
    const double al = G[cur][p2][p2], be = G[cur][q2][q2], ga = G[cur][p2][q2];
    const double b00 = G[cur][p1][p2], b01 = G[cur][p1][q2], b10 = G[cur][q1][p2], b11 = G[cur][q1][q2];
    double rx[2], ry[2];
    if (istep > 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) { rx[h] = R[k1 + 32 * h][pp]; ry[h] = R[k1 + 32 * h][pq]; }
    }
    double c1, s1, c2, s2;
    rotated |= jrot(al, be, ga, tol, &c2, &s2);
    // "above the early-stop level": the columns were further from orthogonal than `early`, OR the rotation that was applied
    // has a large angle -- inside a cluster of singular values (relative width < early) the cosines stay below the
    // level while the angles are O(1), and such rotations re-perturb pairs annihilated earlier to first order
    big |= ga * ga > big2 * al * be || s2 * s2 > 1e-4;
    c1 = __shfl(c2, src, 64);
    s1 = __shfl(s2, src, 64);
    double t00 = c2 * b00 - s2 * b01, t01 = s2 * b00 + c2 * b01;
    double t10 = c2 * b10 - s2 * b11, t11 = s2 * b10 + c2 * b11;
    G[cur ^ 1][p1][p2] = c1 * t00 - s1 * t10;
    G[cur ^ 1][p1][q2] = c1 * t01 - s1 * t11;
    G[cur ^ 1][q1][p2] = s1 * t00 + c1 * t10;
    G[cur ^ 1][q1][q2] = s1 * t01 + c1 * t11;
    if (istep > 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        R[k1 + 32 * h][pp] = pc * rx[h] - ps * ry[h];
        R[k1 + 32 * h][pq] = ps * rx[h] + pc * ry[h];
      }
    }
    pc = c2; ps = s2; pp = p2; pq = q2;
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double x = R[k1 + 32 * h][pp], y = R[k1 + 32 * h][pq];
    R[k1 + 32 * h][pp] = pc * x - ps * y;
    R[k1 + 32 * h][pq] = ps * x + pc * y;
  }
  if (rotated) anyrot = 1;
  if (big) anybig = 1;
  __syncthreads();
  double* ro = Rbuf + (long)blockIdx.x * 4096;
  for (int e = t; e < 4096; e += 1024) ro[e] = R[e >> 6][e & 63];
  if (t == 0) {
    flags[blockIdx.x] = anyrot;
    if (anyrot) atomicAdd(nrot, 1);
    if (anybig) atomicAdd(nrot + 1, 1);
  }
}

__global__ __launch_bounds__(256) void jb2_apply_kernel(double* __restrict__ A, double* __restrict__ V, int n, int nbk2,
                                                        int round, int RC, const double* __restrict__ Rbuf,
                                                        const int* __restrict__ flags) {
  __shared__ double R[64][65];
  if (!flags[blockIdx.x]) return;
  double* Mx = blockIdx.z == 0 ? A : V;
  int I, J;
  rr_pair(nbk2, round, blockIdx.x, &I, &J);
  const int cI = I * JB2, cJ = J * JB2;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ci = lane & 15, kq = lane >> 4;
  const double* ri = Rbuf + (long)blockIdx.x * 4096;
  for (int e = t; e < 4096; e += 256) R[e >> 6][e & 63] = ri[e];
  __syncthreads();
  const int rbeg = blockIdx.y * RC, rend = min(n, rbeg + RC);
  for (int r0 = rbeg + 16 * w; r0 < rend; r0 += 64) {
    const int row = r0 + ci;
    double pk[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      int k = 4 * ks + kq;
      int col = k < 32 ? cI + k : cJ + k - 32;
      pk[ks] = (row < rend && col < n) ? Mx[(long)row + (long)col * n] : 0.0;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      v4f64 acc = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(R[4 * ks + kq][16 * h + ci], pk[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = 16 * h + kq + 4 * r;             // output column within the 64-column panel
        int col = m < 32 ? cI + m : cJ + m - 32;
        if (row < rend && col < n) Mx[(long)row + (long)col * n] = acc[r];
      }
    }
  }
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

int jacobi_svd(lrn_ctx* c, double* A, double* V, double* sigma, int n, int* sweeps_out, bool v_init) {
  hipStream_t st = c->stream;
  const double tol = 2.220446049250313e-16 * sqrt((double)(n > 4 ? n : 4));
  // early stop (option "jacobi_early", default 3e-8; 0 = always run the confirming sweep): see the sweep loop
  const double early = c->opt.jacobi_early;
  const double big2 = early > 0.0 ? early * early : 0.0;
  const int max_sweeps = 40;
  int sweeps = 0;
  LRN_TRY(ensure(c, c->info_dev, 64));
  int* cnt = c->info_dev.as<int>() + 4;
  if (n <= JAC_SMALL) {
    size_t sh = (size_t)2 * n * (n + 1) * 8;
    hipFuncSetAttribute((const void*)jacobi_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(jacobi_small_kernel, dim3(1), dim3(256), sh, st, A, V, n, tol, max_sweeps, cnt, v_init ? 1 : 0);
    LRN_HIP(c, hipMemcpyAsync(&sweeps, cnt, 4, hipMemcpyDeviceToHost, st));
    LRN_HIP(c, hipStreamSynchronize(st));
  } else {
    if (V && !v_init) hipLaunchKernelGGL(set_identity_kernel, dim3(1024), dim3(256), 0, st, V, n);
    const bool wide = c->opt.jacobi_block == 32 || (c->opt.jacobi_block == 0 && n >= 5000);
    const int jb = wide ? JB2 : JB;
    const int gsz = wide ? 2560 : 768, rsz = wide ? 4096 : 1024;
    const int nbk = (n + jb - 1) / jb;
    const int nbk2 = (nbk + 1) & ~1;
    const int npair = nbk2 / 2;
    const int wg_target = c->opt.jacobi_wgs > 0 ? c->opt.jacobi_wgs : (wide ? 1536 : (n < 1500 ? 256 : 512));
    int nchunk = (wg_target + npair - 1) / npair;
    nchunk = std::max(1, std::min(nchunk, (n + 63) / 64));
    int RC = (n + nchunk - 1) / nchunk;
    RC = ((RC + 63) / 64) * 64;
    nchunk = (n + RC - 1) / RC;
    LRN_TRY(ensure(c, c->jscratch, ((size_t)npair * nchunk * gsz + (size_t)npair * rsz) * 8 + (size_t)npair * 4 + 64));
    double* Gpart = c->jscratch.as<double>();
    double* Rbuf = Gpart + (size_t)npair * nchunk * gsz;
    int* flags = reinterpret_cast<int*>(Rbuf + (size_t)npair * rsz);
    static const bool trace = getenv("LRN_JACOBI_TRACE") != nullptr;
    const int inner = c->opt.jacobi_inner > 0 ? c->opt.jacobi_inner : 1;
    const size_t sh2 = (size_t)3 * 64 * 65 * 8 + 63 * 32 * 2;
    if (wide)
      LRN_HIP(c, hipFuncSetAttribute((const void*)jb2_rotate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2));
    for (; sweeps < max_sweeps;) {
      LRN_HIP(c, hipMemsetAsync(cnt, 0, 8, st));
      for (int round = 0; round < nbk2 - 1; ++round) {
        if (wide) {
          hipLaunchKernelGGL(jb2_gram_kernel, dim3(npair, nchunk), dim3(256), 0, st, A, n, nbk2, round, RC, Gpart);
          hipLaunchKernelGGL(jb2_rotate_kernel, dim3(npair), dim3(1024), sh2, st, Gpart, nchunk, n, nbk2, round, tol,
                             big2, inner, (round == 0 || c->opt.jacobi_cross == 0) ? 1 : 0, Rbuf, flags, cnt);
          hipLaunchKernelGGL(jb2_apply_kernel, dim3(npair, nchunk, V ? 2 : 1), dim3(256), 0, st, A, V, n, nbk2, round, RC,
                             Rbuf, flags);
        } else {
          hipLaunchKernelGGL(jb_gram_kernel, dim3(npair, nchunk), dim3(256), 0, st, A, n, nbk2, round, RC, Gpart);
          hipLaunchKernelGGL(jb_rotate_kernel, dim3(npair), dim3(256), 0, st, Gpart, nchunk, n, nbk2, round, tol,
                             big2, inner, (round == 0 || c->opt.jacobi_cross == 0) ? 1 : 0, Rbuf, flags, cnt);
          hipLaunchKernelGGL(jb_apply_kernel, dim3(npair, nchunk, V ? 2 : 1), dim3(256), 0, st, A, V, n, nbk2, round, RC,
                             Rbuf, flags);
        }
      }
      int h[2] = {0, 0};
      LRN_HIP(c, hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, st));
      LRN_HIP(c, hipStreamSynchronize(st));
      ++sweeps;
      if (trace)
        fprintf(stderr, "[jacobi n=%d jb=%d] sweep %d: %d of %d block pairs rotated, %d above the early-stop level\n", n, jb,
                sweeps, h[0], npair * (nbk2 - 1), h[1]);
      // no rotation at all, or none whose columns were further from orthogonal than `early`: what such a sweep leaves
      // is of second order in it (quadratic convergence of the cyclic sweep) -- below tol, so the sweep that would only
      // confirm it is not run
      if (h[0] == 0 || h[1] == 0) break;
    }
  }
  hipLaunchKernelGGL(colnorm_kernel, dim3(n), dim3(256), 0, st, A, n, sigma);
  if (sweeps_out) *sweeps_out = sweeps;
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

}  // namespace lrn
