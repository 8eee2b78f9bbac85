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
#include <cstdlib>

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
  // column j of the factor, as it is produced: written once by the wave and read back by every lane as LDS
  // broadcasts (uniform address) -- the rank-one update a[k] -= l_ij l_kj needs l_kj of lane k in every lane, and 2016
  // such values per block through v_readlane (two per value, one SGPR pair, a wait state each) cost more (41 -> 38 us
  // per block; 63 blocks per factorisation of the C4 Schur matrix.  The 64 columns as ONE basic block with selects
  // instead of the per-column branches: 164 us -- the scheduler hoists the reads and spills).  One wave: its LDS
  // operations execute in order, no barrier.
  __shared__ double colbuf[2][NB];
  const int i = threadIdx.x;
  if (*info != 0) return;
  double a[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) a[j] = (i < nb && j < nb && i >= j) ? A[(long)i + (long)j * ld] : (i == j ? 1.0 : 0.0);
  double d0v = 1.0;                              // lane j: original diagonal entry of column col0 + j (pivot boosting)
  if (diag0 && i < nb) d0v = diag0[col0 + i];
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (bad == 0) {
      double piv = readlane_f64(a[j], j);
      if (diag0 && j < nb) {
        const double d0 = readlane_f64(d0v, j);
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
        const double sq = sqrt(piv);
        const double rl = 1.0 / sq;
        const double lij = a[j] * rl;
        a[j] = (i == j) ? sq : lij;
        double* cb = colbuf[j & 1];
        cb[i] = lij;
#pragma unroll
        for (int k = j + 1; k < NB; ++k) a[k] -= lij * cb[k];
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

// The diagonal block by FOUR wavefronts in eight panels of 8 columns (round 3).  The one-wave kernel above spends 28 of
// its 37.5 us in the per-column chain -- pivot broadcast, sqrt, divide, an LDS round trip -- in front of 63 - j FMAs that
// the in-order wave cannot overlap with it.  Here the block lives in LDS; wave 0 factors a 64 x 8 panel with the rows in
// 8 registers per lane (the chain: v_readlane of the pivot, v_rsq_f64 + one Goldschmidt and one Newton step -- 8 dependent
// operations instead of the ~45 of sqrt and divide -- and at most 7 lagging FMAs per column, their multipliers by
// v_readlane), then all four waves apply the rank-8 update to the 16 x 16 blocks of the trailing part on the MFMA
// (two v_mfma_f64_16x16x4 per block; block columns that the panel itself crosses are masked in the B operand).
// Entries above the diagonal are scratch.  Same pivot rules as above (boosting, first failing column in info[0]).
__device__ __forceinline__ void rsqrt_pair(double p, double& sq, double& rinv) {
  if (p > 1e-280 && p < 1e280) {
    const double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double dd = fma(-g, g, p);
    g = fma(dd, h, g);                    // sqrt(p), one Newton correction
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);                     // 1 / (2 sqrt(p))
    sq = g;
    rinv = 2.0 * h;
  } else {
    sq = sqrt(p);
    rinv = 1.0 / sq;
  }
}

// (M holds the block -- lower part, identity-padded beyond nb -- and bad_s = 0, visible to the whole workgroup on entry;
// on return M holds the factor, or bad_s the failing column; ends with a barrier)
// `leader` = false: a replica of the factorisation (potrf_step_kernel) -- the same arithmetic, boosted pivots included, but no
// count of them (the leader's count decides) .
__device__ __forceinline__ void diag_block_factor(double (*M)[NB + 1], int& bad_s, int nb, int col0, int* __restrict__ info,
                                                  const double* __restrict__ diag0, double boost, int max_boost,
                                                  bool leader = true) {
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  double d0v = 1.0;                              // wave 0, lane j: original diagonal entry of column col0 + j
  if (w == 0 && diag0 && lane < nb) d0v = diag0[col0 + lane];
  for (int p8 = 0; p8 < NB / 8; ++p8) {
    const int c0 = 8 * p8;
    if (w == 0) {
      double pr[8], pr0[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) pr0[q] = pr[q] = M[lane][c0 + q];
      int bad = 0;
      // the eight columns as straight-line code (the lagging FMAs of a column fill the latency of the next column's
      // chain): valid if every pivot is an ordinary positive number above the boosting threshold; otherwise the panel
      // is redone from its saved registers by the careful loop below (rare: a numerically singular Schur matrix)
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = c0 + q;
        const double piv = readlane_f64(pr[q], j);
        double lo = 1e-280;
        if (diag0 && j < nb) {
          const double d0 = readlane_f64(d0v, j);
          if (d0 > 0.0) lo = fmax(lo, boost * d0);
        }
        ok = ok && (piv > lo) && (piv < 1e280);
        const double y = __builtin_amdgcn_rsq(piv);
        double g = piv * y, h = 0.5 * y;
        double r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
        const double dd = fma(-g, g, piv);
        g = fma(dd, h, g);
        r = fma(-h, g, 0.5);
        h = fma(h, r, h);
        const double lij = pr[q] * (2.0 * h);
        pr[q] = (lane == j) ? g : lij;
#pragma unroll
        for (int k = q + 1; k < 8; ++k) pr[k] -= lij * readlane_f64(lij, c0 + k);
      }
      if (!ok) {
#pragma unroll
        for (int q = 0; q < 8; ++q) pr[q] = pr0[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = c0 + q;
          if (bad == 0) {
            double piv = readlane_f64(pr[q], j);
            if (diag0 && j < nb) {
              const double d0 = readlane_f64(d0v, j);
              if (d0 > 0.0 && piv <= boost * d0 && piv == piv) {
                piv = 1e40 * fmax(fabs(d0), 1.0);
                int cnt = 0;
                if (lane == 0 && leader) cnt = atomicAdd(info + 1, 1) + 1;
                cnt = __builtin_amdgcn_readfirstlane(cnt);
                if (cnt > max_boost) bad = col0 + j + 1;
              }
            }
            if (!(piv > 0.0)) bad = col0 + j + 1;        // also catches NaN; wave-uniform
            if (bad == 0) {
              double sq, rinv;
              rsqrt_pair(piv, sq, rinv);
              const double lij = pr[q] * rinv;
              pr[q] = (lane == j) ? sq : lij;
#pragma unroll
              for (int k = q + 1; k < 8; ++k) pr[k] -= lij * readlane_f64(lij, c0 + k);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) M[lane][c0 + q] = pr[q];
      if (bad && lane == 0) bad_s = bad;
    }
    __syncthreads();
    if (bad_s) break;
    // rank-8 update of the blocks (I, J), I >= J, that hold columns >= c0 + 8
    const int c1 = c0 + 8, Jmin = c1 >> 4;
    int cnt = 0;
    for (int J = Jmin; J < NB / 16; ++J)
      for (int I = J; I < NB / 16; ++I, ++cnt) {
        if ((cnt & 3) != w) continue;
        const int cr = lane >> 4, cc = lane & 15;
        v4f64 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = M[16 * I + cr + 4 * r][16 * J + cc];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const double a = -M[16 * I + cc][c0 + 4 * kk + cr];
          const double b = (16 * J + cc >= c1) ? M[16 * J + cc][c0 + 4 * kk + cr] : 0.0;
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) M[16 * I + cr + 4 * r][16 * J + cc] = c[r];
      }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void potrf_diag_blk_kernel(double* __restrict__ A, int ld, int nb, int col0,
                                                             int* __restrict__ info, const double* __restrict__ diag0,
                                                             double boost, int max_boost) {
  __shared__ double M[NB][NB + 1];
  __shared__ int bad_s;
  const int t = threadIdx.x;
  if (*info != 0) return;
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e % NB, j = e / NB;
    M[i][j] = (i < nb && j < nb && i >= j) ? A[(long)i + (long)j * ld] : (i == j ? 1.0 : 0.0);
  }
  if (t == 0) bad_s = 0;
  __syncthreads();
  diag_block_factor(M, bad_s, nb, col0, info, diag0, boost, max_boost);
  if (bad_s) {
    if (t == 0) atomicCAS(info, 0, bad_s);
    return;
  }
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e % NB, j = e / NB;
    if (i < nb && j < nb && i >= j) A[(long)i + (long)j * ld] = M[i][j];
  }
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

// The same panel solve with EIGHT lanes per row (round 3).  One thread per row walks 2016 dependent FMA + LDS-read pairs
// (30 us per panel, n / 64 panels per factorisation: a third of the factorisation of the C2 / C3 / truss matrices).  Here
// lane (row, b) owns the 8 columns [8 b, 8 b + 8) of its row: in stage s the lanes with b = s solve their 8 x 8
// triangular block and publish x through LDS, the lanes with b > s subtract its contribution from their columns -- a
// critical path of 8 x (36 + 64) pairs instead of 2016.  32 rows per workgroup.
__global__ __launch_bounds__(256) void potrf_panel8_kernel(double* __restrict__ A21, int ld, int rem,
                                                           const double* __restrict__ Lkk, double* __restrict__ W,
                                                           const int* __restrict__ info) {
  __shared__ double l[NB][NB + 1];
  __shared__ double rinv[NB];
  __shared__ double xs[32][NB + 1];
  if (*info != 0) return;
  const int t = threadIdx.x;
  for (int e = t; e < NB * NB; e += 256) {
    int i = e % NB, j = e / NB;
    l[i][j] = i >= j ? Lkk[(long)i + (long)j * ld] : 0.0;
  }
  __syncthreads();
  if (t < NB) rinv[t] = 1.0 / l[t][t];
  const int rl = t >> 3, b = t & 7;                 // row of the workgroup, block of 8 columns
  const int row = blockIdx.x * 32 + rl;
  const bool live = row < rem;
  double a[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = live ? A21[(long)row + (long)(8 * b + c) * ld] : 0.0;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (b == s) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= a[k] * l[8 * s + c][8 * s + k];
        v *= rinv[8 * s + c];
        a[c] = v;
        xs[rl][8 * s + c] = v;
      }
    }
    __syncthreads();
    if (b > s) {
      double x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = xs[rl][8 * s + k];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        double v = a[c];
#pragma unroll
        for (int k = 0; k < 8; ++k) v -= x[k] * l[8 * b + c][8 * s + k];
        a[c] = v;
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    A21[(long)row + (long)(8 * b + c) * ld] = a[c];
    W[(long)row + (long)(8 * b + c) * rem] = a[c];
  }
}

// The panel solve by strips of 16 rows, one wavefront per strip, no barrier after the prologue (round 3).  The strip
// X (16 x 64) stays in registers as four 16 x 16 blocks in the MFMA result layout.  For each block column c: the block
// goes to LDS, lanes 0..15 (one row each) solve it against the 16 x 16 diagonal sub-block of L_kk right-looking -- a
// dependent chain of 16 multiply / FMA pairs, the multipliers l_kj as LDS broadcasts -- and the blocks c' > c take
// X_c L[c', c]' off on the MFMA (4 x v_mfma_f64_16x16x4 each).  Critical path per strip: 4 x (16-step chain + 4 MFMAs)
// instead of the 8 x (36 + 64) dependent FMA / LDS pairs and 8 barriers of the eight-lane kernel above
// (19 us at n = 800, 32 us at n = 4000).  No inverse of the diagonal sub-blocks: substitution, as everywhere in this file.
// The solve of one 16 x 64 strip (see the kernel below): C = the strip as four MFMA-layout blocks, X = its LDS image (in / out:
// the solved strip), Ls = L_kk (lower part read only), rinv = 1 / diag(L_kk).  One wave; LDS operations of one wave execute
// in order, no barrier.
__device__ __forceinline__ void strip_solve(v4f64 (&C)[4], double (*X)[NB + 1], double (*Ls)[NB + 1], const double* rinv,
                                            int lane) {
  const int cr = lane >> 4, cc = lane & 15;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) X[cr + 4 * r][16 * c + cc] = C[c][r];
    if (lane < 16) {
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = X[lane][16 * c + k];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x[j] *= rinv[16 * c + j];
#pragma unroll
        for (int k = j + 1; k < 16; ++k) x[k] -= x[j] * Ls[16 * c + k][16 * c + j];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) X[lane][16 * c + k] = x[k];
    }
#pragma unroll
    for (int c2 = c + 1; c2 < 4; ++c2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double a = -X[cc][16 * c + 4 * kk + cr];
        const double b = Ls[16 * c2 + cc][16 * c + 4 * kk + cr];
        C[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C[c2], 0, 0, 0);
      }
  }
}

// General form: strip row r, column c of X at X[r * sr + c * sc] (the panel: sr = 1, sc = ld; the block rows of a matrix
// right-hand side, one strip row per right-hand side: sr = ldb, sc = 1), a second copy at W[r * wr + c * wc]; nb < 64:
// L_kk identity-padded; `reverse`: X L_kk = A instead of X L_kk' = A (columns and L_kk indexed from the end, which
// makes it the same lower-triangular recurrence) -- the diagonal step of L' X = B.
__global__ __launch_bounds__(256) void potrf_panel_mfma_kernel(double* __restrict__ Xg, long sr, long sc, int nrows,
                                                               const double* __restrict__ Lkk, int ld, int nb, int reverse,
                                                               double* __restrict__ W, long wr, long wc,
                                                               const int* __restrict__ info) {
  __shared__ double Ls[NB][NB + 1];
  __shared__ double rinv[NB];
  __shared__ double Xs[4][16][NB + 1];
  if (info && *info != 0) return;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int e = t; e < NB * NB; e += 256) {
    const int i = e % NB, j = e / NB;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j < nb && i >= j) v = reverse ? Lkk[(long)(nb - 1 - j) + (long)(nb - 1 - i) * ld] : Lkk[(long)i + (long)j * ld];
    Ls[i][j] = v;
  }
  if (t < NB) {
    const int dgi = reverse ? nb - 1 - t : t;
    rinv[t] = t < nb ? 1.0 / Lkk[(long)dgi + (long)dgi * ld] : 1.0;
  }
  const int row0 = blockIdx.x * 64 + 16 * w;
  const int cr = lane >> 4, cc = lane & 15;
  v4f64 C[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + cr + 4 * r, col = 16 * cb + cc;
      const int pc = reverse ? nb - 1 - col : col;
      C[cb][r] = (row < nrows && col < nb) ? Xg[(long)row * sr + (long)pc * sc] : 0.0;
    }
  __syncthreads();
  if (row0 >= nrows) return;
  double (*X)[NB + 1] = Xs[w];
  strip_solve(C, X, Ls, rinv, lane);
  // the strip, its memory-contiguous dimension along the lanes
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rl = (sr <= sc) ? cc : cr + 4 * (i & 3);          // strip row
    const int col = (sr <= sc) ? cr + 4 * i : 16 * (i >> 2) + cc;
    const int row = row0 + rl;
    if (row < nrows && col < nb) {
      const double v = X[rl][col];
      const int pc = reverse ? nb - 1 - col : col;
      Xg[(long)row * sr + (long)pc * sc] = v;
      if (W) W[(long)row * wr + (long)pc * wc] = v;
    }
  }
}

// Trailing update of the factorisation, A22 -= Wk Wk' on the lower 64 x 64 tiles (Wk: rem x 64, contiguous), as a
// kernel of its own (round 3).  The general GEMM walks K = 64 in four 16-steps with a barrier each and reads C in its
// epilogue: five dependent global round trips for 2 MFLOP per tile -- 42 us per update at n = 4000 (two thirds of the
// factorisation), 12 us at n = 800.  Here a workgroup requests its C tile (into the accumulators, MFMA result layout, the
// lane-contiguous dimension along the columns of A22 in memory) and its two 64 x 64 panels at once -- ONE round trip --
// then runs 64 MFMAs per wave from LDS and stores the tile.
// The workgroup of tile (0, 0) -- the next diagonal block -- goes on to factor it (diag_block_factor) from LDS instead
// of storing it for a kernel of its own: one launch and one global round trip less per block column, and the pivot
// chain of block k + 1 runs beside the other tiles of update k.  (The sums start from zero and meet C once at the
// end, like in the general GEMM: starting from C costs a factor 5 in the backward error.)
__global__ __launch_bounds__(256) void potrf_syrk_kernel(double* __restrict__ C, int ld, int rem,
                                                         const double* __restrict__ Wp, int* __restrict__ info,
                                                         int fuse_diag, int col0, const double* __restrict__ diag0,
                                                         double boost, int max_boost) {
  constexpr int LS = NB + 8;     // k-rows 16 banks apart: every bank is hit by two of the 64 lanes of a fragment read -- the two
                                 // passes 512 bytes take anyway; with NB + 16 the two panels fill 80 KB and one workgroup a CU
  __shared__ double PA[NB][LS];
  __shared__ double PB[NB][LS];
  __shared__ int bad_s;
  const int I = blockIdx.x, J = blockIdx.y;        // tile row / column of A22
  if (J > I) return;
  if (*info != 0) return;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i0 = NB * I, j0 = NB * J;
  const int cr = lane >> 4, cc = lane & 15;
  // the C tile: wave w owns columns j0 + 16 w .. + 15 of the tile, all 64 rows (four 16 x 16 blocks)
  v4f64 cv[4], acc[4];
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 16 * bb + cc, j = j0 + 16 * w + cr + 4 * r;
      cv[bb][r] = (i < rem && j < rem) ? C[(long)i + (long)j * ld] : 0.0;
      acc[bb][r] = 0.0;
    }
  {
    const int row = t & 63, kq = t >> 6;
    double va[16], vb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = kq + 4 * q;
      va[q] = (j0 + row < rem) ? Wp[(long)(j0 + row) + (long)k * rem] : 0.0;
      vb[q] = (I != J && i0 + row < rem) ? Wp[(long)(i0 + row) + (long)k * rem] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      PA[kq + 4 * q][row] = va[q];
      if (I != J) PB[kq + 4 * q][row] = vb[q];
    }
  }
  if (t == 0) bad_s = 0;
  __syncthreads();
  const double (*Pb)[LS] = (I != J) ? PB : PA;
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    const double a = PA[4 * kk + cr][16 * w + cc];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
      acc[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Pb[4 * kk + cr][16 * bb + cc], acc[bb], 0, 0, 0);
  }
  if (fuse_diag && I == 0 && J == 0) {
    // this tile is the next diagonal block: factor it here
    const int nb = rem < NB ? rem : NB;
    __syncthreads();                               // (every wave is done with the panels)
    double (*M)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(&PA[0][0]);
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * bb + cc, j = 16 * w + cr + 4 * r;
        M[i][j] = (i < nb && j < nb && i >= j) ? cv[bb][r] - acc[bb][r] : (i == j ? 1.0 : 0.0);
      }
    __syncthreads();
    diag_block_factor(M, bad_s, nb, col0, info, diag0, boost, max_boost);
    if (bad_s) {
      if (t == 0) atomicCAS(info, 0, bad_s);
      return;
    }
    for (int e = t; e < NB * NB; e += 256) {
      const int i = e % NB, j = e / NB;
      if (i < nb && j < nb && i >= j) C[(long)i + (long)j * ld] = M[i][j];
    }
    return;
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 16 * bb + cc, j = j0 + 16 * w + cr + 4 * r;
      if (i < rem && j < rem) C[(long)i + (long)j * ld] = cv[bb][r] - acc[bb][r];
    }
}

// One launch per block column (round 3): the trailing update with panel k, the factorisation of diagonal block k + 1 AND
// the panel solve k + 1.  The workgroups of tile column 0 hold the rows of the next panel; each of them also forms tile
// (0, 0) from the panel rows it has loaded anyway (64 more MFMAs per wave), factors it -- a replica of what the workgroup
// of tile (0, 0) does, same arithmetic, same bits, no synchronisation between workgroups -- and solves its 64 rows by
// strips (strip_solve) straight from the registers of the update: the panel never makes the round trip through memory as
// an unsolved block, and the chain per block column is one kernel (update + 13 us of pivots + 4 x 16-step substitutions)
// instead of two.  Wn: the next panel (rem - 64 rows, contiguous) -- a second work buffer, the update reads Wp.
// The replicas must not read tile (0, 0) from A: the leader overwrites it with the factor, and a replica that is dispatched
// late (two factorisations on two streams share the workgroup slots) would read L for A.  They read T00 (64 x 64,
// contiguous), a copy of the tile that the workgroup of tile (1, 1) of the PREVIOUS step stored beside the matrix
// (T00n: the copy this step's (1, 1) workgroup leaves for the next step; both in the slack of the panel buffers).
__global__ __launch_bounds__(256) void potrf_step_kernel(double* __restrict__ C, int ld, int rem,
                                                         const double* __restrict__ Wp, double* __restrict__ Wn,
                                                         const double* __restrict__ T00, double* __restrict__ T00n,
                                                         int* __restrict__ info, int col0, const double* __restrict__ diag0,
                                                         double boost, int max_boost) {
  constexpr int LS = NB + 8;
  __shared__ double PA[NB][LS];
  __shared__ double PB[NB][LS];
  __shared__ double rinv[NB];
  __shared__ int bad_s;
  const int I = blockIdx.x, J = blockIdx.y;        // tile row / column of A22
  if (J > I) return;
  if (*info != 0) return;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i0 = NB * I, j0 = NB * J;
  const int cr = lane >> 4, cc = lane & 15;
  const bool replica = (J == 0 && I > 0);          // this workgroup solves the rows i0 .. i0 + 63 of the next panel
  v4f64 cv[4], acc[4], c00[4], a00[4];
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 16 * bb + cc, j = j0 + 16 * w + cr + 4 * r;
      cv[bb][r] = (i < rem && j < rem) ? C[(long)i + (long)j * ld] : 0.0;
      acc[bb][r] = 0.0;
      a00[bb][r] = 0.0;
      c00[bb][r] = replica ? T00[(16 * bb + cc) + (16 * w + cr + 4 * r) * NB] : 0.0;               // (rem > 64 here)
    }
  {
    const int row = t & 63, kq = t >> 6;
    double va[16], vb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = kq + 4 * q;
      va[q] = (j0 + row < rem) ? Wp[(long)(j0 + row) + (long)k * rem] : 0.0;
      vb[q] = (I != J && i0 + row < rem) ? Wp[(long)(i0 + row) + (long)k * rem] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      PA[kq + 4 * q][row] = va[q];
      if (I != J) PB[kq + 4 * q][row] = vb[q];
    }
  }
  if (t == 0) bad_s = 0;
  __syncthreads();
  const double (*Pb)[LS] = (I != J) ? PB : PA;
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    const double a = PA[4 * kk + cr][16 * w + cc];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
      acc[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Pb[4 * kk + cr][16 * bb + cc], acc[bb], 0, 0, 0);
    if (replica) {
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
        a00[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, PA[4 * kk + cr][16 * bb + cc], a00[bb], 0, 0, 0);
    }
  }
  if (J > 0) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * bb + cc, j = j0 + 16 * w + cr + 4 * r;
        if (i < rem && j < rem) {
          const double v = cv[bb][r] - acc[bb][r];
          C[(long)i + (long)j * ld] = v;
          if (I == 1 && J == 1) T00n[(i - NB) + (j - NB) * NB] = v;       // the next step's tile (0, 0), for its replicas
        }
      }
    return;
  }
  // ---- tile column 0: the next diagonal block (I == 0: the leader, which stores it) and the next panel (I > 0)
  const int nb = rem < NB ? rem : NB;
  __syncthreads();                                 // (every wave is done with the panels)
  double (*M)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(&PA[0][0]);
  double (*Xs)[16][NB + 1] = reinterpret_cast<double (*)[16][NB + 1]>(&PB[0][0]);
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * bb + cc, j = 16 * w + cr + 4 * r;
      const double d = replica ? c00[bb][r] - a00[bb][r] : cv[bb][r] - acc[bb][r];
      M[i][j] = (i < nb && j < nb && i >= j) ? d : (i == j ? 1.0 : 0.0);
      if (replica) Xs[bb][cc][j] = (i0 + i < rem) ? cv[bb][r] - acc[bb][r] : 0.0;       // row i of this tile, column j
    }
  __syncthreads();
  diag_block_factor(M, bad_s, nb, col0, info, diag0, boost, max_boost, !replica);
  if (bad_s) {
    if (!replica && t == 0) atomicCAS(info, 0, bad_s);
    return;
  }
  if (!replica) {
    for (int e = t; e < NB * NB; e += 256) {
      const int i = e % NB, j = e / NB;
      if (i < nb && j < nb && i >= j) C[(long)i + (long)j * ld] = M[i][j];
    }
    return;
  }
  if (t < NB) rinv[t] = 1.0 / M[t][t];
  __syncthreads();
  {
    // strip w of this tile: rows i0 + 16 w .. + 15, against the factor just computed
    double (*X)[NB + 1] = Xs[w];
    v4f64 S[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[cb][r] = X[cr + 4 * r][16 * cb + cc];
    strip_solve(S, X, M, rinv, lane);
    const int remn = rem - NB;                     // rows of the next panel
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = i0 + 16 * w + cc, col = cr + 4 * q;
      if (row < rem) {
        const double v = X[cc][col];
        C[(long)row + (long)col * ld] = v;
        Wn[(long)(row - NB) + (long)col * remn] = v;
      }
    }
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
  static const bool diag_wave = getenv("LRN_POTRF_DIAG1") != nullptr;       // (measurement: the one-wave kernel)
  static const bool one_lane = getenv("LRN_POTRF_PANEL1") != nullptr;
  static const bool eight_lanes = getenv("LRN_POTRF_PANEL8") != nullptr;    // (measurement: the round-3a kernel)
  static const bool use_gemm = getenv("LRN_POTRF_GEMM") != nullptr;         // (measurement: the general GEMM, as in round 2)
  static const bool no_fuse = getenv("LRN_POTRF_NOFUSE") != nullptr;        // (measurement: diagonal blocks as launches of their own)
  static const bool no_step = getenv("LRN_POTRF_NOSTEP") != nullptr;        // (measurement: diagonal + panel + update kernels)
  if (!no_step && !diag_wave && !one_lane && !eight_lanes && !use_gemm && !no_fuse && Linv && n > NB) {
    // one launch per block column (potrf_step_kernel); the panels alternate between `work` and `Linv` (n x NB doubles each)
    hipLaunchKernelGGL(potrf_diag_blk_kernel, dim3(1), dim3(256), 0, st, A, ld, NB, 0, info_dev, diag0, boost, max_boost);
    double* cur = work;
    hipLaunchKernelGGL(potrf_panel_mfma_kernel, dim3((n - NB + 63) / 64), dim3(256), 0, st, A + NB, 1L, (long)ld, n - NB, A, ld,
                       NB, 0, cur, 1L, (long)(n - NB), info_dev);
    // the copies of the next diagonal tile live behind the largest panel a buffer can hold: (n - NB) x NB doubles of
    // n x NB (work) / >= n x NB (Linv)
    const size_t t00_off = (size_t)(n - NB) * NB;
    {
      const int e0 = n - NB < NB ? n - NB : NB;        // tile (0, 0) of the first trailing matrix, as it is in A
      if (hipMemcpy2DAsync(cur + t00_off, (size_t)NB * 8, A + (long)NB + (long)NB * ld, (size_t)ld * 8, (size_t)e0 * 8, e0,
                           hipMemcpyDeviceToDevice, st) != hipSuccess)
        return LRN_ERR_HIP;
    }
    // Second blocking level (round 4; n >= 9000): above n ~ 6000 the step kernel is bound by the HBM traffic of the
    // trailing matrix, which it re-streams for every 64 columns (n = 20 000: 333 GB, 100 ms at 26 TFLOP/s).  Super-blocks of
    // SB = 1024 columns: inside one, a step updates only the super-block's own remaining columns (all rows below); the rest
    // of the trailing matrix gets the sixteen panels at once, C -= L_sb L_sb' with K = 1024 on the 128-tile direct-to-LDS GEMM
    // (a sixteenth of the traffic), and the next super-block starts like the factorisation itself (diagonal block + panel
    // as launches of their own, the copy of tile (0, 0) for the replicas).
    static const int sb_min = getenv("LRN_POTRF_SB_MIN") ? atoi(getenv("LRN_POTRF_SB_MIN")) : 9000;   // (6144: 5.1 -> 5.8 ms, slower; 10^4: 16.1 -> 14.8)
    // (width sweep at n = 10^4 / 20 000: 256 -> 14.8 / 73.8 ms, 512 -> 13.1 / 61.9, 1024 -> 12.4 / 58.0; one level: 16.1 / 100)
    static const int SB = getenv("LRN_POTRF_SB") ? std::max(2 * NB, (atoi(getenv("LRN_POTRF_SB")) / NB) * NB) : 16 * NB;
    if (n >= sb_min && (ld & 1) == 0) {
      for (int ks = 0; ks < n; ks += SB) {
        if (ks > 0) {
          // start of a super-block: its first diagonal block and panel (the trailing matrix is up to date: big update below)
          const int remk = n - ks;
          if (remk <= 0) break;
          hipLaunchKernelGGL(potrf_diag_blk_kernel, dim3(1), dim3(256), 0, st, A + (long)ks + (long)ks * ld, ld,
                             remk < NB ? remk : NB, ks, info_dev, diag0, boost, max_boost);
          if (remk <= NB) break;
          hipLaunchKernelGGL(potrf_panel_mfma_kernel, dim3((remk - NB + 63) / 64), dim3(256), 0, st,
                             A + (long)(ks + NB) + (long)ks * ld, 1L, (long)ld, remk - NB, A + (long)ks + (long)ks * ld, ld, NB, 0,
                             cur, 1L, (long)(remk - NB), info_dev);
          const int e0 = remk - NB < NB ? remk - NB : NB;
          if (hipMemcpy2DAsync(cur + t00_off, (size_t)NB * 8, A + (long)(ks + NB) + (long)(ks + NB) * ld, (size_t)ld * 8,
                               (size_t)e0 * 8, e0, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return LRN_ERR_HIP;
        }
        const int ke = ks + SB < n ? ks + SB : n;                 // end of the super-block
        for (int k0 = ks; k0 + NB < ke && n - k0 - NB > 0; k0 += NB) {
          const int rem = n - k0 - NB;
          const int nt = (rem + NB - 1) / NB;
          const int ntj = (ke - k0 - NB + NB - 1) / NB;           // tile columns of the trailing matrix inside the super-block
          double* nxt = (cur == work) ? Linv : work;
          hipLaunchKernelGGL(potrf_step_kernel, dim3(nt, ntj < nt ? ntj : nt), dim3(256), 0, st,
                             A + (long)(k0 + NB) + (long)(k0 + NB) * ld, ld, rem, cur, nxt, cur + t00_off, nxt + t00_off,
                             info_dev, k0 + NB, diag0, boost, max_boost);
          cur = nxt;
        }
        if (ke >= n) break;
        GemmDesc u;                                               // A[ke:, ke:] -= L[ke:, ks:ke] L[ke:, ks:ke]'  (lower tiles)
        u.A = A + (long)ke + (long)ks * ld; u.sAm = 1; u.sAk = ld;
        u.B = A + (long)ke + (long)ks * ld; u.sBk = ld; u.sBn = 1;
        u.C = A + (long)ke + (long)ke * ld; u.sCm = 1; u.sCn = ld;
        u.M = n - ke; u.N = n - ke; u.K = ke - ks;
        u.alpha = -1.0; u.beta = 1.0; u.flags = GEMM_TRI_LOWER;
        const int rcg = gemm(st, u);
        if (rcg) return rcg;
      }
      return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
    }
    for (int k0 = 0; n - k0 - NB > 0; k0 += NB) {
      const int rem = n - k0 - NB;
      const int nt = (rem + NB - 1) / NB;
      double* nxt = (cur == work) ? Linv : work;
      hipLaunchKernelGGL(potrf_step_kernel, dim3(nt, nt), dim3(256), 0, st, A + (long)(k0 + NB) + (long)(k0 + NB) * ld, ld,
                         rem, cur, nxt, cur + t00_off, nxt + t00_off, info_dev, k0 + NB, diag0, boost, max_boost);
      cur = nxt;
    }
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  bool diag_done = false;                 // the diagonal block of this step was factored by the previous update kernel
  for (int b = 0; b < nblk; ++b) {
    int k0 = b * NB;
    int nb = n - k0 < NB ? n - k0 : NB;
    double* Akk = A + (long)k0 + (long)k0 * ld;
    if (!diag_done) {
      if (diag_wave)
        hipLaunchKernelGGL(potrf_diag_wave_kernel, dim3(1), dim3(64), 0, st, Akk, ld, nb, k0, info_dev, diag0, boost,
                           max_boost);
      else
        hipLaunchKernelGGL(potrf_diag_blk_kernel, dim3(1), dim3(256), 0, st, Akk, ld, nb, k0, info_dev, diag0, boost,
                           max_boost);
    }
    diag_done = false;
    int rem = n - k0 - nb;
    if (rem <= 0) break;
    // panel: Wk = A21 * Lkk^-T      (rem x nb), by substitution
    if (one_lane || nb < NB)
      hipLaunchKernelGGL(potrf_panel_kernel, dim3((rem + 255) / 256), dim3(256), 0, st,
                         A + (long)(k0 + nb) + (long)k0 * ld, ld, rem, Akk, work, info_dev);
    else if (!eight_lanes)
      hipLaunchKernelGGL(potrf_panel_mfma_kernel, dim3((rem + 63) / 64), dim3(256), 0, st,
                         A + (long)(k0 + nb) + (long)k0 * ld, 1L, (long)ld, rem, Akk, ld, NB, 0, work, 1L, (long)rem, info_dev);
    else
      hipLaunchKernelGGL(potrf_panel8_kernel, dim3((rem + 31) / 32), dim3(256), 0, st,
                         A + (long)(k0 + nb) + (long)k0 * ld, ld, rem, Akk, work, info_dev);
    int rc;
    // trailing: A22 -= Wk Wk^T (lower tiles)
    if (!use_gemm && nb == NB) {
      const int nt = (rem + NB - 1) / NB;
      const int fuse = (!no_fuse && !diag_wave) ? 1 : 0;
      hipLaunchKernelGGL(potrf_syrk_kernel, dim3(nt, nt), dim3(256), 0, st, A + (long)(k0 + nb) + (long)(k0 + nb) * ld, ld,
                         rem, work, info_dev, fuse, k0 + nb, diag0, boost, max_boost);
      diag_done = fuse != 0;
      continue;
    }
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

// whole L^-T L^-1 h in ONE workgroup (n <= POTRS_SMALL): 2*nblk dependent block steps without
// launch gaps; the right-hand side lives in LDS.
static constexpr int POTRS_SMALL = 512;       // (above it the super-block kernels below win: n = 801 357 -> ~180 us)
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

// ------------------------------------------------------------------ triangular solves (vector), n > POTRS_SMALL
// Super-blocks of SB = 256 rows, one launch per super-block and direction (n = 4000: 16 + 16 launches per
// L'\(L\h) where the 64-row stepping took 126).  Launch b of the forward solve:
//   workgroup 0   subtracts the previous super-block's contribution from ITS 256 rows (a 256 x 256 mat-vec over 16
//                 waves), then solves the 256 x 256 diagonal block: four 64 x 64 substitutions, each by ONE wavefront
//                 with its row of the block in registers (no LDS traffic, no division: x_c = r_c * (1 / l_cc) travels
//                 by v_readlane), interleaved with the in-block updates by all 16 waves;
//   workgroups 1+ subtract the previous super-block's contribution from the rows further down (256 rows each).
// The chain of dependent work per launch is workgroup 0's; everything else overlaps with it.  The backward solve
// mirrors this on columns (dot products down the columns, one wavefront per column: coalesced) with the diagonal
// blocks transposed through LDS so that the same readlane substitution applies.  Substitution through the blocks
// themselves -- no inverted blocks (see the header of this file).
static constexpr int SB = 256;

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

static constexpr int TRSV_T = 512;      // 8 wavefronts: 2 per SIMD, so a wave may keep its 64 x 64 block row in registers

__global__ __launch_bounds__(TRSV_T) void trsv_fwd_sb_kernel(const double* __restrict__ L, int ld, int n, int k0,
                                                             double* __restrict__ r, double* __restrict__ y) {
  __shared__ double xs[SB];          // x of the previous super-block
  __shared__ double xc[SB];          // x of this one, sub-block by sub-block
  __shared__ double part[2][SB];
  const int t = threadIdx.x, rr = t & (SB - 1), cg = t >> 8;       // row in the block, half of the columns
  const int kp = k0 - SB;
  const int row = (blockIdx.x == 0 ? k0 : k0 + SB + ((int)blockIdx.x - 1) * SB) + rr;
  const int lane = t & 63, w = t >> 6;
  // workgroup 0, waves 0..3: the wave's own 64 x 64 diagonal block row, in flight under the mat-vec below
  double a[64];
  double rinv = 1.0;
  if (blockIdx.x == 0 && t < SB) {
    const int c0 = k0 + 64 * w;
#pragma unroll
    for (int c = 0; c < 64; ++c) a[c] = (c < lane && row < n) ? L[row + (long)(c0 + c) * ld] : 0.0;   // strictly lower part
    rinv = row < n ? 1.0 / L[row + (long)row * ld] : 1.0;
  }
  if (k0 > 0 && t < SB) xs[t] = y[kp + t];
  __syncthreads();
  double s = 0.0;
  if (k0 > 0 && row < n) {
    const double* Lr = L + row + (long)(kp + cg * 128) * ld;
    const double* xq = xs + cg * 128;
    // one CU streams this 256 x 256 block: what bounds it is bytes in flight -- 32 independent loads per lane
#pragma unroll
    for (int c8 = 0; c8 < 128; c8 += 32) {
      double v[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = Lr[(long)(c8 + c) * ld];
#pragma unroll
      for (int c = 0; c < 32; ++c) s += v[c] * xq[c8 + c];
    }
  }
  part[cg][rr] = s;
  __syncthreads();
  if (blockIdx.x > 0) {
    if (cg == 0 && row < n) r[row] -= part[0][rr] + part[1][rr];
    return;
  }
  double rt = 0.0;
  if (t < SB && row < n) rt = r[row] - (part[0][rr] + part[1][rr]);
  for (int sb = 0; sb < 4; ++sb) {
    const int c0 = k0 + 64 * sb;
    __syncthreads();                 // part[] free again
    if (w == sb) {
      // lane i ends with r_i - sum_{c<i} l_ic x_c (a[c] = 0 for c >= i: no compares in the chain); x_c = that / l_cc is
      // complete in lane c when step c broadcasts it
#pragma unroll
      for (int c = 0; c < 64; ++c) rt -= a[c] * readlane_f64(rt * rinv, c);
      rt *= rinv;
      xc[64 * sb + lane] = rt;
      if (row < n) y[row] = rt;
    }
    __syncthreads();
    // rows of the later sub-blocks: 32 columns per half
    double u = 0.0;
    if (rr >= 64 * (sb + 1) && row < n) {
      const double* Lr = L + row + (long)(c0 + cg * 32) * ld;
      const double* xq = xc + 64 * sb + cg * 32;
      double v[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = Lr[(long)c * ld];
#pragma unroll
      for (int c = 0; c < 32; ++c) u += v[c] * xq[c];
    }
    part[cg][rr] = u;
    __syncthreads();
    if (t < SB && rr >= 64 * (sb + 1)) rt -= part[0][rr] + part[1][rr];
  }
}

__device__ __forceinline__ double wave_allsum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sums of 8 per-lane values over the wavefront in 10 shuffles instead of 48: three exchange steps halve the values a
// lane carries (8 -> 4 -> 2 -> 1) while folding lane bits 5, 4, 3, three more fold bits 2, 1, 0.  Returns, in every lane,
// the wave-wide sum of p[(lane >> 3) & 7].
__device__ __forceinline__ double wave_sum8(const double (&p)[8], int lane) {
  double q4[4], q2[2];
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) q4[k] = (b5 ? p[k + 4] : p[k]) + __shfl_xor(b5 ? p[k] : p[k + 4], 32, 64);
#pragma unroll
  for (int k = 0; k < 2; ++k) q2[k] = (b4 ? q4[k + 2] : q4[k]) + __shfl_xor(b4 ? q4[k] : q4[k + 2], 16, 64);
  double v = (b3 ? q2[1] : q2[0]) + __shfl_xor(b3 ? q2[0] : q2[1], 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

__global__ __launch_bounds__(TRSV_T) void trsv_bwd_sb_kernel(const double* __restrict__ L, int ld, int n, int k0,
                                                             double* __restrict__ r, double* __restrict__ x) {
  __shared__ double xn[SB];          // x of the NEXT super-block (rows k0 + SB ...), final
  __shared__ double xc[SB];          // x of this one
  __shared__ double upd[SB];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;       // 8 wavefronts
  constexpr int NW = TRSV_T / 64;
  const int k1 = k0 + SB;
  const int nn = n - k1 < 0 ? 0 : (n - k1 > SB ? SB : n - k1);
  // workgroup 0, waves 0..3: lane k of wave s takes COLUMN k of diagonal block s (strictly lower part), so that
  // L' x = r is the same broadcast-and-subtract chain as the forward solve.  The loads are strided across the lanes
  // (each instruction touches 64 lines, all of which later instructions reuse from L1); issued first, they are in
  // flight under the dot products below.
  double b[64];
  double rinv = 1.0;
  if (blockIdx.x == 0 && t < SB) {
    const int col = k0 + t;
    const int c0 = k0 + 64 * w;
#pragma unroll
    for (int rw = 0; rw < 64; ++rw) b[rw] = (rw > lane && c0 + rw < n) ? L[(long)(c0 + rw) + (long)col * ld] : 0.0;
    rinv = col < n ? 1.0 / L[(long)col + (long)col * ld] : 1.0;
  }
  if (t < SB) xn[t] = t < nn ? x[k1 + t] : 0.0;
  __syncthreads();
  // dots of the columns col0 .. col0 + 7 (those < cend) with x of the next super-block, by one wavefront: the 32
  // loads of the batch are all issued before the first is used (the loads, not the arithmetic, are the time);
  // returns the dot of column col0 + j in the lanes 8 j .. 8 j + 7
  auto dots8 = [&](int col0, int cend) -> double {
    double v[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        v[j][q] = (col0 + j < cend && i < nn) ? L[(long)(k1 + i) + (long)(col0 + j) * ld] : 0.0;
      }
    double pj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pj[j] = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) pj[j] += v[j][q] * xn[lane + 64 * q];
    }
    return wave_sum8(pj, lane);
  };
  if (blockIdx.x > 0) {                                       // columns before this super-block, 64 per workgroup
    const int col0 = ((int)blockIdx.x - 1) * (8 * NW) + 8 * w;
    if (col0 < k0 && nn > 0) {
      const double d = dots8(col0, k0);
      if ((lane & 7) == 0 && col0 + (lane >> 3) < k0) r[col0 + (lane >> 3)] -= d;
    }
    return;
  }
  for (int j8 = 0; j8 < SB / NW; j8 += 8) {                   // the columns of this super-block, 32 per wavefront
    const int cc0 = w * (SB / NW) + j8;
    const double d = nn > 0 ? dots8(k0 + cc0, n) : 0.0;
    const int cj = cc0 + (lane >> 3);
    if ((lane & 7) == 0) upd[cj] = (k0 + cj < n ? r[k0 + cj] : 0.0) - d;
  }
  __syncthreads();
  double rt = t < SB ? upd[t] : 0.0;                          // thread t < 256 owns column k0 + t
  for (int sb = 3; sb >= 0; --sb) {
    const int c0 = k0 + 64 * sb;
    __syncthreads();                                          // upd[] read
    if (w == sb) {
#pragma unroll
      for (int c = 63; c >= 0; --c) rt -= b[c] * readlane_f64(rt * rinv, c);
      rt *= rinv;
      xc[64 * sb + lane] = rt;
      if (c0 + lane < n) x[c0 + lane] = rt;
    }
    __syncthreads();
    // earlier columns of the super-block: dot over the 64 rows of sub-block sb, one wavefront per column; a wave's
    // columns cc = w, w + 8, ... (at most 24): all loads first
    if (sb > 0) {
      const double xl = (c0 + lane < n) ? xc[64 * sb + lane] : 0.0;
      double v[24];
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        const int cc = w + NW * q;
        v[q] = (cc < 64 * sb && c0 + lane < n) ? L[(long)(c0 + lane) + (long)(k0 + cc) * ld] : 0.0;
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        if (w + NW * 8 * g < 64 * sb) {                       // wave-uniform: this group of 8 has live columns
          double pj[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pj[j] = v[8 * g + j] * xl;
          const double sq = wave_sum8(pj, lane);
          const int cc = w + NW * (8 * g + (lane >> 3));
          if ((lane & 7) == 0 && cc < 64 * sb) upd[cc] = sq;
        }
      }
    }
    __syncthreads();
    if (t < 64 * sb) rt -= upd[t];
  }
}

// x = L^{-T} L^{-1} h ; r, y: scratch (n doubles each); h is read-only.
int potrs_vec(hipStream_t st, const double* L, int n, int ld, const double* Linv, const double* h,
              double* x, double* r, double* y) {
  (void)Linv;
  if (n <= POTRS_SMALL) {
    hipLaunchKernelGGL(potrs_small_kernel, dim3(1), dim3(1024), 0, st, L, ld, n, Linv, h, x);
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  hipMemcpyAsync(r, h, (size_t)n * 8, hipMemcpyDeviceToDevice, st);
  const int nsb = (n + SB - 1) / SB;
  for (int b = 0; b < nsb; ++b) {
    const int k0 = b * SB;
    const int below = n - k0 - SB;                            // rows further down (they take x of super-block b - 1)
    const unsigned wgs = 1u + (b > 0 && below > 0 ? (unsigned)((below + SB - 1) / SB) : 0u);
    hipLaunchKernelGGL(trsv_fwd_sb_kernel, dim3(wgs), dim3(TRSV_T), 0, st, L, ld, n, k0, r, y);
  }
  for (int b = nsb - 1; b >= 0; --b) {
    const int k0 = b * SB;
    const bool have_next = k0 + SB < n;
    const unsigned wgs = 1u + (have_next && k0 > 0 ? (unsigned)((k0 + 63) / 64) : 0u);      // 64 columns per workgroup
    hipLaunchKernelGGL(trsv_bwd_sb_kernel, dim3(wgs), dim3(TRSV_T), 0, st, L, ld, n, k0, y, x);
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
    // X_b = op(L_kk)^-1 B_b: per right-hand side a row of the strip solve x L_kk' = b (x L_kk = b for the transposed
    // system); nrhs >= 16: by strips of 16 right-hand sides on the MFMA (round 3: the one-thread-per-column kernel walks
    // 2016 dependent FMA / LDS pairs, 53 us per block at nrhs = 801), else one thread per right-hand side
    static const bool trsm_thread = getenv("LRN_TRSM_THREAD") != nullptr;
    if (nrhs >= 16 && !trsm_thread)
      hipLaunchKernelGGL(potrf_panel_mfma_kernel, dim3((nrhs + 63) / 64), dim3(256), 0, st, B + k0, (long)ldb, 1L, nrhs,
                         L + (long)k0 + (long)k0 * ld, ld, nb, trans ? 1 : 0, tmp, (long)NB, 1L, (const int*)nullptr);
    else
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
