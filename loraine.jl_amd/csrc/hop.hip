// CG operator through the ASSEMBLED Schur matrix (round 4).
//
// MyA (reference src/Solvers.jl:582-614) applies  Ax = AA vec(W mat(AA'x) W)  matrix-free because a CPU cannot afford
// the nvar x nvar matrix H of src/makeBBBB.jl:67-218 for every IP iteration.  It is the same linear map: Ax = H x.
// On an MI355X H fits up to nvar ~ 1.5e5 and its assembly is cheap next to hundreds of operator applications (C5: 66 ms
// against 729 applications of 2.6 ms), so when the cost model below says so the operator of lrn_pcg / lrn_matvec is
//   assemble H once per NT scaling (the kit=0 kernels of schur.hip), then  y = H x  per CG iteration
// as ONE pass over the LOWER triangle: 4 nvar^2 bytes per application, the HBM roofline of the CG iteration.
//
// symv_lower_tiles_kernel: workgroup tile = 128 K rows x 128 columns of the lower triangle (column-major: a wave reads
// 1 KB of one column per load instruction, 16 B per lane).  Every element H[r,c], r >= c, is used twice while in
// registers: y[c] += H[r,c] x[r] (column sums: per-lane partials over the K row segments, one reduce-scatter butterfly
// per four columns) and, for r > c, y[r] += H[r,c] x[c] (row sums: registers across the tile's columns, the four waves
// combined through LDS).  Partial results go to slabs indexed by row block / column chunk and are added in a fixed
// order by symv_lower_reduce_kernel: no floating-point atomics, the result is bit-reproducible.
// One process per GPU: a rank applies the column chunks it owns (the block-cyclic ownership of the Schur columns,
// lrn_common.h::shard_owner) -- with column-block assembly these are the columns it assembled, no exchange of H at
// all -- and the nvar-vector is all-reduced as for the matrix-free operator.
#include <algorithm>
#include <cmath>

#include "../../include/loraine_hip.h"
#include "ctx.h"
#include "ops.h"

namespace lrn {

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double shx(double v, int mask) { return __shfl_xor(v, mask, 64); }

// VEC2: rows 2 lane, 2 lane + 1 of each 128-row segment (one 16-byte load; n even); else rows lane, lane + 64
template <int K, bool VEC2, bool MASK>
__device__ __forceinline__ void symv_tile_body(const double* __restrict__ H, int n, int r0, int cw0, int lane,
                                               const double* __restrict__ xs_w, const double (&xr)[K][2],
                                               double (&yr)[K][2], double* __restrict__ colout) {
  const int ra = VEC2 ? 2 * lane : lane, rb = VEC2 ? 2 * lane + 1 : lane + 64;
#pragma unroll 1
  for (int g = 0; g < 32; g += 4) {
    double h[K][4][2];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int rs = r0 + 128 * k;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cw0 + g + j;
        h[k][j][0] = 0.0;
        h[k][j][1] = 0.0;
        // (wave-uniform) skip of segments that lie entirely above the diagonal of this column
        if (c < n && (!MASK || rs + 127 >= c)) {
          const double* col = H + (size_t)c * n + rs;
          if (VEC2) {
            if (rs + ra < n) {
              const v2d v = *reinterpret_cast<const v2d*>(col + ra);
              h[k][j][0] = v.x;
              h[k][j][1] = v.y;
            }
          } else {
            if (rs + ra < n) h[k][j][0] = col[ra];
            if (rs + rb < n) h[k][j][1] = col[rb];
          }
        }
      }
    }
    double cs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cw0 + g + j;
      const double xc = xs_w[g + j];
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int rs = r0 + 128 * k;
        double a = h[k][j][0], b = h[k][j][1];
        if (MASK) {                      // the upper triangle is not authoritative (and may hold anything): select, not multiply
          a = (rs + ra >= c) ? a : 0.0;
          b = (rs + rb >= c) ? b : 0.0;
        }
        s += a * xr[k][0] + b * xr[k][1];
        if (MASK) {
          a = (rs + ra > c) ? a : 0.0;
          b = (rs + rb > c) ? b : 0.0;
        }
        yr[k][0] += a * xc;
        yr[k][1] += b * xc;
      }
      cs[j] = s;
    }
    // reduce-scatter over the 64 lanes: lanes [0,32) keep columns 0,1, lanes [32,64) columns 2,3; then by bit 4
    const bool hi5 = (lane & 32) != 0, hi4 = (lane & 16) != 0;
    double k0 = (hi5 ? cs[2] : cs[0]) + shx(hi5 ? cs[0] : cs[2], 32);
    double k1 = (hi5 ? cs[3] : cs[1]) + shx(hi5 ? cs[1] : cs[3], 32);
    double kk = (hi4 ? k1 : k0) + shx(hi4 ? k0 : k1, 16);
    kk += shx(kk, 8);
    kk += shx(kk, 4);
    kk += shx(kk, 2);
    kk += shx(kk, 1);
    if ((lane & 15) == 0) {
      const int c = cw0 + g + 2 * (lane >> 5) + ((lane >> 4) & 1);
      if (c < n) colout[c] = kk;
    }
  }
}

template <int K, bool VEC2>
__global__ __launch_bounds__(256) void symv_lower_tiles_kernel(const double* __restrict__ H, int n,
                                                               const int* __restrict__ idx, const double* __restrict__ x,
                                                               double* __restrict__ colpart, double* __restrict__ rowpart,
                                                               int rank, int world, int shard_bs) {
  constexpr int RB = 128 * K;
  const int J = blockIdx.x, I = blockIdx.y;
  const int r0 = I * RB, c0 = J * 128;
  if (c0 >= r0 + RB || c0 >= n || r0 >= n) return;            // tile entirely above the diagonal / outside
  if (world > 1 && shard_owner(c0 / shard_bs, world) != rank) return;
  __shared__ double xs[128];
  __shared__ double rsum[4][RB];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  if (t < 128) {
    const int cc = c0 + t;
    xs[t] = cc < n ? x[idx ? idx[cc] : cc] : 0.0;
  }
  const int ra = VEC2 ? 2 * lane : lane, rb = VEC2 ? 2 * lane + 1 : lane + 64;
  double xr[K][2], yr[K][2];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int a = r0 + 128 * k + ra, b = r0 + 128 * k + rb;
    xr[k][0] = a < n ? x[idx ? idx[a] : a] : 0.0;
    xr[k][1] = b < n ? x[idx ? idx[b] : b] : 0.0;
    yr[k][0] = 0.0;
    yr[k][1] = 0.0;
  }
  __syncthreads();
  double* colout = colpart + (size_t)I * n;
  if (c0 + 127 >= r0)
    symv_tile_body<K, VEC2, true>(H, n, r0, c0 + 32 * w, lane, xs + 32 * w, xr, yr, colout);
  else
    symv_tile_body<K, VEC2, false>(H, n, r0, c0 + 32 * w, lane, xs + 32 * w, xr, yr, colout);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    rsum[w][128 * k + ra] = yr[k][0];
    rsum[w][128 * k + rb] = yr[k][1];
  }
  __syncthreads();
  double* rowout = rowpart + (size_t)J * n;
  for (int i = t; i < RB; i += 256)
    if (r0 + i < n) rowout[r0 + i] = (rsum[0][i] + rsum[1][i]) + (rsum[2][i] + rsum[3][i]);
}

// y[idx[i]] = sum of the partials of row / column i in a fixed order: 64 outputs per workgroup, the slabs dealt to
// the four waves (slab s to wave s % 4), the four sums added through LDS.  qpart (may be null): per-workgroup partial
// sums of x[idx[i]] y[idx[i]] -- the p'Ap of the CG iteration without another pass over the vectors.
__global__ __launch_bounds__(256) void symv_lower_reduce_kernel(const double* __restrict__ colpart,
                                                                const double* __restrict__ rowpart, int n, int RB,
                                                                const int* __restrict__ idx, const double* __restrict__ x,
                                                                double* __restrict__ y, double* __restrict__ qpart,
                                                                int rank, int world, int shard_bs) {
  __shared__ double sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (i < n) {
    const int nrb = (n + RB - 1) / RB;
    const int J = i >> 7;
    const int I0 = (J * 128) / RB;
    const int ncol = (world <= 1 || shard_owner((J * 128) / shard_bs, world) == rank) ? nrb - I0 : 0;
    const int Jmax = min((n - 1) >> 7, ((i / RB) * RB + RB - 1) >> 7);
    // unified slab list: [0, ncol) column slabs I0 + k, then the row slabs 0 .. Jmax
    for (int k = w; k < ncol; k += 4) s += colpart[(size_t)(I0 + k) * n + i];
    for (int k = ((w - ncol) % 4 + 4) % 4; k <= Jmax; k += 4)
      if (world <= 1 || shard_owner((k * 128) / shard_bs, world) == rank) s += rowpart[(size_t)k * n + i];
  }
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0) {
    double v = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    double q = 0.0;
    if (i < n) {
      const int o = idx ? idx[i] : i;
      y[o] = v;
      q = x[o] * v;
    }
    if (qpart) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
      if (lane == 0) qpart[blockIdx.x] = q;
    }
  }
}

static bool hop_shardable(const lrn_ctx* c) {
  return c->pos_space && c->shard_bs > 0 && c->shard_bs % 128 == 0;
}

// y = H x (natural constraint order on both sides; H lives in sigma-position space when nlmi == 1).  world > 1: this
// rank's column chunks only, the caller all-reduces.
// y = A x for a symmetric n x n matrix of which the lower triangle is read (idx: permutation of both sides or null)
int symv_lower(lrn_ctx* c, const double* H, int n, const int* idx, const double* x, double* y, double* qpart, int* nq,
               bool sharded) {
  const bool small = n < 8192;
  const int K = small ? 1 : 4, RB = 128 * K;
  const int nrb = (n + RB - 1) / RB, nch = (n + 127) / 128;
  LRN_TRY(ensure(c, c->hopbuf, (size_t)(nrb + nch) * n * 8));
  double* colpart = c->hopbuf.as<double>();
  double* rowpart = colpart + (size_t)nrb * n;
  const int rank = sharded ? c->rank : 0, world = sharded ? c->world : 1;
  const dim3 grid(nch, nrb);
  const bool vec2 = (n & 1) == 0;
#define LRN_SYMV(KK, VV)                                                                                               \
  hipLaunchKernelGGL((symv_lower_tiles_kernel<KK, VV>), grid, dim3(256), 0, c->stream, H, n, idx, x, colpart, rowpart, \
                     rank, world, c->shard_bs)
  if (small) { if (vec2) LRN_SYMV(1, true); else LRN_SYMV(1, false); }
  else { if (vec2) LRN_SYMV(4, true); else LRN_SYMV(4, false); }
#undef LRN_SYMV
  // (sharded: the partial products x'y would be those of this rank's share -- the caller forms the dot after the all-reduce)
  double* qp = sharded ? nullptr : qpart;
  hipLaunchKernelGGL(symv_lower_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, c->stream, colpart, rowpart, n, RB,
                     idx, x, y, qp, rank, world, c->shard_bs);
  if (nq) *nq = qp ? (n + 63) / 64 : 0;
  return LRN_OK;
}

int hop_apply(lrn_ctx* c, const double* x, double* y, double* qpart, int* nq) {
  if (!c->have_H) return set_error(c, LRN_ERR_STATE, "no assembled H");
  c->counts["hop_matvec"] += 1;
  return symv_lower(c, c->H.as<double>(), c->nvar, c->pos_space ? c->lmi[0].sigma_d.as<int>() : nullptr, x, y, qpart, nq,
                    c->comm && c->world > 1);
}

// ------------------------------------------------------------------ cost model (static: the choice must not depend on
// measured times, or two runs of one problem would differ at rounding level)
static double est_operator_s(const lrn_ctx* c) {
  double t = 0.0;
  for (const auto& b : c->lmi) {
    const double m = b.msz;
    if (use_sparse_matvec(c, b))
      t += (b.msz < 1500 ? 30e-6 : 10e-6) + (double)b.ncq * m * 1.43e-12;      // pattern route (C5: 2.58 ms, C3: 36 us)
    else
      t += 4.0 * m * m * m / 5.0e13 + (double)b.nd * m * m * 8.0 / 6.0e12 + 40e-6;      // two products + two passes over the column tails
  }
  return t;
}

static double est_assemble_s(const lrn_ctx* c) {
  const double n = c->nvar;
  double t = n * n * 8.0 / 4.0e12 + 100e-6;
  for (const auto& b : c->lmi) {
    const double m = b.msz, nd = b.nd;
    if (b.nd > 0) t += (4.0 / 3.0 * nd * m * m * m + 0.5 * nd * nd * m * m + nd * (n - nd) * m * m * 0.5) / 6.0e13;
    double s1 = 0.0, s2 = 0.0;       // sum over sparse pairs of nnz_i nnz_j = ((sum nnz)^2 + sum nnz^2) / 2
    for (int p = b.nd; p < b.npos_nz; ++p) { s1 += (double)b.nnz[p]; s2 += (double)b.nnz[p] * (double)b.nnz[p]; }
    const double ns = b.npos_nz - b.nd;
    t += std::max(0.5 * (s1 * s1 + s2) * 4.0e-12, 0.5 * ns * (ns + 1.0) * 5.0e-11);      // C5: 2e8 pairs of 9 x 9 in 66 ms
  }
  return t;
}

static double est_symv_s(const lrn_ctx* c) {
  const double n = c->nvar;
  return n * n * 4.0 / 4.5e12 + 12e-6;
}

// Decide, once per NT scaling, whether lrn_pcg / lrn_matvec go through the assembled matrix.  expected_iters: operator
// applications the caller expects under this scaling (the CG iterations of the previous IP iteration).
bool hop_worthwhile(lrn_ctx* c, long expected_iters) {
  if (c->opt.matvec_h == 1 || c->nvar <= 0 || c->nlmi < 1) return false;
  if (c->world > 1 && (!c->comm || !hop_shardable(c))) return false;      // (lrn_set_shard alone: the caller exchanges H itself)
  for (const auto& b : c->lmi)
    if (!b.have_W) return false;
  if (c->opt.matvec_h == 2) return true;
  const double gain = (double)expected_iters * (est_operator_s(c) - est_symv_s(c));
  return gain > 1.2 * est_assemble_s(c);
}

// Make c->H the Schur matrix of the current scaling (general mode: the map MyA applies).  One process per GPU: every
// rank enters the status reduction; column blocks are NOT exchanged (each rank multiplies the columns it assembled),
// partial sums of the factor path are all-reduced as in lrn_schur_assemble.
int hop_prepare(lrn_ctx* c) {
  if (c->have_H && !c->H_shifted && !c->H_partial && c->H_version == c->scal_version && c->H_mode == 0) return LRN_OK;
  const bool sharded = c->comm && c->world > 1;
  int rc;
  if (sharded) {
    LRN_TRY(comm_agree_plan(c, 0));
    rc = schur_assemble(c, 0);
    rc = comm_schur_exchange(c, rc, /*gather_blocks=*/false);
  } else {
    rc = schur_assemble(c, 0);
  }
  if (rc == LRN_OK) c->counts["hop_assemble"] += 1;
  return rc;
}

}  // namespace lrn
