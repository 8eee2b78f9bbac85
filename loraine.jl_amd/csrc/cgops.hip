// CG operator, right-hand side, low-rank preconditioners and device-resident PCG.
// Replaces MyA / MyM_no / MyM_beta / MyM, Prec_for_CG_beta, Prec_for_CG_tilS_prep,
// prec_alpha_S! (reference src/Solvers.jl:572-904), makeRHS (src/makeBBBB.jl:221-228) and
// cg of ConjugateGradients.jl 0.1 (call sites src/predictor_corrector.jl:134,235).
//
// Mat-vec  Ax = AA vec(W mat(AA' x) W):  AA' x is a deterministic gather over the stored
// columns of AA (no atomics), the two msz^3 products run on the FP64 MFMA GEMM, AA vec(.)
// is one wavefront per constraint.  All vectors live in natural constraint order.
//
// H_alpha apply uses the algebraically identical "ts" form of the SMW formula:
//   ts = D^-1/2 AA (U (x) Z)   (nvar x k*msz, built once per IP iteration)
//   M^-1 x = D^-1/2 [ v - ts (I + ts' ts)^-1 ts' v ],  v = D^-1/2 x
// i.e. two bandwidth-bound GEMVs and one POTRS on the (k*msz)^2 factor -- this also covers
// erank > 1 without materialising kron(Umat, Z) (Solvers.jl:759 would need msz^2 x k*msz).
#include <algorithm>
#include <cmath>

#include "../../include/loraine_hip.h"
#include "ctx.h"
#include "jacobi.h"
#include "ops.h"

namespace lrn {

struct Prec {
  int kind = 0, erank = 0, ksz = 0;
  double dsum = 0.0;
  DBuf d;        // nvar
  DBuf ts;       // nvar x ksz
  DBuf cholS, linvS, cw;
  DBuf y, y2, y3, y4, zpart;
  DBuf E, Um, AU, sig;
  // H_alpha with linear constraints: AAAATtau = tau^2 I + C_lin diag(X_lin ./ S_lin) C_lin' is not
  // diagonal (Solvers.jl:743-745); its dense Cholesky factor L_D replaces the D^-1/2 scalings
  bool has_LD = false;
  DBuf LD, linvD, wD, Cd;
  // SMW core as an explicit inverse (option prec_inv): Sm = S + I, Ainv = (S + I)^-1 = L^-T L^-1
  bool has_inv = false;
  DBuf Sm, Ainv;
  // the whole preconditioner as one dense symmetric matrix (lower triangle), round 4:
  //   M^-1 = D^-1/2 (I - ts (S + I)^-1 ts') D^-1/2
  // -- inside lrn_pcg the seven launches of the SMW apply become one pass over nvar (nvar + 1) / 2 doubles (symv_lower)
  bool has_dense = false;
  DBuf Minv, T1;
};

// y = alpha M x + beta z for a symmetric n x n matrix, 16 rows per workgroup, x staged in LDS (n <= 8192): one launch
// (the solves through cholS are eight launches of 15-20 us at ksz = 801)
__global__ __launch_bounds__(256) void symv_rows_kernel(const double* __restrict__ M, int n, const double* __restrict__ x,
                                                        const double* __restrict__ z, double alpha, double beta,
                                                        double* __restrict__ y) {
  extern __shared__ double xs_[];
  __shared__ double sh_[16 * 16];
  const int t = threadIdx.x;
  for (int i = t; i < n; i += 256) xs_[i] = x[i];
  __syncthreads();
  const int r = t & 15, g = t >> 4;
  const int i = blockIdx.x * 16 + r;
  double acc = 0.0;
  if (i < n) {
    const double* mrow = M + i;
#pragma unroll 4
    for (int col = g; col < n; col += 16) acc += mrow[(size_t)col * n] * xs_[col];
  }
  sh_[g * 16 + r] = acc;
  __syncthreads();
  if (t < 16) {
    double v = 0.0;
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) v += sh_[gg * 16 + t];
    const int ii = blockIdx.x * 16 + t;
    if (ii < n) y[ii] = alpha * v + (z ? beta * z[ii] : 0.0);
  }
}

__global__ void eye_fill_kernel(double* __restrict__ V, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    V[e] = (e % n == e / n) ? 1.0 : 0.0;
}

__global__ void transpose_sq_kernel(const double* __restrict__ A, int n, double* __restrict__ B) {
  __shared__ double tile[32][33];
  int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    int i = bx + threadIdx.x, j = by + r;
    if (i < n && j < n) tile[r][threadIdx.x] = A[(long)i + (long)j * n];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    int i = by + threadIdx.x, j = bx + r;
    if (i < n && j < n) B[(long)i + (long)j * n] = tile[threadIdx.x][r];
  }
}

// lower triangle -> full symmetric (in place)
__global__ void mirror_lower_full_kernel(double* __restrict__ A, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    if (i < j) A[e] = A[(long)j + (long)i * n];
  }
}

void prec_free(lrn_ctx* c) {
  if (!c->prec) return;
  Prec* p = c->prec;
  for (DBuf* d : {&p->d, &p->ts, &p->cholS, &p->linvS, &p->cw, &p->y, &p->y2, &p->y3, &p->y4, &p->zpart, &p->E, &p->Um,
                  &p->AU, &p->sig, &p->LD, &p->linvD, &p->wD, &p->Cd, &p->Sm, &p->Ainv, &p->Minv, &p->T1})
    release(*d);
  delete p;
  c->prec = nullptr;
}

static inline unsigned nb(long n, long cap = 4096) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------ AA' x  and  AA vec(Z)
// M[q] = sum_k cq_v[k] * x[cq_j[k]] over the stored columns of AA (sparse constraints):
// one wavefront per stored column (a column can hold one entry per constraint -- e.g. the
// shared corner of thetaG11's 1600 edge blocks), fixed lane partition -> deterministic.
__global__ __launch_bounds__(256) void aat_gather_kernel(const long* __restrict__ cq_q, const long* __restrict__ cq_ptr,
                                                         const int* __restrict__ cq_j, const double* __restrict__ cq_v,
                                                         long ncq, const double* __restrict__ x, double* __restrict__ M) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ncq) return;
  double s = 0.0;
  for (long k = cq_ptr[t] + lane; k < cq_ptr[t + 1]; k += 64) s += cq_v[k] * x[cq_j[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) M[cq_q[t]] = s;
}

// M[q] -= sum_{p<nd} x[sigma[p]] * Adense[p][q]
__global__ void aat_dense_kernel(const double* __restrict__ Ad, int nd, long mm, const int* __restrict__ sigma,
                                 const double* __restrict__ x, double* __restrict__ M) {
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < mm; q += (long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int p = 0; p < nd; ++p) s += x[sigma[p]] * Ad[(long)p * mm + q];
    M[q] -= s;
  }
}

// mat(): (M + M')/2 in place  (kron_etc.jl:13-18)
__global__ void symmetrize_kernel(double* __restrict__ M, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % n), j = (int)(e / n);
    if (i < j) {
      double a = M[e], b = M[(long)j + (long)i * n];
      double s = (a + b) / 2.0;
      M[e] = s;
      M[(long)j + (long)i * n] = s;
    }
  }
}

// the same by 32 x 32 tile pairs (both accesses coalesced; the element-wise kernel above reads M' with stride n: 2 ms at
// msz 10^4 against 0.7): tile (bi, bj), bi >= bj, and its mirror image are averaged and written back together
__global__ __launch_bounds__(256) void symmetrize_tiled_kernel(double* __restrict__ M, int n) {
  __shared__ double ta[32][33], tb[32][33];
  const int nt = (n + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (long t = blockIdx.x; t < (long)nt * nt; t += gridDim.x) {
    const int ti = (int)(t % nt), tj = (int)(t / nt);
    if (ti < tj) continue;
    const int bi = ti * 32, bj = tj * 32;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi + tx, j = bj + r;       // tile (bi, bj): element (i, j) -> ta[r][tx]
      ta[r][tx] = (i < n && j < n) ? M[(long)i + (long)j * n] : 0.0;
      const int i2 = bj + tx, j2 = bi + r;     // tile (bj, bi): element (i2, j2) -> tb[r][tx]
      tb[r][tx] = (i2 < n && j2 < n) ? M[(long)i2 + (long)j2 * n] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi + tx, j = bj + r;
      if (i < n && j < n && i != j) {
        const double a = ta[r][tx], b = tb[tx][r];          // M[i,j], M[j,i]
        M[(long)i + (long)j * n] = (i > j ? a + b : b + a) / 2.0;        // (the pair adds in the order of the lower element first, as symmetrize_kernel)
      }
      if (ti != tj) {
        const int i2 = bj + tx, j2 = bi + r;
        if (i2 < n && j2 < n) {
          const double a = tb[r][tx], b = ta[tx][r];        // M[i2,j2] (upper), M[j2,i2] (lower)
          M[(long)i2 + (long)j2 * n] = (b + a) / 2.0;
        }
      }
    }
  }
}

static void symmetrize_dev(hipStream_t st, double* M, int n) {
  const long total = (long)n * n;
  if (n >= 512) {
    const long nt = (n + 31) / 32;
    hipLaunchKernelGGL(symmetrize_tiled_kernel, dim3((unsigned)std::min<long>(4096, nt * nt)), dim3(256), 0, st, M, n);
  } else {
    const long bl = (total + 255) / 256;
    hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)(bl < 1 ? 1 : (bl > 4096 ? 4096 : bl))), dim3(256), 0, st, M, n);
  }
}

// out[sigma[p]] += -sum_e a_e Z[r_e,c_e]   (one wavefront per sparse position)
__global__ __launch_bounds__(256) void aa_times_kernel(const long* __restrict__ ptr, const int* __restrict__ er,
                                                       const int* __restrict__ ec, const double* __restrict__ ev,
                                                       const double* __restrict__ Z, int msz, int p_lo, int p_end,
                                                       const int* __restrict__ sigma, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = p_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= p_end) return;
  double s = 0.0;
  for (long e = ptr[p] + lane; e < ptr[p + 1]; e += 64) s += ev[e] * Z[(long)er[e] + (long)ec[e] * msz];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[sigma[p]] -= s;
}

// out[sigma[p]] += -<Adense[p], Z>   (one workgroup per dense slot)
__global__ __launch_bounds__(256) void aa_dense_dot_kernel(const double* __restrict__ Ad, long mm,
                                                           const double* __restrict__ Z, const int* __restrict__ sigma,
                                                           double* __restrict__ out) {
  __shared__ double sh[4];
  const double* a = Ad + (long)blockIdx.x * mm;
  double s = 0.0;
  for (long q = threadIdx.x; q < mm; q += 256) s += a[q] * Z[q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[sigma[blockIdx.x]] -= sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- the passes over the dense constraint data at sizes where they are HBM streams (C4: 4000 matrices of 32 MB; round 3).
// The one-element-per-thread kernels above keep one 8-byte load per lane in flight and re-read Z from the MALL for every
// constraint (5.1 TB/s of constraint data, 25 ms per pass).  Here: 16-byte loads, eight of them in flight per lane, and FOUR
// constraints per workgroup against one read of Z.  msz even (16-byte alignment of every matrix); else the kernels above.
typedef double v2f64 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void aa_dense_dot4_kernel(const double* __restrict__ Ad, long mm, int nd,
                                                            const double* __restrict__ Z, const int* __restrict__ sigma,
                                                            double* __restrict__ out) {
  __shared__ double sh[4][4];
  const int p0 = blockIdx.x * 4;
  const long n2 = mm >> 1;
  const v2f64* z2 = reinterpret_cast<const v2f64*>(Z);
  const v2f64* a2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) a2[k] = reinterpret_cast<const v2f64*>(Ad + (long)(p0 + k < nd ? p0 + k : p0) * mm);
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  long q = threadIdx.x;
  for (; q + 256 < n2; q += 512) {
    const v2f64 z0 = z2[q], z1 = z2[q + 256];
    v2f64 a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a0[k] = __builtin_nontemporal_load(a2[k] + q); a1[k] = __builtin_nontemporal_load(a2[k] + q + 256); }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += (a0[k].x * z0.x + a0[k].y * z0.y) + (a1[k].x * z1.x + a1[k].y * z1.y);
  }
  for (; q < n2; q += 256) {
    const v2f64 z0 = z2[q];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const v2f64 a = a2[k][q]; s[k] += a.x * z0.x + a.y * z0.y; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 4 && p0 + (int)threadIdx.x < nd) {
    const int k = threadIdx.x;
    out[sigma[p0 + k]] -= sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
  }
}

// two products in one pass (aa_times2)
__global__ __launch_bounds__(256) void aa_dense_dot4x2_kernel(const double* __restrict__ Ad, long mm, int nd,
                                                              const double* __restrict__ Z1, const double* __restrict__ Z2,
                                                              const int* __restrict__ sigma, double* __restrict__ out1,
                                                              double* __restrict__ out2) {
  __shared__ double sh[8][4];
  const int p0 = blockIdx.x * 4;
  const long n2 = mm >> 1;
  const v2f64* y2 = reinterpret_cast<const v2f64*>(Z1);
  const v2f64* z2 = reinterpret_cast<const v2f64*>(Z2);
  const v2f64* a2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) a2[k] = reinterpret_cast<const v2f64*>(Ad + (long)(p0 + k < nd ? p0 + k : p0) * mm);
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (long q = threadIdx.x; q < n2; q += 256) {
    const v2f64 y0 = y2[q], z0 = z2[q];
    v2f64 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = a2[k][q];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] += a[k].x * y0.x + a[k].y * y0.y;
      s[4 + k] += a[k].x * z0.x + a[k].y * z0.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 8 && p0 + (int)(threadIdx.x & 3) < nd) {
    const int k = threadIdx.x;
    const double v = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
    if (k < 4) out1[sigma[p0 + k]] -= v;
    else out2[sigma[p0 + k - 4]] -= v;
  }
}

// M[q] -= sum_p x[sigma[p]] Adense[p][q], two entries per lane, eight matrices in flight
__global__ __launch_bounds__(256) void aat_dense2_kernel(const double* __restrict__ Ad, int nd, long mm,
                                                         const int* __restrict__ sigma, const double* __restrict__ x,
                                                         double* __restrict__ M) {
  const long n2 = mm >> 1;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= n2) return;
  const v2f64* a2 = reinterpret_cast<const v2f64*>(Ad) + q;
  v2f64 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
  int p = 0;
  for (; p + 8 <= nd; p += 8) {
    v2f64 a[8];
    double xs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = __builtin_nontemporal_load(a2 + (long)(p + k) * n2); xs[k] = x[sigma[p + k]]; }
#pragma unroll
    for (int k = 0; k < 8; k += 2) { s0 += xs[k] * a[k]; s1 += xs[k + 1] * a[k + 1]; }
  }
  // (rotating the order of the matrices per group of workgroups -- eight addresses 32 MB apart meet the same DRAM banks --
  // changes nothing: 23.5 ms per pass = 5.4 TB/s either way)
  for (; p < nd; ++p) s0 += x[sigma[p]] * a2[(long)p * n2];
  v2f64* m2 = reinterpret_cast<v2f64*>(M) + q;
  *m2 -= s0 + s1;
}

// ---- round 4: the same passes over HALF the bytes.  Every A_k is symmetric, so <A_k, Z> = sum_{i>=j} A_k[i,j] w[i,j] with
// w = Z + Z' below the diagonal, Z on it, and mat(AA'x) is the mirror image of its lower triangle.  Only the column tails
// rows >= j of each matrix are streamed (they are contiguous in the column-major storage), cut into chunks of 64 x 16 bytes
// = 128 rows listed in a table built once per matrix side (TriChunk; the tails start at j rounded down to 16 rows so that
// every chunk is 128-byte aligned: 0.8 % of over-read at msz 2000, covered by zero weights / masked stores).  One wave per
// chunk; the workgroup's four chunks are consecutive in the table, i.e. mostly one contiguous 4 KB piece of a column.
struct TriChunk { long off; int col; int nv2; };      // first element (doubles), column, valid 16-byte pairs (<= 64)

// w = tri-weights of Z: Z + Z' strictly below the diagonal, Z on it, zero above
__global__ void tri_weights_kernel(const double* __restrict__ Z, int n, double* __restrict__ Wt) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // output tile rows bx.., columns by..
  if (bx + 31 < by) {                                          // entirely above the diagonal
    for (int r = threadIdx.y; r < 32; r += 8) {
      const int i = bx + threadIdx.x, j = by + r;
      if (i < n && j < n) Wt[(long)i + (long)j * n] = 0.0;
    }
    return;
  }
  for (int r = threadIdx.y; r < 32; r += 8) {                  // tile[r][t] = Z[by + t, bx + r]  (the transposed block)
    const int i = by + threadIdx.x, j = bx + r;
    if (i < n && j < n) tile[r][threadIdx.x] = Z[(long)i + (long)j * n];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = bx + threadIdx.x, j = by + r;
    if (i < n && j < n) {
      const double a = Z[(long)i + (long)j * n];
      Wt[(long)i + (long)j * n] = i > j ? a + tile[threadIdx.x][r] : (i == j ? a : 0.0);
    }
  }
}

// upper triangle <- lower triangle
__global__ void mirror_lower_tiled_kernel(double* __restrict__ A, int n) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // source tile rows bx.., columns by.. (on or below the diagonal)
  if (bx < by) return;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = bx + threadIdx.x, j = by + r;
    if (i < n && j < n) tile[r][threadIdx.x] = A[(long)i + (long)j * n];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = by + threadIdx.x, j = bx + r;              // destination (i, j) = transposed position
    if (i < n && j < n && i < j) A[(long)i + (long)j * n] = tile[threadIdx.x][r];
  }
}

// out[sigma[p]] -= <A_p, Z> for NZ weight matrices at once (NZ = 1, 2), four constraints per workgroup
template <int NZ>
__global__ __launch_bounds__(256) void aa_dense_tri_dot4_kernel(const double* __restrict__ Ad, long mm, int nd,
                                                                const double* __restrict__ W1, const double* __restrict__ W2,
                                                                const TriChunk* __restrict__ tab, int nch,
                                                                const int* __restrict__ sigma, double* __restrict__ out1,
                                                                double* __restrict__ out2) {
  __shared__ double sh[4 * NZ][4];
  const int p0 = blockIdx.x * 4;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const v2f64* a2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) a2[k] = reinterpret_cast<const v2f64*>(Ad + (long)(p0 + k < nd ? p0 + k : p0) * mm);
  const v2f64* w1 = reinterpret_cast<const v2f64*>(W1);
  const v2f64* w2 = reinterpret_cast<const v2f64*>(NZ > 1 ? W2 : W1);
  double s[4 * NZ];
#pragma unroll
  for (int k = 0; k < 4 * NZ; ++k) s[k] = 0.0;
  for (int ch = w; ch < nch; ch += 8) {
    const TriChunk e0 = tab[ch];
    const bool two = ch + 4 < nch;
    const TriChunk e1 = tab[two ? ch + 4 : ch];
    const bool v0 = lane < e0.nv2, v1 = two && lane < e1.nv2;
    const long q0 = (e0.off >> 1) + (v0 ? lane : 0), q1 = (e1.off >> 1) + (v1 ? lane : 0);
    v2f64 a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a0[k] = __builtin_nontemporal_load(a2[k] + q0); a1[k] = __builtin_nontemporal_load(a2[k] + q1); }
    v2f64 y0 = w1[q0], y1 = w1[q1];
    if (!v0) y0 = (v2f64){0.0, 0.0};
    if (!v1) y1 = (v2f64){0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += (a0[k].x * y0.x + a0[k].y * y0.y) + (a1[k].x * y1.x + a1[k].y * y1.y);
    if (NZ > 1) {
      v2f64 z0 = w2[q0], z1 = w2[q1];
      if (!v0) z0 = (v2f64){0.0, 0.0};
      if (!v1) z1 = (v2f64){0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 4; ++k) s[4 + k] += (a0[k].x * z0.x + a0[k].y * z0.y) + (a1[k].x * z1.x + a1[k].y * z1.y);
    }
  }
#pragma unroll
  for (int k = 0; k < 4 * NZ; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
    if (lane == 0) sh[k][w] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 4 * NZ && p0 + (int)(threadIdx.x & 3) < nd) {
    const int k = threadIdx.x;
    const double v = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
    if (k < 4) out1[sigma[p0 + k]] -= v;
    else out2[sigma[p0 + k - 4]] -= v;
  }
}

// lower triangle of M -= sum_p x[sigma[p]] A_p: one wave per chunk, eight matrices in flight per lane
__global__ __launch_bounds__(256) void aat_dense_tri_kernel(const double* __restrict__ Ad, int nd, long mm, int m,
                                                            const int* __restrict__ sigma, const double* __restrict__ x,
                                                            const TriChunk* __restrict__ tab, int nch, double* __restrict__ M) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (ch >= nch) return;
  const TriChunk e = tab[ch];
  if (lane >= e.nv2) return;
  const long n2 = mm >> 1;
  const v2f64* a2 = reinterpret_cast<const v2f64*>(Ad) + (e.off >> 1) + lane;
  v2f64 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
  int p = 0;
  for (; p + 8 <= nd; p += 8) {
    v2f64 a[8];
    double xs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = __builtin_nontemporal_load(a2 + (long)(p + k) * n2); xs[k] = x[sigma[p + k]]; }
#pragma unroll
    for (int k = 0; k < 8; k += 2) { s0 += xs[k] * a[k]; s1 += xs[k + 1] * a[k + 1]; }
  }
  for (; p < nd; ++p) s0 += x[sigma[p]] * a2[(long)p * n2];
  const v2f64 t = s0 + s1;
  double* mp = M + e.off + 2 * lane;
  const int r = (int)(e.off - (long)e.col * m) + 2 * lane;      // row of the first element of the pair
  if (r >= e.col) mp[0] -= t.x;
  if (r + 1 >= e.col) mp[1] -= t.y;
}

// is every dense constraint matrix symmetric?  flag[0] = 1 when a pair differs
__global__ __launch_bounds__(256) void dense_sym_check_kernel(const double* __restrict__ Ad, int m, int* __restrict__ flag) {
  const double* A = Ad + (long)blockIdx.x * m * m;
  bool bad = false;
  for (long e = threadIdx.x; e < (long)m * m; e += 256) {
    const int i = (int)(e % m), j = (int)(e / m);
    if (i > j && A[e] != A[(long)j + (long)i * m]) bad = true;
  }
  if (bad) flag[0] = 1;
}

// row-sharded variants (multi-GPU mat-vec): only entries with r0 <= row < r1; Zg holds the rows
// [r0,r1) of Z with leading dimension ldz
__global__ __launch_bounds__(256) void aa_times_rows_kernel(const long* __restrict__ ptr, const int* __restrict__ er,
                                                            const int* __restrict__ ec, const double* __restrict__ ev,
                                                            const double* __restrict__ Zg, int ldz, int r0, int r1,
                                                            int p_lo, int p_end, const int* __restrict__ sigma,
                                                            double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = p_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= p_end) return;
  double s = 0.0;
  for (long e = ptr[p] + lane; e < ptr[p + 1]; e += 64) {
    int r = er[e];
    if (r >= r0 && r < r1) s += ev[e] * Zg[(long)(r - r0) + (long)ec[e] * ldz];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[sigma[p]] -= s;
}

__global__ __launch_bounds__(256) void aa_dense_dot_rows_kernel(const double* __restrict__ Ad, int m,
                                                                const double* __restrict__ Zg, int ldz, int r0, int r1,
                                                                const int* __restrict__ sigma, double* __restrict__ out) {
  __shared__ double sh[4];
  const double* a = Ad + (long)blockIdx.x * m * m;
  const int nr = r1 - r0;
  double s = 0.0;
  for (long q = threadIdx.x; q < (long)nr * m; q += 256) {
    int r = (int)(q % nr), cc = (int)(q / nr);
    s += a[(long)(r0 + r) + (long)cc * m] * Zg[(long)r + (long)cc * ldz];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[sigma[blockIdx.x]] -= sh[0] + sh[1] + sh[2] + sh[3];
}

// linear block, two deterministic passes (no floating-point atomics):
//   t_l = xs_l * sum_i C[i,l] x_i   (by column);   y_i += sum_l C[i,l] t_l   (by row, CSR built at upload)
__global__ void lin_t_kernel(const long* __restrict__ ptr, const int* __restrict__ row, const double* __restrict__ val,
                             const double* __restrict__ xs, int nlin, const double* __restrict__ x,
                             double* __restrict__ tl) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlin) return;
  double t = 0.0;
  for (long k = ptr[l]; k < ptr[l + 1]; ++k) t += val[k] * x[row[k]];
  tl[l] = t * xs[l];
}

__global__ void lin_rows_kernel(const long* __restrict__ rptr, const int* __restrict__ rcol,
                                const double* __restrict__ rval, const double* __restrict__ tl, int n,
                                double* __restrict__ y) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (long k = rptr[i]; k < rptr[i + 1]; ++k) s += rval[k] * tl[rcol[k]];
  y[i] += s;
}

// d_i += sum_l C[i,l]^2 xs_l
__global__ void lin_diag_kernel(const long* __restrict__ rptr, const int* __restrict__ rcol,
                                const double* __restrict__ rval, const double* __restrict__ xs, int n,
                                double* __restrict__ d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (long k = rptr[i]; k < rptr[i + 1]; ++k) s += rval[k] * rval[k] * xs[rcol[k]];
  d[i] += s;
}

static int lin_matvec(lrn_ctx* c, const double* x, double* y) {
  LRN_TRY(ensure(c, c->redbuf, (size_t)std::max(c->nlin, 64) * 8));
  double* tl = c->redbuf.as<double>();
  const unsigned gl = (unsigned)((c->nlin + 255) / 256), gn = (unsigned)((c->nvar + 255) / 256);
  hipLaunchKernelGGL(lin_t_kernel, dim3(gl), dim3(256), 0, c->stream, c->cl_ptr.as<long>(), c->cl_rown.as<int>(),
                     c->cl_val.as<double>(), c->lin_xs.as<double>(), c->nlin, x, tl);
  hipLaunchKernelGGL(lin_rows_kernel, dim3(gn), dim3(256), 0, c->stream, c->cr_ptr.as<long>(), c->cr_col.as<int>(),
                     c->cr_val.as<double>(), tl, c->nvar, y);
  return LRN_OK;
}


// ------------------------------------------------------------------ sparse-aware mat-vec
// When M = mat(AA'x) is sparse (every constraint sparse, e.g. C5: 9 nnz each, 18 per column of M),
// AA vec(W M W) needs Z = W M W only on the pattern of M.  With N = M W  (N(:,q) = M W(:,q)):
//   Z[p,q] = W(:,p) . N(:,q)        -- two contiguous columns
// and N' = W M is a sparse combination of columns of W:  N'(:,r) = sum_s M[s,r] W(:,s).
// 2 nnz(M) msz + nnz(M) msz flop instead of 4 msz^3; the kernels are bandwidth-bound (L2 / MALL).

__global__ __launch_bounds__(256) void sp_gather_kernel(const long* __restrict__ cq_ptr, const int* __restrict__ cq_j,
                                                        const double* __restrict__ cq_v, long ncq,
                                                        const double* __restrict__ x, double* __restrict__ raw) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ncq) return;
  double s = 0.0;
  for (long k = cq_ptr[t] + lane; k < cq_ptr[t + 1]; k += 64) s += cq_v[k] * x[cq_j[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) raw[t] = s;
}

// mat(): (M + M')/2 on the pattern  (kron_etc.jl:13-18)
__global__ void sp_symmetrize_kernel(const double* __restrict__ raw, const int* __restrict__ pc_t, long ncq,
                                     double* __restrict__ Mv) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ncq) Mv[t] = (raw[t] + raw[pc_t[t]]) / 2.0;
}

// N[r, q] = sum_{t in column r of M} Mv[t] * W[q, row(t)]   for q in [q_lo, q_hi), all r.
// One thread per q, 16 consecutive r per workgroup: W is read along q (coalesced 2 KB per stored
// entry), N is written as 128 contiguous bytes per thread.  blockIdx.x walks r (fast) so that the
// 256 rows of W a q-tile touches (msz * 2 KB) stay in L2 / MALL across the r-tiles.
__global__ __launch_bounds__(256) void sp_wm_kernel(const long* __restrict__ pc_ptr, const int* __restrict__ pc_r,
                                                    const double* __restrict__ Mv, const double* __restrict__ W, int m,
                                                    int q_lo, int q_hi, double* __restrict__ N) {
  const int q = q_lo + blockIdx.y * 256 + threadIdx.x;
  const int r0 = blockIdx.x * 16;
  const bool live = q < q_hi;
  const double* wq = W + (live ? q : q_lo);
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + i;
    if (r < m) {
      const long t1 = pc_ptr[r + 1];
      for (long t = pc_ptr[r]; t < t1; ++t) acc[i] += Mv[t] * wq[(long)pc_r[t] * m];
    }
  }
  if (!live) return;
  double* dst = N + (long)q * m + r0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (r0 + i < m) dst[i] = acc[i];
}

// The same product with the q-tiles dealt to the XCDs (round 3).  Above, blockIdx.x walks the row tiles, so the eight XCDs
// work on the SAME 256 columns of W at a time and each of their L2s misses on the whole 20 MB slab (msz 10^4): 10.8 GB of
// L2-miss traffic per mat-vec for 0.8 GB of W.  Here XCD x = blockIdx.x % 8 sweeps all row tiles of q-tile 8 (j / R) + x,
// j = blockIdx.x / 8, 64 columns wide: the 64 x msz slab (5 MB) it gathers from -- every row 36 times at C5 -- stays in its
// own L2.  One wave per 4 rows r, lanes over q (512-byte segments of W).
__global__ __launch_bounds__(256) void sp_wm_xcd_kernel(const long* __restrict__ pc_ptr, const int* __restrict__ pc_r,
                                                        const double* __restrict__ Mv, const double* __restrict__ W, int m,
                                                        int q_lo, int q_hi, double* __restrict__ N) {
  const int R = (m + 15) / 16, Q = (q_hi - q_lo + 63) / 64;
  const int xcd = blockIdx.x & 7;
  const long j = blockIdx.x >> 3;
  const int qt = 8 * (int)(j / R) + xcd, rt = (int)(j % R);
  if (qt >= Q) return;
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = q_lo + qt * 64 + lane;
  const bool live = q < q_hi;
  const double* wq = W + (live ? q : q_lo);
  const int r0 = rt * 16 + 4 * g;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + i;
    if (r >= m) break;
    const long t0 = pc_ptr[r], t1 = pc_ptr[r + 1];
    for (long t = t0; t < t1; t += 4) {
      double w[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = t + k < t1;
        v[k] = ok ? Mv[t + k] : 0.0;
        w[k] = wq[(long)pc_r[ok ? t + k : t] * m];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[i] += v[k] * w[k];
    }
  }
  if (!live) return;
  double* dst = N + (long)q * m + r0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (r0 + i < m) dst[i] = acc[i];
}

// Zs[t] = W(:,p_t) . N(:,q_t) for the stored entries of pattern column q (one workgroup per column);
// mirror: only p <= q is computed and copied to the transposed entry (Z is symmetric).
__global__ __launch_bounds__(256) void sp_dot_kernel(const long* __restrict__ pc_ptr, const int* __restrict__ pc_r,
                                                     const int* __restrict__ pc_t, const double* __restrict__ W,
                                                     const double* __restrict__ N, int m, int q_lo, int mirror,
                                                     double* __restrict__ Zs) {
  __shared__ double sh[4];
  const int q = q_lo + blockIdx.x;
  const double* nq = N + (long)q * m;
  const long t1 = pc_ptr[q + 1];
  for (long t = pc_ptr[q]; t < t1; ++t) {
    const int p = pc_r[t];
    if (mirror && p > q) break;                 // rows ascending within a column
    const double* wp = W + (long)p * m;
    double s = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) s += wp[i] * nq[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = sh[0] + sh[1] + sh[2] + sh[3];
      Zs[t] = v;
      if (mirror && p != q) Zs[pc_t[t]] = v;
    }
  }
}

// ---- the same operator for matrices that sit in L2 (msz < 1500: thetaG11, msz 801, 3 % of the entries of M stored) --
// round 3.  There the dense route costs two msz^3 products and three passes over msz^2 (97 us per mat-vec at msz 801)
// for 2 x 2 nnz(M) msz flop of useful work, and the kernels above -- one workgroup per column with a block reduction per
// entry -- are bound by their serial loops.  Here: the gather and the symmetrisation in one launch (one thread per stored
// entry, both halves of (M + M') / 2), and ONE WAVE per stored entry for Z[p, q] = W(:, p) . N(:, q).
__global__ __launch_bounds__(256) void sp_gather_sym_kernel(const long* __restrict__ cq_ptr, const int* __restrict__ cq_j,
                                                            const double* __restrict__ cq_v, const int* __restrict__ pc_t,
                                                            long ncq, const double* __restrict__ x, double* __restrict__ Mv) {
  // one wave per stored entry (a position can be shared by every constraint: thetaG11's corner entry by 1600)
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ncq) return;
  const long u = pc_t[t];
  double a = 0.0, b = 0.0;
  for (long k = cq_ptr[t] + lane; k < cq_ptr[t + 1]; k += 64) a += cq_v[k] * x[cq_j[k]];
  if (u != t)
    for (long k = cq_ptr[u] + lane; k < cq_ptr[u + 1]; k += 64) b += cq_v[k] * x[cq_j[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
  // (raw[t] + raw[twin]) / 2 in a fixed order: the twins get the same bits
  if (lane == 0) Mv[t] = (u == t) ? (a + a) / 2.0 : (t < u ? a + b : b + a) / 2.0;
}

// N[r, q] for 4 consecutive r per workgroup, one thread per q; the W values of up to 8 stored entries are requested
// before they are used (one workgroup per CU at this size: nothing else hides the L2 latency).  Columns with more than
// SP_LONG stored entries (thetaG11: one column of 801 among columns of 7) are left to sp_wm_long_kernel.
static constexpr int SP_LONG = 64;
__global__ __launch_bounds__(256) void sp_wm_small_kernel(const long* __restrict__ pc_ptr, const int* __restrict__ pc_r,
                                                          const double* __restrict__ Mv, const double* __restrict__ W, int m,
                                                          int q_lo, int q_hi, double* __restrict__ N) {
  const int q = q_lo + blockIdx.y * 256 + threadIdx.x;
  const int r0 = blockIdx.x * 4;
  const bool live = q < q_hi;
  const double* wq = W + (live ? q : q_lo);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + i;
    if (r >= m) break;
    const long t0 = pc_ptr[r], t1 = pc_ptr[r + 1];
    if (t1 - t0 > SP_LONG) continue;
    for (long t = t0; t < t1; t += 8) {
      double w[8], v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool ok = t + k < t1;
        v[k] = ok ? Mv[t + k] : 0.0;
        w[k] = wq[(long)pc_r[ok ? t + k : t] * m];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[i] += v[k] * w[k];
    }
  }
  if (!live) return;
  double* dst = N + (long)q * m + r0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (r0 + i < m) {
      const long cnt = pc_ptr[r0 + i + 1] - pc_ptr[r0 + i];
      if (cnt <= SP_LONG) dst[i] = acc[i];
    }
}

// N[r, q] of a long column r: one wave per q, the lanes over the stored entries (W(:, q) read at the rows of the
// entries -- ascending, nearly contiguous)
__global__ __launch_bounds__(256) void sp_wm_long_kernel(const long* __restrict__ pc_ptr, const int* __restrict__ pc_r,
                                                         const double* __restrict__ Mv, const double* __restrict__ W, int m,
                                                         int r, int q_lo, int q_hi, double* __restrict__ N) {
  const int lane = threadIdx.x & 63;
  const int q = q_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= q_hi) return;
  const double* wq = W + (long)q * m;
  double s = 0.0;
  for (long t = pc_ptr[r] + lane; t < pc_ptr[r + 1]; t += 64) s += Mv[t] * wq[pc_r[t]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) N[(long)q * m + r] = s;
}

__global__ __launch_bounds__(256) void sp_dot_wave_kernel(const long* __restrict__ cq_q, const int* __restrict__ pc_t, long ncq,
                                                          const double* __restrict__ W, const double* __restrict__ N, int m,
                                                          int q_lo, int q_hi, int mirror, double* __restrict__ Zs) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ncq) return;
  const long key = cq_q[t];
  const int q = (int)(key / m), p = (int)(key % m);
  if (q < q_lo || q >= q_hi || (mirror && p > q)) return;
  const double* wp = W + (long)p * m;
  const double* nq = N + (long)q * m;
  double s = 0.0;
  for (int i = lane; i < m; i += 64) s += wp[i] * nq[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) {
    Zs[t] = s;
    if (mirror && p != q) Zs[pc_t[t]] = s;
  }
}

// out[sigma[p]] -= sum_e a_e Zs[ent_t[e]]   (entries with column in [c_lo, c_hi))
__global__ __launch_bounds__(256) void sp_aa_times_kernel(const long* __restrict__ ptr, const int* __restrict__ ec,
                                                          const double* __restrict__ ev, const int* __restrict__ ent_t,
                                                          const double* __restrict__ Zs, int c_lo, int c_hi, int p_end,
                                                          const int* __restrict__ sigma, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= p_end) return;
  double s = 0.0;
  for (long e = ptr[p] + lane; e < ptr[p + 1]; e += 64) {
    const int cc = ec[e];
    if (cc >= c_lo && cc < c_hi) s += ev[e] * Zs[ent_t[e]];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[sigma[p]] -= s;
}

// The half-traffic passes (column tails of the symmetric matrices): table of chunks, built once per block
static int tri_table(lrn_ctx* c, LmiBlock& b) {
  if (b.tri_nch > 0) return LRN_OK;
  const int m = b.msz;
  std::vector<TriChunk> tab;
  for (int j = 0; j < m; ++j) {
    const int r0 = j & ~15;
    for (int r = r0; r < m; r += 128) tab.push_back({(long)j * m + r, j, std::min(64, (m - r) / 2)});
  }
  LRN_TRY(ensure(c, b.tri_tab, tab.size() * sizeof(TriChunk)));
  LRN_TRY(copy_in(c, b.tri_tab.p, tab.data(), tab.size() * sizeof(TriChunk)));
  b.tri_nch = (int)tab.size();
  return LRN_OK;
}

// symmetric dense data (checked once on the device), msz even (16-byte pairs), enough data to be a stream
static bool dense_tri_ok(lrn_ctx* c, LmiBlock& b) {
  static const bool off = getenv("LRN_DENSE_PASS_FULL") != nullptr;       // (measurement: the round-3 kernels, both triangles)
  if (off || b.nd <= 0 || (b.msz & 1) != 0 || b.msz < 256 || ((uintptr_t)b.Adense.p & 15) != 0) return false;
  if (b.dense_sym < 0) {
    if (ensure(c, c->info_dev, 64) != LRN_OK) return false;
    (void)hipMemsetAsync(c->info_dev.p, 0, 4, c->stream);
    hipLaunchKernelGGL(dense_sym_check_kernel, dim3(b.nd), dim3(256), 0, c->stream, b.Adense.as<double>(), b.msz,
                       c->info_dev.as<int>());
    int f = 1;
    if (copy_out(c, &f, c->info_dev.p, 4) != LRN_OK) return false;
    (void)hipMemsetAsync(c->info_dev.p, 0, 4, c->stream);
    b.dense_sym = f == 0 ? 1 : 0;
  }
  return b.dense_sym == 1 && tri_table(c, b) == LRN_OK;
}

// w = tri-weights of Z into slot `which` of the block's weight workspace
static int tri_weights(lrn_ctx* c, LmiBlock& b, const double* Z, int which, double** out) {
  const int m = b.msz;
  LRN_TRY(ensure(c, c->triw, (size_t)2 * m * m * 8));
  double* Wt = c->triw.as<double>() + (size_t)which * m * m;
  hipLaunchKernelGGL(tri_weights_kernel, dim3((m + 31) / 32, (m + 31) / 32), dim3(32, 8), 0, c->stream, Z, m, Wt);
  *out = Wt;
  return LRN_OK;
}

// the 16-byte kernels of the dense passes: every matrix 16-byte aligned (msz even), enough data to be a stream
static bool dense_stream_ok(const LmiBlock& b, const double* Z) {
  static const bool off = getenv("LRN_DENSE_PASS_SCALAR") != nullptr;      // (measurement: the one-element-per-lane kernels)
  return !off && (b.msz & 1) == 0 && b.msz >= 256 && (((uintptr_t)Z | (uintptr_t)b.Adense.p) & 15) == 0;
}

bool use_sparse_matvec(const lrn_ctx* c, const LmiBlock& b) {
  if (!b.sp_ok || c->opt.matvec_sparse == 1) return false;
  if (c->opt.matvec_sparse == 2) return true;
  // ~4e-12 ncq msz s against 4 msz^3 / 6e13 s.  Below msz ~ 1500 both routes are bound by their launches and L2: the
  // wave-per-entry kernels win where a twelfth of M or less is stored
  if (b.msz < 1500)
    return b.msz >= 256 && (double)b.ncq * 12.0 < (double)b.msz * (double)b.msz && b.sp_long_cols.size() <= 4;
  return (double)b.ncq * 60.0 < (double)b.msz * (double)b.msz;
}

// y += AA vec(W mat(AA'x) W) restricted to the pattern columns [q_lo, q_hi) of Z
static int matvec_sparse_block(lrn_ctx* c, LmiBlock& b, const double* x, double* y, int q_lo, int q_hi, bool mirror) {
  const int m = b.msz;
  hipStream_t st = c->stream;
  LRN_TRY(ensure(c, c->m1, (size_t)m * m * 8));
  double* N = c->m1.as<double>();
  const bool small = m < 1500;        // W and N sit in L2: one wave per stored entry (see sp_dot_wave_kernel)
  if (small) {
    hipLaunchKernelGGL(sp_gather_sym_kernel, dim3((unsigned)((b.ncq + 3) / 4)), dim3(256), 0, st, b.cq_ptr.as<long>(),
                       b.cq_j.as<int>(), b.cq_v.as<double>(), b.pc_t.as<int>(), b.ncq, x, b.Mv.as<double>());
  } else {
    hipLaunchKernelGGL(sp_gather_kernel, dim3((unsigned)((b.ncq + 3) / 4)), dim3(256), 0, st, b.cq_ptr.as<long>(),
                       b.cq_j.as<int>(), b.cq_v.as<double>(), b.ncq, x, b.Zs.as<double>());
    hipLaunchKernelGGL(sp_symmetrize_kernel, dim3((unsigned)((b.ncq + 255) / 256)), dim3(256), 0, st, b.Zs.as<double>(),
                       b.pc_t.as<int>(), b.ncq, b.Mv.as<double>());
  }
  if (q_hi > q_lo) {
    if (small) {
      hipLaunchKernelGGL(sp_wm_small_kernel, dim3((m + 3) / 4, (q_hi - q_lo + 255) / 256), dim3(256), 0, st,
                         b.pc_ptr.as<long>(), b.pc_r.as<int>(), b.Mv.as<double>(), b.W.as<double>(), m, q_lo, q_hi, N);
      for (int r : b.sp_long_cols)
        hipLaunchKernelGGL(sp_wm_long_kernel, dim3((q_hi - q_lo + 3) / 4), dim3(256), 0, st, b.pc_ptr.as<long>(),
                           b.pc_r.as<int>(), b.Mv.as<double>(), b.W.as<double>(), m, r, q_lo, q_hi, N);
    } else {
      static const bool wm_plain = getenv("LRN_SP_WM_PLAIN") != nullptr;      // (measurement: the round-1 tiling)
      const long R = (m + 15) / 16, Q8 = ((q_hi - q_lo + 63) / 64 + 7) / 8;
      if (!wm_plain && 8 * Q8 * R < 0x7fffffffL)
        hipLaunchKernelGGL(sp_wm_xcd_kernel, dim3((unsigned)(8 * Q8 * R)), dim3(256), 0, st, b.pc_ptr.as<long>(),
                           b.pc_r.as<int>(), b.Mv.as<double>(), b.W.as<double>(), m, q_lo, q_hi, N);
      else
      hipLaunchKernelGGL(sp_wm_kernel, dim3((m + 15) / 16, (q_hi - q_lo + 255) / 256), dim3(256), 0, st, b.pc_ptr.as<long>(),
                         b.pc_r.as<int>(), b.Mv.as<double>(), b.W.as<double>(), m, q_lo, q_hi, N);
    }
    // one wave per stored entry at every size (round 3; C5: 1.74 -> 1.16 ms -- the workgroup-per-column kernel exposes the
    // HBM latency at each of its entries: a block reduction and two barriers between one 80 KB column of W and the next)
    static const bool dot_wg = getenv("LRN_SP_DOT_WG") != nullptr;           // (measurement: the round-1 kernel)
    if (small || !dot_wg)
      hipLaunchKernelGGL(sp_dot_wave_kernel, dim3((unsigned)((b.ncq + 3) / 4)), dim3(256), 0, st, b.cq_q.as<long>(),
                         b.pc_t.as<int>(), b.ncq, b.W.as<double>(), N, m, q_lo, q_hi, mirror ? 1 : 0, b.Zs.as<double>());
    else
      hipLaunchKernelGGL(sp_dot_kernel, dim3(q_hi - q_lo), dim3(256), 0, st, b.pc_ptr.as<long>(), b.pc_r.as<int>(),
                         b.pc_t.as<int>(), b.W.as<double>(), N, m, q_lo, mirror ? 1 : 0, b.Zs.as<double>());
    hipLaunchKernelGGL(sp_aa_times_kernel, dim3((b.npos_nz + 3) / 4), dim3(256), 0, st, b.ent_ptr.as<long>(),
                       b.ent_c.as<int>(), b.ent_v.as<double>(), b.ent_t.as<int>(), b.Zs.as<double>(), q_lo, q_hi, b.npos_nz,
                       b.sigma_d.as<int>(), y);
  }
  return LRN_OK;
}

__global__ void vec_add_kernel(double* __restrict__ y, const double* __restrict__ t, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) y[e] += t[e];
}

// multi-GPU: are the passes over the dense constraint matrices split over the ranks?  (communicator present, enough dense
// constraints for every rank; option "shard_passes" = 0 keeps them replicated)
static bool dense_passes_sharded(lrn_ctx* c, const LmiBlock& b) {
  return c->comm && c->world > 1 && c->opt.shard_passes != 0 && b.nd >= 8 * c->world;
}

int wmw(lrn_ctx* c, LmiBlock& b, double* M, double* P, double* Z) {
  const int m = b.msz;
  if (products_sharded(c, c->stream, m)) {      // P = W M' (M symmetric), Z = P W' (W symmetric): column blocks + all-gather
    LRN_TRY(pgemm_nt(c, c->stream, m, b.W.as<double>(), M, P));
    return pgemm_nt_sym(c, c->stream, m, P, b.W.as<double>(), Z);
  }
  GemmDesc g1;     // P = W M   (M symmetric: read as M[n + k*m] -> direct-to-LDS path)
  g1.A = b.W.as<double>(); g1.sAm = 1; g1.sAk = m;
  g1.B = M; g1.sBk = m; g1.sBn = 1;
  g1.C = P; g1.sCm = 1; g1.sCn = m;
  g1.M = g1.N = g1.K = m;
  LRN_TRY(gemm(c->stream, g1));
  if (m >= 1500) return gemm_nt_sym(c->stream, m, P, b.W.as<double>(), Z);      // Z = P W symmetric: lower tiles + mirror
  GemmDesc g2;     // Z = P W   (W symmetric)
  g2.A = P; g2.sAm = 1; g2.sAk = m;
  g2.B = b.W.as<double>(); g2.sBk = m; g2.sBn = 1;
  g2.C = Z; g2.sCm = 1; g2.sCn = m;
  g2.M = g2.N = g2.K = m;
  return gemm(c->stream, g2);
}

// AA vec(W M W) for a dense symmetric M when every constraint of the block is sparse (C5: 9 entries each): the entries
// of Z = W M W are needed on the pattern of the constraints only -- N = M W is one product, Z[p,q] = W(:,p) . N(:,q) one
// wave per stored entry (the kernels of the pattern-restricted CG operator above) -- instead of the second n^3 product.
bool wmw_pattern_ok(const lrn_ctx* c, const LmiBlock& b) {
  return b.sp_ok && b.nd == 0 && b.msz >= c->opt.wmw_pattern_min && b.have_W;
}

int aa_times_wmw_pattern(lrn_ctx* c, LmiBlock& b, const double* M, double* N, double* y) {
  const int m = b.msz;
  LRN_TRY(pgemm_nt(c, c->stream, m, M, b.W.as<double>(), N));                   // N = M W' = M W
  hipLaunchKernelGGL(sp_dot_wave_kernel, dim3((unsigned)((b.ncq + 3) / 4)), dim3(256), 0, c->stream, b.cq_q.as<long>(),
                     b.pc_t.as<int>(), b.ncq, b.W.as<double>(), N, m, 0, m, 1, b.Zs.as<double>());
  hipLaunchKernelGGL(sp_aa_times_kernel, dim3((b.npos_nz + 3) / 4), dim3(256), 0, c->stream, b.ent_ptr.as<long>(),
                     b.ent_c.as<int>(), b.ent_v.as<double>(), b.ent_t.as<int>(), b.Zs.as<double>(), 0, m, b.npos_nz,
                     b.sigma_d.as<int>(), y);
  return LRN_OK;
}

int ensure_m(lrn_ctx* c, int m) {
  size_t mm = (size_t)m * m * 8;
  LRN_TRY(ensure(c, c->m0, mm));
  LRN_TRY(ensure(c, c->m1, mm));
  LRN_TRY(ensure(c, c->m2, mm));
  return LRN_OK;
}

// y += AA vec(Z)
int aa_times(lrn_ctx* c, LmiBlock& b, const double* Z, double* y) {
  if (b.npos_nz > b.nd)
    hipLaunchKernelGGL(aa_times_kernel, dim3((b.npos_nz - b.nd + 3) / 4), dim3(256), 0, c->stream, b.ent_ptr.as<long>(),
                       b.ent_r.as<int>(), b.ent_c.as<int>(), b.ent_v.as<double>(), Z, b.msz, b.nd, b.npos_nz,
                       b.sigma_d.as<int>(), y);
  if (b.nd > 0) {
    int p0 = 0, p1 = b.nd;
    if (dense_passes_sharded(c, b)) {
      // one process per GPU: every pass over the dense constraint data (128 GB at C4: 25 ms at 5.1 TB/s, six of them per IP
      // iteration in the replicated part of the loop) is split by constraints; the nvar-vector of partial results is summed
      // by one all-reduce on the library's stream
      const int per = (b.nd + c->world - 1) / c->world;
      p0 = std::min(b.nd, c->rank * per);
      p1 = std::min(b.nd, p0 + per);
      LRN_TRY(ensure(c, c->commvec, (size_t)c->nvar * 8));
      LRN_HIP(c, hipMemsetAsync(c->commvec.p, 0, (size_t)c->nvar * 8, c->stream));
      if (p1 > p0) {
        if (dense_tri_ok(c, b)) {
          double* Wt = nullptr;
          LRN_TRY(tri_weights(c, b, Z, 0, &Wt));
          hipLaunchKernelGGL(aa_dense_tri_dot4_kernel<1>, dim3((p1 - p0 + 3) / 4), dim3(256), 0, c->stream,
                             b.Adense.as<double>() + (long)p0 * b.msz * b.msz, (long)b.msz * b.msz, p1 - p0, Wt, Wt,
                             b.tri_tab.as<TriChunk>(), b.tri_nch, b.sigma_d.as<int>() + p0, c->commvec.as<double>(),
                             c->commvec.as<double>());
        } else if (dense_stream_ok(b, Z))
          hipLaunchKernelGGL(aa_dense_dot4_kernel, dim3((p1 - p0 + 3) / 4), dim3(256), 0, c->stream,
                             b.Adense.as<double>() + (long)p0 * b.msz * b.msz, (long)b.msz * b.msz, p1 - p0, Z,
                             b.sigma_d.as<int>() + p0, c->commvec.as<double>());
        else
        hipLaunchKernelGGL(aa_dense_dot_kernel, dim3(p1 - p0), dim3(256), 0, c->stream,
                           b.Adense.as<double>() + (long)p0 * b.msz * b.msz, (long)b.msz * b.msz, Z, b.sigma_d.as<int>() + p0,
                           c->commvec.as<double>());
      }
      LRN_TRY(comm_allreduce(c, c->commvec.as<double>(), c->nvar, 0));
      hipLaunchKernelGGL(vec_add_kernel, dim3(nb(c->nvar)), dim3(256), 0, c->stream, y, c->commvec.as<double>(), c->nvar);
      return LRN_OK;
    }
    if (dense_tri_ok(c, b)) {
      double* Wt = nullptr;
      LRN_TRY(tri_weights(c, b, Z, 0, &Wt));
      hipLaunchKernelGGL(aa_dense_tri_dot4_kernel<1>, dim3((b.nd + 3) / 4), dim3(256), 0, c->stream, b.Adense.as<double>(),
                         (long)b.msz * b.msz, b.nd, Wt, Wt, b.tri_tab.as<TriChunk>(), b.tri_nch, b.sigma_d.as<int>(), y, y);
    } else if (dense_stream_ok(b, Z))
      hipLaunchKernelGGL(aa_dense_dot4_kernel, dim3((b.nd + 3) / 4), dim3(256), 0, c->stream, b.Adense.as<double>(),
                         (long)b.msz * b.msz, b.nd, Z, b.sigma_d.as<int>(), y);
    else
    hipLaunchKernelGGL(aa_dense_dot_kernel, dim3(b.nd), dim3(256), 0, c->stream, b.Adense.as<double>(),
                       (long)b.msz * b.msz, Z, b.sigma_d.as<int>(), y);
  }
  return LRN_OK;
}

// out1[sigma[p]] -= <Adense[p], Z1>, out2[sigma[p]] -= <Adense[p], Z2>: ONE pass over the dense constraint data for two
// products (C4: 128 GB per pass)
__global__ __launch_bounds__(256) void aa_dense_dot2_kernel(const double* __restrict__ Ad, long mm, const double* __restrict__ Z1,
                                                            const double* __restrict__ Z2, const int* __restrict__ sigma,
                                                            double* __restrict__ out1, double* __restrict__ out2) {
  __shared__ double sh[8];
  const double* a = Ad + (long)blockIdx.x * mm;
  double s1 = 0.0, s2 = 0.0;
  for (long q = threadIdx.x; q < mm; q += 256) { const double v = a[q]; s1 += v * Z1[q]; s2 += v * Z2[q]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s1; sh[4 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out1[sigma[blockIdx.x]] -= sh[0] + sh[1] + sh[2] + sh[3];
    out2[sigma[blockIdx.x]] -= sh[4] + sh[5] + sh[6] + sh[7];
  }
}

// y1 += AA vec(Z1), y2 += AA vec(Z2) with the dense constraint data read once
int aa_times2(lrn_ctx* c, LmiBlock& b, const double* Z1, double* y1, const double* Z2, double* y2) {
  if (b.nd <= 0 || dense_passes_sharded(c, b)) {      // (the sharded pass is 1/world of the data already)
    LRN_TRY(aa_times(c, b, Z1, y1));
    return aa_times(c, b, Z2, y2);
  }
  if (b.npos_nz > b.nd) {
    for (int h = 0; h < 2; ++h)
      hipLaunchKernelGGL(aa_times_kernel, dim3((b.npos_nz - b.nd + 3) / 4), dim3(256), 0, c->stream, b.ent_ptr.as<long>(),
                         b.ent_r.as<int>(), b.ent_c.as<int>(), b.ent_v.as<double>(), h ? Z2 : Z1, b.msz, b.nd, b.npos_nz,
                         b.sigma_d.as<int>(), h ? y2 : y1);
  }
  if (dense_tri_ok(c, b)) {
    double *W1 = nullptr, *W2 = nullptr;
    LRN_TRY(tri_weights(c, b, Z1, 0, &W1));
    LRN_TRY(tri_weights(c, b, Z2, 1, &W2));
    hipLaunchKernelGGL(aa_dense_tri_dot4_kernel<2>, dim3((b.nd + 3) / 4), dim3(256), 0, c->stream, b.Adense.as<double>(),
                       (long)b.msz * b.msz, b.nd, W1, W2, b.tri_tab.as<TriChunk>(), b.tri_nch, b.sigma_d.as<int>(), y1, y2);
  } else if (dense_stream_ok(b, Z1) && dense_stream_ok(b, Z2))
    hipLaunchKernelGGL(aa_dense_dot4x2_kernel, dim3((b.nd + 3) / 4), dim3(256), 0, c->stream, b.Adense.as<double>(),
                       (long)b.msz * b.msz, b.nd, Z1, Z2, b.sigma_d.as<int>(), y1, y2);
  else
  hipLaunchKernelGGL(aa_dense_dot2_kernel, dim3(b.nd), dim3(256), 0, c->stream, b.Adense.as<double>(), (long)b.msz * b.msz,
                     Z1, Z2, b.sigma_d.as<int>(), y1, y2);
  return LRN_OK;
}

// M = mat(AA' x)  (symmetrised msz x msz, kron_etc.jl:13-18)
int aat_to_mat(lrn_ctx* c, LmiBlock& b, const double* x, double* M) {
  const int m = b.msz;
  const long mm = (long)m * m;
  LRN_HIP(c, hipMemsetAsync(M, 0, (size_t)mm * 8, c->stream));
  if (b.ncq > 0)
    hipLaunchKernelGGL(aat_gather_kernel, dim3((unsigned)((b.ncq + 3) / 4)), dim3(256), 0, c->stream, b.cq_q.as<long>(),
                       b.cq_ptr.as<long>(), b.cq_j.as<int>(), b.cq_v.as<double>(), b.ncq, x, M);
  // symmetric dense data: the sparse part is symmetrised first, the dense constraints are added to the LOWER triangle only
  // (half the bytes of the pass) and the result is mirrored -- the same matrix up to the order of the additions
  const bool tri = b.nd > 0 && dense_tri_ok(c, b);
  if (tri && b.ncq > 0) symmetrize_dev(c->stream, M, m);
  if (b.nd > 0) {
    int p0 = 0, p1 = b.nd;
    double* Tm = M;
    const bool sharded = dense_passes_sharded(c, b);
    if (sharded) {
      // this rank's constraints into a zeroed buffer, one all-reduce of the msz x msz partial sums, then added to M
      const int per = (b.nd + c->world - 1) / c->world;
      p0 = std::min(b.nd, c->rank * per);
      p1 = std::min(b.nd, p0 + per);
      LRN_TRY(ensure(c, c->commmat, (size_t)mm * 8));
      Tm = c->commmat.as<double>();
      LRN_HIP(c, hipMemsetAsync(Tm, 0, (size_t)mm * 8, c->stream));
    }
    if (p1 > p0) {
      const double* Ap0 = b.Adense.as<double>() + (long)p0 * mm;
      const int* sg = b.sigma_d.as<int>() + p0;
      if (tri)
        hipLaunchKernelGGL(aat_dense_tri_kernel, dim3((unsigned)((b.tri_nch + 3) / 4)), dim3(256), 0, c->stream, Ap0, p1 - p0,
                           mm, m, sg, x, b.tri_tab.as<TriChunk>(), b.tri_nch, Tm);
      else if (dense_stream_ok(b, Tm))
        hipLaunchKernelGGL(aat_dense2_kernel, dim3((unsigned)((mm / 2 + 255) / 256)), dim3(256), 0, c->stream, Ap0, p1 - p0, mm,
                           sg, x, Tm);
      else
        hipLaunchKernelGGL(aat_dense_kernel, dim3(nb(mm)), dim3(256), 0, c->stream, Ap0, p1 - p0, mm, sg, x, Tm);
    }
    if (sharded) {
      LRN_TRY(comm_allreduce(c, Tm, mm, 0));
      hipLaunchKernelGGL(vec_add_kernel, dim3(nb(mm)), dim3(256), 0, c->stream, M, Tm, mm);
    }
  }
  if (tri)
    hipLaunchKernelGGL(mirror_lower_tiled_kernel, dim3((m + 31) / 32, (m + 31) / 32), dim3(32, 8), 0, c->stream, M, m);
  else
    symmetrize_dev(c->stream, M, m);
  return LRN_OK;
}

int matvec_dev(lrn_ctx* c, const double* x, double* y) {
  const int n = c->nvar;
  LRN_HIP(c, hipMemsetAsync(y, 0, (size_t)n * 8, c->stream));
  for (auto& b : c->lmi) {
    if (!b.have_W) return set_error(c, LRN_ERR_STATE, "W not set");
    const int m = b.msz;
    if (use_sparse_matvec(c, b)) {
      LRN_TRY(matvec_sparse_block(c, b, x, y, 0, m, true));
      continue;
    }
    LRN_TRY(ensure_m(c, m));
    double* M = c->m0.as<double>();
    LRN_TRY(aat_to_mat(c, b, x, M));
    LRN_TRY(wmw(c, b, M, c->m1.as<double>(), c->m2.as<double>()));
    LRN_TRY(aa_times(c, b, c->m2.as<double>(), y));
  }
  if (c->nlin > 0) LRN_TRY(lin_matvec(c, x, y));
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// Partial mat-vec of rank `rank` of `world`: rows R_g of Z = W M W (row blocks of msz/world),
// y_g = AA[:, idx(R_g)] vec(Z[R_g,:]); the caller all-reduces y_g over the ranks
// (one all-reduce of an nvar-vector per mat-vec).  The C_lin term is added by rank 0.
int matvec_partial_dev(lrn_ctx* c, const double* x, double* y, int rank, int world) {
  const int n = c->nvar;
  LRN_HIP(c, hipMemsetAsync(y, 0, (size_t)n * 8, c->stream));
  for (auto& b : c->lmi) {
    if (!b.have_W) return set_error(c, LRN_ERR_STATE, "W not set");
    const int m = b.msz;
    const int per = (m + world - 1) / world;
    const int r0 = std::min(m, rank * per), r1 = std::min(m, r0 + per);
    if (r1 <= r0) continue;
    const int nr = r1 - r0;
    if (use_sparse_matvec(c, b)) {       // shard the pattern columns of Z instead of its rows
      LRN_TRY(matvec_sparse_block(c, b, x, y, r0, r1, false));
      continue;
    }
    LRN_TRY(ensure_m(c, m));
    double* M = c->m0.as<double>();
    LRN_TRY(aat_to_mat(c, b, x, M));              // replicated: sparse, cheap
    GemmDesc g1;                                  // P_g = W[R_g,:] M
    g1.A = b.W.as<double>() + r0; g1.sAm = 1; g1.sAk = m;
    g1.B = M; g1.sBk = m; g1.sBn = 1;             // M symmetric
    g1.C = c->m1.as<double>(); g1.sCm = 1; g1.sCn = nr;
    g1.M = nr; g1.N = m; g1.K = m;
    LRN_TRY(gemm(c->stream, g1));
    GemmDesc g2;                                  // Z_g = P_g W
    g2.A = c->m1.as<double>(); g2.sAm = 1; g2.sAk = nr;
    g2.B = b.W.as<double>(); g2.sBk = m; g2.sBn = 1;
    g2.C = c->m2.as<double>(); g2.sCm = 1; g2.sCn = nr;
    g2.M = nr; g2.N = m; g2.K = m;
    LRN_TRY(gemm(c->stream, g2));
    if (b.npos_nz > b.nd)
      hipLaunchKernelGGL(aa_times_rows_kernel, dim3((b.npos_nz - b.nd + 3) / 4), dim3(256), 0, c->stream,
                         b.ent_ptr.as<long>(), b.ent_r.as<int>(), b.ent_c.as<int>(), b.ent_v.as<double>(),
                         c->m2.as<double>(), nr, r0, r1, b.nd, b.npos_nz, b.sigma_d.as<int>(), y);
    if (b.nd > 0)
      hipLaunchKernelGGL(aa_dense_dot_rows_kernel, dim3(b.nd), dim3(256), 0, c->stream, b.Adense.as<double>(), m,
                         c->m2.as<double>(), nr, r0, r1, b.sigma_d.as<int>(), y);
  }
  if (c->nlin > 0 && rank == 0) LRN_TRY(lin_matvec(c, x, y));
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// ------------------------------------------------------------------ vector kernels (single workgroup)
__device__ __forceinline__ double wg_sum1024(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += sh[i];
  return s;
}

// scal: [0]=gamma [1]=pAp [2]=alpha [3]=rr [4]=flag(alpha invalid) [5]=beta [6]=||b|| [7]=zr
__global__ __launch_bounds__(1024) void cg_norm_kernel(const double* __restrict__ b, int n, double* __restrict__ scal, int slot) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += b[i] * b[i];
  s = wg_sum1024(s, sh);
  if (threadIdx.x == 0) scal[slot] = s;
}

// ---- the CG recurrence as two multi-workgroup launches per iteration (round 4; round 3: two single-workgroup kernels of
// 33 + 15 us at nvar 20 000 and a host read between them).
//   cg_b_kernel(it): q = p'Ap (from the partial sums the operator left, or a redundant dot per workgroup), alpha = g / q,
//                    x += alpha p, r -= alpha Ap on the workgroup's slice, z = M^-1 r inline when the preconditioner is
//                    diagonal (none, H_beta), partial sums of r'r and z'r
//   cg_d_kernel(it): r'r -> the convergence test of ConjugateGradients.jl (relative residual <= tol); z'r, beta,
//                    p = z + beta p on the slice
// Every workgroup forms the global scalars itself from the same partials in the same order: no grid barrier, no atomics,
// all workgroups take the same branch.  scal[8] = exit code once the iteration has ended (30 converged, -13 alpha
// invalid), scal[9] = the iteration it ended in, both mirrored to host-mapped words: the host queues iterations ahead of
// the test it has read (option pcg_lookahead); kernels queued beyond the last iteration find the flag set and leave
// x, r, p alone -- count and result are those of the loop that tests after every step.  g = r'z alternates between
// scal[10] and scal[11] (a workgroup of cg_d_kernel must not overwrite the value its neighbours are still reading).
__device__ __forceinline__ double wg_sum256(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

static constexpr int CG_MAXWG = 256;

__global__ __launch_bounds__(256) void cg_b_kernel(const double* __restrict__ p, const double* __restrict__ Ap,
                                                   double* __restrict__ r, double* __restrict__ x, double* __restrict__ z,
                                                   int kind, const double* __restrict__ d, int n, int per,
                                                   double* __restrict__ scal, const double* __restrict__ qpart, int nq,
                                                   double* __restrict__ rrpart, double* __restrict__ zrpart,
                                                   double* __restrict__ hostw, int it) {
  __shared__ double sh[4];
  if (scal[8] != 0.0) return;
  double q = 0.0;
  if (qpart) for (int i = threadIdx.x; i < nq; i += 256) q += qpart[i];
  else for (int i = threadIdx.x; i < n; i += 256) q += p[i] * Ap[i];
  q = wg_sum256(q, sh);
  const double g = scal[10 + ((it - 1) & 1)];
  const double alpha = g / q;
  const bool bad = !(alpha >= 0.0) || isinf(alpha);
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal[1] = q; scal[2] = alpha; scal[4] = bad ? 1.0 : 0.0; }
  if (bad) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      scal[9] = (double)it;
      hostw[1] = (double)it;
      hostw[0] = -13.0;
      __threadfence_system();
      scal[8] = -13.0;
    }
    return;
  }
  const int i0 = blockIdx.x * per, i1 = min(n, i0 + per);
  double rr = 0.0, zr = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    rr += ri * ri;
    if (kind == 0) { z[i] = ri; zr += ri * ri; }                       // MyM_no   (Solvers.jl:620-622)
    else if (kind == 2) { const double zi = ri / d[i]; z[i] = zi; zr += zi * ri; }   // MyM_beta (:670-672)
  }
  rr = wg_sum256(rr, sh);
  zr = wg_sum256(zr, sh);
  if (threadIdx.x == 0) { rrpart[blockIdx.x] = rr; zrpart[blockIdx.x] = zr; }
}

// first != 0: the start of the iteration, p = z, g = z'r (no test, no beta)
__global__ __launch_bounds__(256) void cg_d_kernel(const double* __restrict__ z, const double* __restrict__ r,
                                                   double* __restrict__ p, int n, int per, int nwg,
                                                   double* __restrict__ scal, const double* __restrict__ rrpart,
                                                   const double* __restrict__ zrpart, double* __restrict__ hostw, int it,
                                                   double res0, double tol, int first) {
  __shared__ double sh[4];
  if (scal[8] != 0.0) return;
  if (!first) {
    double rr = 0.0;
    for (int i = threadIdx.x; i < nwg; i += 256) rr += rrpart[i];
    rr = wg_sum256(rr, sh);
    if (sqrt(rr) / res0 <= tol) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal[3] = rr;
        scal[9] = (double)it;
        hostw[1] = (double)it;
        hostw[0] = 30.0;
        __threadfence_system();
        scal[8] = 30.0;
      }
      return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[3] = rr;
  }
  double zr = 0.0;
  if (zrpart) for (int i = threadIdx.x; i < nwg; i += 256) zr += zrpart[i];
  else for (int i = threadIdx.x; i < n; i += 256) zr += z[i] * r[i];
  zr = wg_sum256(zr, sh);
  const double beta = first ? 0.0 : zr / scal[10 + ((it - 1) & 1)];
  const int i0 = blockIdx.x * per, i1 = min(n, i0 + per);
  if (first) for (int i = i0 + threadIdx.x; i < i1; i += 256) p[i] = z[i];
  else for (int i = i0 + threadIdx.x; i < i1; i += 256) p[i] = z[i] + beta * p[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal[10 + (it & 1)] = zr; scal[5] = beta; scal[7] = zr; }
}

__global__ void div_kernel(const double* __restrict__ x, const double* __restrict__ d, double* __restrict__ y, int n, int sq) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] / (sq ? sqrt(d[i]) : d[i]);
}

__global__ void fill_kernel(double* __restrict__ d, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = v;
}

// y[c] = sum_i ts[i + c*n] v[i]   (one workgroup per column); d != null: v[i] = x[i] / sqrt(d[i]) formed on the fly
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ ts, int n, const double* __restrict__ v,
                                                     const double* __restrict__ d, double* __restrict__ y) {
  __shared__ double sh[4];
  const double* col = ts + (long)blockIdx.x * n;
  double s = 0.0;
  if (d) for (int i = threadIdx.x; i < n; i += 256) s += col[i] * (v[i] / sqrt(d[i]));
  else
  for (int i = threadIdx.x; i < n; i += 256) s += col[i] * v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) y[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// zpart[chunk][i] = sum_{c in chunk} ts[i + c*n] y[c]
__global__ __launch_bounds__(256) void gemv_n_part_kernel(const double* __restrict__ ts, int n, int ncol, int cper,
                                                          const double* __restrict__ y, double* __restrict__ zpart) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int c0 = blockIdx.y * cper, c1 = min(ncol, c0 + cper);
  if (i >= n) return;
  double s = 0.0;
  for (int cc = c0; cc < c1; ++cc) s += ts[(long)i + (long)cc * n] * y[cc];
  zpart[(long)blockIdx.y * n + i] = s;
}

// out = (v - sum_chunks zpart) / sqrt(d); scaled != 0: v = x / sqrt(d) formed on the fly from x
__global__ void smw_final_kernel(const double* __restrict__ v, const double* __restrict__ zpart, int nchunk, int n,
                                 const double* __restrict__ d, double* __restrict__ out, int scaled) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < nchunk; ++k) s += zpart[(long)k * n + i];
  const double sd = sqrt(d[i]);
  out[i] = ((scaled ? v[i] / sd : v[i]) - s) / sd;
}

// ------------------------------------------------------------------ H_alpha setup kernels
// Um[:,a] = E[:,idx[a]] * coef[a]
__global__ void umat_kernel(const double* __restrict__ E, int m, const int* __restrict__ idx,
                            const double* __restrict__ coef, int k, double* __restrict__ Um) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * k) return;
  int i = e % m, a = e / m;
  Um[e] = E[(long)i + (long)idx[a] * m] * coef[a];
}

// Zf = 2 W - Um Um'
__global__ void zfull_kernel(const double* __restrict__ W, const double* __restrict__ Um, int m, int k,
                             double* __restrict__ Zf) {
  long total = (long)m * m;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = (int)(e % m), j = (int)(e / m);
    double s = 0.0;
    for (int a = 0; a < k; ++a) s += Um[i + a * m] * Um[j + a * m];
    Zf[e] = 2.0 * W[e] - s;
  }
}

__global__ void tril2_kernel(double* __restrict__ A, int n) {
  long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    if ((int)(e % n) < (int)(e / n)) A[e] = 0.0;
}

// AU[j, r] += -a_e * Um[c, a] / sqrt(d_j), j = sigma[p]   (one thread per sparse position)
__global__ void au_sparse_kernel(const long* __restrict__ ptr, const int* __restrict__ er, const int* __restrict__ ec,
                                 const double* __restrict__ ev, int p_lo, int p_end, const int* __restrict__ sigma,
                                 const double* __restrict__ Ucol, const double* __restrict__ d, int nvar,
                                 double* __restrict__ AU) {
  int p = p_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= p_end) return;
  int j = sigma[p];
  double sc = -1.0 / sqrt(d[j]);
  for (long e = ptr[p]; e < ptr[p + 1]; ++e) AU[(long)j + (long)er[e] * nvar] += sc * ev[e] * Ucol[ec[e]];
}

// dense slot p: AU[sigma[p], r] = -(A_p u)[r] / sqrt(d)
__global__ __launch_bounds__(256) void au_dense_kernel(const double* __restrict__ Ad, int m, const int* __restrict__ sigma,
                                                       const double* __restrict__ Ucol, const double* __restrict__ d,
                                                       int nvar, double* __restrict__ AU) {
  const double* A = Ad + (long)blockIdx.x * m * m;
  int j = sigma[blockIdx.x];
  double sc = -1.0 / sqrt(d[j]);
  for (int r = threadIdx.x; r < m; r += 256) {
    double s = 0.0;
    for (int cidx = 0; cidx < m; ++cidx) s += A[(long)r + (long)cidx * m] * Ucol[cidx];
    AU[(long)j + (long)r * nvar] = sc * s;
  }
}

// Cd[i, l] = C_lin[i, l] * sqrt(xs_l)   (dense nvar x nlin image of the linear block)
__global__ void lin_dense_kernel(const long* __restrict__ ptr, const int* __restrict__ row, const double* __restrict__ val,
                                 const double* __restrict__ xs, int nlin, int n, double* __restrict__ Cd) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlin) return;
  double s = sqrt(xs[l]);
  for (long k = ptr[l]; k < ptr[l + 1]; ++k) Cd[(long)row[k] + (long)l * n] += val[k] * s;
}

__global__ void prec_add_diag_kernel(double* __restrict__ S, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) S[(long)i * n + i] += v;
}

__global__ void add_eye_kernel(double* __restrict__ S, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) S[(long)i * n + i] += 1.0;
}


static double tau_of(const std::vector<double>& lam_s, int aamat) {
  // Solvers.jl:646-650 / :715-719
  double mn = lam_s[0], mean = 0.0;
  for (double v : lam_s) { mn = std::min(mn, v); mean += v; }
  mean /= (double)lam_s.size();
  if (aamat == 0) return 1.0 * mn;
  return (mn + mean) / 2.0 - 1.0e-14;
}

int prec_setup(lrn_ctx* c, int kind, int erank, int aamat, int* info) {
  if (info) *info = 0;
  if (!c->prec) c->prec = new Prec();
  Prec* P = c->prec;
  P->kind = kind;
  P->erank = erank;
  P->has_dense = false;
  const int n = c->nvar;
  hipStream_t st = c->stream;
  if (kind == 0) return LRN_OK;
  if (kind != 1 && kind != 2) return set_error(c, LRN_ERR_ARG, "preconditioner %d not supported", kind);
  if (c->nlmi < 1) return set_error(c, LRN_ERR_STATE, "preconditioner needs at least one LMI block");
  P->has_LD = kind == 1 && c->nlin > 0;
  hipEvent_t a0, a1;
  if (c->profile) { (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventRecord(a0, st); }
  int ksz = 0;
  for (auto& b : c->lmi) ksz += erank * b.msz;
  P->ksz = ksz;
  LRN_TRY(ensure(c, P->d, (size_t)n * 8));
  double dsum = 0.0;
  struct BlkEig { std::vector<int> idx; std::vector<double> coef; double tau; bool lanczos = false; double trace = 0.0; };
  std::vector<BlkEig> be(c->nlmi);
  for (int il = 0; il < c->nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    const int m = b.msz, k = erank;
    if (!b.have_G && !b.have_W) return set_error(c, LRN_ERR_STATE, "preconditioner setup needs G or W (lrn_prepare_w)");
    const bool fromW = !b.have_G;     // eigen-free scaling: eig(W) from W itself (its singular values ARE its eigenvalues)
    if (k >= m) return set_error(c, LRN_ERR_ARG, "erank >= matrix size");
    size_t mm = (size_t)m * m * 8;
    LRN_TRY(ensure(c, P->E, mm));
    LRN_TRY(ensure(c, P->sig, (size_t)m * 8));
    const bool use_lz = c->opt.prec_eig == 2 || (c->opt.prec_eig == 0 && m >= 256);
    be[il].lanczos = use_lz;
    be[il].idx.resize(k);
    be[il].coef.resize(k);
    if (use_lz) {
      // the setup only consumes the k largest eigenpairs, lambda_min and the mean of the rest
      // (lanczos.hip): O(steps * msz^2) instead of a full eigendecomposition
      std::vector<double> lam_top(std::max(1, k));
      double lam_min = 0.0, tr = 0.0;
      int steps = 0;
      LRN_TRY(lanczos_extremes(c, b.W.as<double>(), m, k, lam_top.data(), kind == 1 ? P->E.as<double>() : nullptr,
                               &lam_min, &tr, &steps));
      c->counts["prec_lanczos_steps"] = steps;
      double top = 0.0;
      for (int a = 0; a < k; ++a) top += lam_top[a];
      const double mean = (tr - top) / (double)(m - k);
      const double tau = aamat == 0 ? lam_min : (lam_min + mean) / 2.0 - 1.0e-14;     // Solvers.jl:646-650
      be[il].tau = tau;
      be[il].trace = tr;
      if (aamat < 3) dsum += tau * tau;
      for (int a = 0; a < k; ++a) {
        be[il].idx[a] = a;                                                  // Ritz vectors are unit columns 0..k-1
        be[il].coef[a] = std::sqrt(std::max(lam_top[a] - tau, 0.0));
      }
    } else {
      LRN_HIP(c, hipMemcpyAsync(P->E.p, fromW ? b.W.p : b.G.p, mm, hipMemcpyDeviceToDevice, st));
      int sweeps = 0;
      // eig(W) = svd(G)^2 : columns of E become sigma_j u_j   (Solvers.jl:642,706)
      LRN_TRY(jacobi_svd(c, P->E.as<double>(), nullptr, P->sig.as<double>(), m, &sweeps));
      std::vector<double> sg(m);
      LRN_TRY(copy_out(c, sg.data(), P->sig.p, (size_t)m * 8));
      std::vector<int> ord(m);
      for (int i = 0; i < m; ++i) ord[i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return sg[x] < sg[y]; });   // ascending
      auto lam_of = [&](int i) { return fromW ? sg[i] : sg[i] * sg[i]; };
      std::vector<double> lam_s(m - k);
      for (int i = 0; i < m - k; ++i) lam_s[i] = lam_of(ord[i]);
      double tau = tau_of(lam_s, aamat);
      be[il].tau = tau;
      for (int i = 0; i < m; ++i) be[il].trace += lam_of(i);
      if (aamat < 3) dsum += tau * tau;
      for (int a = 0; a < k; ++a) {
        int id = ord[m - k + a];
        double lam = lam_of(id);
        be[il].idx[a] = id;
        be[il].coef[a] = std::sqrt(std::max(lam - tau, 0.0)) / sg[id];   // Umat = v_l sqrt(lambda_l - tau)
      }
    }
    if (kind == 2) continue;
  }
  P->dsum = dsum;
  hipLaunchKernelGGL(fill_kernel, dim3(nb(n)), dim3(256), 0, st, P->d.as<double>(), n, P->has_LD ? 1.0 : dsum);
  if (P->has_LD) {
    // L_D L_D' = dsum I + (C_lin sqrt(xs)) (C_lin sqrt(xs))'
    const int nl = c->nlin;
    LRN_TRY(ensure(c, P->Cd, (size_t)n * nl * 8));
    LRN_TRY(ensure(c, P->LD, (size_t)n * n * 8));
    LRN_TRY(ensure(c, P->linvD, chol_linv_doubles(n) * 8));
    LRN_TRY(ensure(c, P->wD, (size_t)n * CHOL_NB * 8 + (size_t)CHOL_NB * std::max(ksz, 1) * 8));
    LRN_HIP(c, hipMemsetAsync(P->Cd.p, 0, (size_t)n * nl * 8, st));
    hipLaunchKernelGGL(lin_dense_kernel, dim3(nb(nl)), dim3(256), 0, st, c->cl_ptr.as<long>(), c->cl_rown.as<int>(),
                       c->cl_val.as<double>(), c->lin_xs.as<double>(), nl, n, P->Cd.as<double>());
    GemmDesc gd;
    gd.A = P->Cd.as<double>(); gd.sAm = 1; gd.sAk = n;
    gd.B = P->Cd.as<double>(); gd.sBk = n; gd.sBn = 1;
    gd.C = P->LD.as<double>(); gd.sCm = 1; gd.sCn = n;
    gd.M = gd.N = n; gd.K = nl;
    gd.flags = GEMM_TRI_LOWER;
    LRN_TRY(gemm(st, gd));
    hipLaunchKernelGGL(prec_add_diag_kernel, dim3(nb(n)), dim3(256), 0, st, P->LD.as<double>(), n, dsum);
    LRN_HIP(c, hipMemsetAsync(c->info_dev.p, 0, 4, st));
    LRN_TRY(potrf_lower(st, P->LD.as<double>(), n, n, P->linvD.as<double>(), P->wD.as<double>(), c->info_dev.as<int>()));
    int hd = 0;
    LRN_TRY(copy_out(c, &hd, c->info_dev.p, 4));
    if (hd != 0) { if (info) *info = hd; return LRN_OK; }
  } else if (c->nlin > 0)
    hipLaunchKernelGGL(lin_diag_kernel, dim3(nb(n)), dim3(256), 0, st, c->cr_ptr.as<long>(), c->cr_col.as<int>(),
                       c->cr_val.as<double>(), c->lin_xs.as<double>(), n, P->d.as<double>());
  if (kind == 1) {
    LRN_TRY(ensure(c, P->ts, (size_t)n * ksz * 8));
    int col0 = 0;
    for (int il = 0; il < c->nlmi; ++il) {
      LmiBlock& b = c->lmi[il];
      const int m = b.msz, k = erank;
      size_t mm = (size_t)m * m * 8;
      // eigenvectors again (E was overwritten by later blocks only when nlmi > 1)
      if (c->nlmi > 1) {
        if (be[il].lanczos) {
          std::vector<double> lt(std::max(1, k));
          double lmn, trc;
          LRN_TRY(lanczos_extremes(c, b.W.as<double>(), m, k, lt.data(), P->E.as<double>(), &lmn, &trc, nullptr));
        } else {
          LRN_HIP(c, hipMemcpyAsync(P->E.p, b.have_G ? b.G.p : b.W.p, mm, hipMemcpyDeviceToDevice, st));
          int sw = 0;
          LRN_TRY(jacobi_svd(c, P->E.as<double>(), nullptr, P->sig.as<double>(), m, &sw));
        }
      }
      LRN_TRY(ensure(c, P->Um, (size_t)m * k * 8 + (size_t)k * 16));
      double* Um = P->Um.as<double>();
      double* coef_d = Um + (size_t)m * k;
      int* idx_d = reinterpret_cast<int*>(coef_d + k);
      LRN_TRY(copy_in(c, coef_d, be[il].coef.data(), (size_t)k * 8));
      LRN_TRY(copy_in(c, idx_d, be[il].idx.data(), (size_t)k * 4));
      hipLaunchKernelGGL(umat_kernel, dim3(nb((long)m * k)), dim3(256), 0, st, P->E.as<double>(), m, idx_d, coef_d, k, Um);
      // Z = chol(2 W0 + Um Um') = chol(2W - Um Um')   (Solvers.jl:725-731)
      LRN_TRY(ensure_m(c, m));
      double* Zf = c->m0.as<double>();
      LRN_TRY(ensure(c, P->linvS, chol_linv_doubles(std::max(m, ksz)) * 8));
      LRN_TRY(ensure(c, P->cw, (size_t)std::max(m, ksz) * CHOL_NB * 8));
      // 2W - UU' is positive definite in exact arithmetic; late in the solve cond(W) passes 1e16 and
      // the rounding of W = GG' can cost the factorisation (the reference's eigen-based W0 is exposed to
      // the same, Solvers.jl:725-731 raise PosDefException).  A preconditioner only has to be SPD:
      // retry with a relative diagonal shift instead of giving up.
      int h = 0;
      for (int attempt = 0; attempt < 4; ++attempt) {
        hipLaunchKernelGGL(zfull_kernel, dim3(nb((long)m * m)), dim3(256), 0, st, b.W.as<double>(), Um, m, k, Zf);
        if (attempt > 0) {
          const double shift = be[il].trace / m * 1e-15 * std::pow(100.0, attempt);
          hipLaunchKernelGGL(prec_add_diag_kernel, dim3(nb(m)), dim3(256), 0, st, Zf, m, shift);
          c->counts["prec_z_shift"] += 1;
        }
        LRN_HIP(c, hipMemsetAsync(c->info_dev.p, 0, 4, st));
        LRN_TRY(potrf_lower(st, Zf, m, m, P->linvS.as<double>(), P->cw.as<double>(), c->info_dev.as<int>()));
        LRN_TRY(copy_out(c, &h, c->info_dev.p, 4));
        if (h == 0) break;
      }
      if (h != 0) { if (info) *info = h; return LRN_OK; }
      hipLaunchKernelGGL(tril2_kernel, dim3(nb((long)m * m)), dim3(256), 0, st, Zf, m);
      // ts[:, block a] = (D^-1/2 AU_a) Z
      LRN_TRY(ensure(c, P->AU, (size_t)n * m * 8));
      for (int a = 0; a < k; ++a) {
        LRN_HIP(c, hipMemsetAsync(P->AU.p, 0, (size_t)n * m * 8, st));
        const double* Ucol = Um + (size_t)a * m;
        if (b.npos_nz > b.nd)
          hipLaunchKernelGGL(au_sparse_kernel, dim3(nb(b.npos_nz - b.nd)), dim3(256), 0, st, b.ent_ptr.as<long>(),
                             b.ent_r.as<int>(), b.ent_c.as<int>(), b.ent_v.as<double>(), b.nd, b.npos_nz,
                             b.sigma_d.as<int>(), Ucol, P->d.as<double>(), n, P->AU.as<double>());
        if (b.nd > 0)
          hipLaunchKernelGGL(au_dense_kernel, dim3(b.nd), dim3(256), 0, st, b.Adense.as<double>(), m, b.sigma_d.as<int>(),
                             Ucol, P->d.as<double>(), n, P->AU.as<double>());
        GemmDesc g;
        g.A = P->AU.as<double>(); g.sAm = 1; g.sAk = n;
        g.B = Zf; g.sBk = 1; g.sBn = m;
        g.C = P->ts.as<double>() + (size_t)(col0 + a * m) * n; g.sCm = 1; g.sCn = n;
        g.M = n; g.N = m; g.K = m;
        LRN_TRY(gemm(st, g));
      }
      col0 += k * m;
    }
    if (P->has_LD)      // ts = L_D^-1 t   (the reference: AAAATtau \ t, Solvers.jl:767)
      LRN_TRY(trsm_left_lower(st, P->LD.as<double>(), n, n, P->linvD.as<double>(), false, P->ts.as<double>(), ksz, n,
                              P->wD.as<double>() + (size_t)n * CHOL_NB));
    // S = ts' ts + I ; cholS   (Solvers.jl:804-805)
    LRN_TRY(ensure(c, P->cholS, (size_t)ksz * ksz * 8));
    GemmDesc g;
    g.A = P->ts.as<double>(); g.sAm = n; g.sAk = 1;
    g.B = P->ts.as<double>(); g.sBk = 1; g.sBn = n;
    g.C = P->cholS.as<double>(); g.sCm = 1; g.sCn = ksz;
    g.M = g.N = ksz; g.K = n;
    g.flags = GEMM_TRI_LOWER;
    LRN_TRY(gemm(st, g));
    hipLaunchKernelGGL(add_eye_kernel, dim3(nb(ksz)), dim3(256), 0, st, P->cholS.as<double>(), ksz);
    // explicit (S + I)^-1 with one step of iterative refinement in the apply: three single-launch mat-vecs instead of the
    // eight super-block launches of the two triangular solves per CG iteration (Solvers.jl:883)
    P->has_inv = ksz <= 8192 && (c->opt.prec_inv == 1 || (c->opt.prec_inv < 0 && ksz >= 256));
    if (P->has_inv) {
      LRN_TRY(ensure(c, P->Sm, (size_t)ksz * ksz * 8));
      LRN_HIP(c, hipMemcpyAsync(P->Sm.p, P->cholS.p, (size_t)ksz * ksz * 8, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(mirror_lower_full_kernel, dim3(nb((long)ksz * ksz)), dim3(256), 0, st, P->Sm.as<double>(), ksz);
    }
    LRN_TRY(ensure(c, P->linvS, chol_linv_doubles(ksz) * 8));
    LRN_TRY(ensure(c, P->cw, (size_t)ksz * CHOL_NB * 8));
    LRN_HIP(c, hipMemsetAsync(c->info_dev.p, 0, 4, st));
    LRN_TRY(potrf_lower(st, P->cholS.as<double>(), ksz, ksz, P->linvS.as<double>(), P->cw.as<double>(),
                        c->info_dev.as<int>()));
    int h = 0;
    LRN_TRY(copy_out(c, &h, c->info_dev.p, 4));
    if (h != 0) { if (info) *info = h; return LRN_OK; }
    if (P->has_inv) {
      LRN_TRY(ensure(c, P->Ainv, (size_t)ksz * ksz * 8));
      LRN_TRY(ensure_m(c, ksz));
      double* Li = c->m0.as<double>();
      double* LiT = c->m1.as<double>();
      LRN_TRY(ensure(c, P->cw, ((size_t)ksz * CHOL_NB + (size_t)CHOL_NB * ksz) * 8));
      hipLaunchKernelGGL(eye_fill_kernel, dim3(nb((long)ksz * ksz)), dim3(256), 0, st, Li, ksz);
      LRN_TRY(trsm_left_lower(st, P->cholS.as<double>(), ksz, ksz, P->linvS.as<double>(), false, Li, ksz, ksz,
                              P->cw.as<double>() + (size_t)ksz * CHOL_NB));
      hipLaunchKernelGGL(transpose_sq_kernel, dim3((ksz + 31) / 32, (ksz + 31) / 32), dim3(32, 8), 0, st, Li, ksz, LiT);
      LRN_TRY(gemm_nt_sym(st, ksz, LiT, LiT, P->Ainv.as<double>(), 1.0));      // L^-T L^-1
    }
    LRN_TRY(ensure(c, P->y, (size_t)(ksz + 64) * 8));
    LRN_TRY(ensure(c, P->y2, (size_t)(ksz + 64) * 8));
    LRN_TRY(ensure(c, P->y3, (size_t)(ksz + 64) * 8));
    LRN_TRY(ensure(c, P->y4, (size_t)(ksz + 64) * 8));
    LRN_TRY(ensure(c, P->zpart, (size_t)32 * n * 8));
  }
  if (c->profile) {
    (void)hipEventRecord(a1, st); (void)hipEventSynchronize(a1);
    float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
    c->timing["prec_setup"] += ms; c->counts["prec_setup"] += 1;
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
  }
  LRN_HIP(c, hipGetLastError());
  return LRN_OK;
}

// Minv[i,j] = (delta_ij - G[i,j]) / sqrt(d_i d_j) on the lower triangle (G = T1 ts', lower tiles)
__global__ void minv_finish_kernel(double* __restrict__ Mi, int n, const double* __restrict__ d) {
  const long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e % n), j = (int)(e / n);
    if (i >= j) Mi[e] = ((i == j ? 1.0 : 0.0) - Mi[e]) / sqrt(d[i] * d[j]);
  }
}

// H_alpha as a dense matrix: is it worth forming for `expected` applications?  (static model, as hop_worthwhile)
static bool prec_dense_worthwhile(const lrn_ctx* c, const Prec* P, long expected) {
  if (!P || P->kind != 1 || P->has_LD || !P->has_inv || c->opt.prec_dense == 1) return false;
  const double n = c->nvar, k = P->ksz;
  if (c->nvar > 8192 || c->nvar < 256) return false;
  if (c->opt.prec_dense == 2) return true;
  const double setup = (6.0 * n * k * k + n * n * k) / 3.5e13 + 80e-6;      // T1 with one refinement step, T1 ts' (lower tiles)
  const double per_apply = 45e-6;                                            // seven launches -> two
  return (double)expected * per_apply > 1.2 * setup;
}

// T1 = ts (S + I)^-1 with one step of refinement (the accuracy of the SMW apply, which refines too), Minv from it
static int prec_dense_build(lrn_ctx* c, Prec* P) {
  const int n = c->nvar, ksz = P->ksz;
  hipStream_t st = c->stream;
  LRN_TRY(ensure(c, P->Minv, (size_t)n * n * 8));
  LRN_TRY(ensure(c, P->T1, (size_t)2 * n * ksz * 8));
  double* T1 = P->T1.as<double>();
  double* R = T1 + (size_t)n * ksz;
  auto mm = [&](const double* A, const double* B, double* C, double alpha, double beta) -> int {   // C = alpha A B + beta C, (n x ksz)(ksz x ksz), B symmetric
    GemmDesc g;
    g.A = A; g.sAm = 1; g.sAk = n;
    g.B = B; g.sBk = 1; g.sBn = ksz;
    g.C = C; g.sCm = 1; g.sCn = n;
    g.M = n; g.N = ksz; g.K = ksz;
    g.alpha = alpha; g.beta = beta;
    return gemm(st, g);
  };
  LRN_TRY(mm(P->ts.as<double>(), P->Ainv.as<double>(), T1, 1.0, 0.0));
  LRN_HIP(c, hipMemcpyAsync(R, P->ts.p, (size_t)n * ksz * 8, hipMemcpyDeviceToDevice, st));
  LRN_TRY(mm(T1, P->Sm.as<double>(), R, -1.0, 1.0));                        // R = ts - T1 (S + I)
  LRN_TRY(mm(R, P->Ainv.as<double>(), T1, 1.0, 1.0));                       // T1 += R (S + I)^-1
  GemmDesc g;                                                               // G = T1 ts' (lower tiles)
  g.A = T1; g.sAm = 1; g.sAk = n;
  g.B = P->ts.as<double>(); g.sBk = n; g.sBn = 1;
  g.C = P->Minv.as<double>(); g.sCm = 1; g.sCn = n;
  g.M = g.N = n; g.K = ksz;
  g.flags = GEMM_TRI_LOWER;
  LRN_TRY(gemm(st, g));
  hipLaunchKernelGGL(minv_finish_kernel, dim3(nb((long)n * n)), dim3(256), 0, st, P->Minv.as<double>(), n, P->d.as<double>());
  P->has_dense = true;
  c->counts["prec_dense_build"] += 1;
  return LRN_OK;
}

// Mx = M^-1 x, device vectors; tmpv: nvar scratch
int prec_apply_dev(lrn_ctx* c, const double* x, double* Mx, double* tmpv) {
  Prec* P = c->prec;
  const int n = c->nvar;
  hipStream_t st = c->stream;
  if (!P || P->kind == 0) {                                  // MyM_no  (Solvers.jl:620-622)
    LRN_HIP(c, hipMemcpyAsync(Mx, x, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    return LRN_OK;
  }
  if (P->kind == 2) {                                        // MyM_beta (Solvers.jl:670-672)
    hipLaunchKernelGGL(div_kernel, dim3(nb(n)), dim3(256), 0, st, x, P->d.as<double>(), Mx, n, 0);
    return LRN_OK;
  }
  if (P->has_dense) {                                        // (lrn_pcg when the cost model formed it; option prec_dense = 2)
    c->counts["prec_dense_apply"] += 1;
    return symv_lower(c, P->Minv.as<double>(), n, nullptr, x, Mx);
  }
  const int ksz = P->ksz;                                    // MyM (Solvers.jl:866-904), ts form
  if (P->has_LD) {      // v = L_D^-1 x  (d holds ones)
    hipLaunchKernelGGL(div_kernel, dim3(nb(n)), dim3(256), 0, st, x, P->d.as<double>(), tmpv, n, 1);
    LRN_TRY(trsm_left_lower(st, P->LD.as<double>(), n, n, P->linvD.as<double>(), false, tmpv, 1, n,
                            P->wD.as<double>() + (size_t)n * CHOL_NB));
    hipLaunchKernelGGL(gemv_t_kernel, dim3(ksz), dim3(256), 0, st, P->ts.as<double>(), n, tmpv, (const double*)nullptr,
                       P->y.as<double>());
  } else {              // v = x ./ sqrt(d) formed inside the two kernels that read it (same operations, one launch less)
    hipLaunchKernelGGL(gemv_t_kernel, dim3(ksz), dim3(256), 0, st, P->ts.as<double>(), n, x, P->d.as<double>(),
                       P->y.as<double>());
  }
  if (P->has_inv) {
    // y2 = Ainv y; refinement: y3 = y - (S + I) y2, y2 += Ainv y3 -- the accuracy of the triangular solves, three launches
    const unsigned g16 = (unsigned)((ksz + 15) / 16);
    const size_t lds = (size_t)ksz * 8;
    hipLaunchKernelGGL(symv_rows_kernel, dim3(g16), dim3(256), lds, st, P->Ainv.as<double>(), ksz, P->y.as<double>(),
                       (const double*)nullptr, 1.0, 0.0, P->y4.as<double>());
    hipLaunchKernelGGL(symv_rows_kernel, dim3(g16), dim3(256), lds, st, P->Sm.as<double>(), ksz, P->y4.as<double>(),
                       P->y.as<double>(), -1.0, 1.0, P->y3.as<double>());
    hipLaunchKernelGGL(symv_rows_kernel, dim3(g16), dim3(256), lds, st, P->Ainv.as<double>(), ksz, P->y3.as<double>(),
                       P->y4.as<double>(), 1.0, 1.0, P->y2.as<double>());
  } else {
    LRN_TRY(potrs_vec(st, P->cholS.as<double>(), ksz, ksz, P->linvS.as<double>(), P->y.as<double>(), P->y2.as<double>(),
                      P->y3.as<double>(), P->y4.as<double>()));
  }
  const int nchunk = std::min(32, std::max(1, ksz / 32));
  const int cper = (ksz + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(gemv_n_part_kernel, dim3((n + 255) / 256, nchunk), dim3(256), 0, st, P->ts.as<double>(), n, ksz,
                     cper, P->y2.as<double>(), P->zpart.as<double>());
  hipLaunchKernelGGL(smw_final_kernel, dim3(nb(n)), dim3(256), 0, st, P->has_LD ? tmpv : x, P->zpart.as<double>(), nchunk, n,
                     P->d.as<double>(), Mx, P->has_LD ? 0 : 1);
  if (P->has_LD)
    LRN_TRY(trsm_left_lower(st, P->LD.as<double>(), n, n, P->linvD.as<double>(), true, Mx, 1, n,
                            P->wD.as<double>() + (size_t)n * CHOL_NB));
  return LRN_OK;
}

// Which operator serves this NT scaling: the assembled Schur matrix (hop.hip) or the matrix-free MyA.  Decided once per
// scaling from the static cost model and the CG iterations of the previous scaling; with a communicator the ranks take
// the decision together (free memory and so the assembly path can differ between them).
int op_select(lrn_ctx* c, bool* use_h) {
  if (c->hop_version != c->scal_version) {
    c->cg_prev_iters = c->cg_cur_iters;
    c->cg_cur_iters = 0;
    bool use = hop_worthwhile(c, c->cg_prev_iters);
    if (c->comm && c->world > 1) {
      double w[1] = {use ? 0.0 : 1.0};
      LRN_TRY(comm_status_max(c, w, 1));
      use = w[0] == 0.0;
    }
    c->hop_use = use;
    c->hop_version = c->scal_version;
  }
  if (c->hop_use) {
    const int rc = hop_prepare(c);
    if (rc != LRN_OK) {
      if (c->comm && c->world > 1) return rc;       // (every rank returns it: the status reduction of the exchange)
      c->hop_use = false;                            // one GPU: the matrix-free operator needs no workspace
      c->counts["hop_fallback"] += 1;
      c->err.clear();
    }
  }
  *use_h = c->hop_use;
  return LRN_OK;
}

// Ap = A p by the selected operator (all-reduced when sharded)
static int op_apply(lrn_ctx* c, bool use_h, const double* p, double* Ap, double* qpart = nullptr, int* nq = nullptr) {
  const bool sharded = c->comm && c->world > 1;
  if (nq) *nq = 0;
  if (use_h) {
    if (c->opt.profile_symv) tic(c);
    LRN_TRY(hop_apply(c, p, Ap, qpart, nq));
    if (c->opt.profile_symv) toc(c, "hop_symv");        // (measurement: this rank's share of H x alone, tools/shard_balance_c5.py)
  } else if (sharded) {
    // one process per GPU: this rank's rows of W M W, then ONE all-reduce of the nvar-vector on this stream -- the
    // recurrence is replicated and stays on the device, as on one GPU
    LRN_TRY(matvec_partial_dev(c, p, Ap, c->rank, c->world));
  } else {
    return matvec_dev(c, p, Ap);
  }
  if (sharded) LRN_TRY(comm_allreduce(c, Ap, c->nvar, 0));
  return LRN_OK;
}

// cg(A, b; tol, maxIter, precon) -- restates ConjugateGradients.jl 0.1 (see oracle.cg)
int pcg_dev(lrn_ctx* c, const double* b, double tol, int maxit, double* x, int* exit_code, int* iters) {
  const int n = c->nvar;
  hipStream_t st = c->stream;
  LRN_TRY(ensure(c, c->cgbuf, (size_t)(6 * (size_t)n + 64) * 8));
  double* r = c->cgbuf.as<double>();
  double* z = r + n;
  double* p = z + n;
  double* Ap = p + n;
  double* tmpv = Ap + n;
  double* scal = tmpv + n;          // 16 doubles
  constexpr int NSLOT = 16;
  if (!c->pin) {
    LRN_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->pin), NSLOT * 16 * 8, hipHostMallocMapped));
    LRN_HIP(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->pin_dev), c->pin, 0));
  }
  if (!c->pcg_ev[0])
    for (int i = 0; i < NSLOT; ++i) LRN_HIP(c, hipEventCreateWithFlags(&c->pcg_ev[i], hipEventDisableTiming));
  volatile double* hs = c->pin;
  // partial sums of the recurrence kernels and of the operator (hop.hip): after the six vectors
  const int nwg = std::max(1, std::min(CG_MAXWG, (n + 255) / 256));
  const int per = (n + nwg - 1) / nwg;
  LRN_TRY(ensure(c, c->cgpart, (size_t)(2 * CG_MAXWG + (n + 63) / 64 + 64) * 8));
  double* rrpart = c->cgpart.as<double>();
  double* zrpart = rrpart + CG_MAXWG;
  double* qpart = zrpart + CG_MAXWG;
  long nmv = 0;
  LRN_HIP(c, hipMemsetAsync(x, 0, (size_t)n * 8, st));
  LRN_HIP(c, hipMemsetAsync(scal, 0, 16 * 8, st));
  hipLaunchKernelGGL(cg_norm_kernel, dim3(1), dim3(1024), 0, st, b, n, scal, 6);
  double bb = 0.0;
  LRN_HIP(c, hipMemcpyAsync(&bb, scal + 6, 8, hipMemcpyDeviceToHost, st));
  LRN_HIP(c, hipStreamSynchronize(st));
  if (std::sqrt(bb) == 0.0) { *exit_code = 1; *iters = 0; return LRN_OK; }
  // r = b - A*0 = b
  LRN_HIP(c, hipMemcpyAsync(r, b, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  const double residual_0 = std::sqrt(bb);
  if (residual_0 <= tol) { *exit_code = 2; *iters = 0; return LRN_OK; }
  for (int i = 0; i < NSLOT * 16; ++i) hs[i] = 0.0;
  bool use_h = false;
  LRN_TRY(op_select(c, &use_h));
  if (c->prec && !c->prec->has_dense && prec_dense_worthwhile(c, c->prec, c->cg_prev_iters)) LRN_TRY(prec_dense_build(c, c->prec));
  const int kind = (c->prec && (c->prec->kind == 1 || c->prec->kind == 2)) ? c->prec->kind : 0;
  const double* dprec = kind == 2 ? c->prec->d.as<double>() : nullptr;
  LRN_TRY(prec_apply_dev(c, r, z, tmpv));
  hipLaunchKernelGGL(cg_d_kernel, dim3(nwg), dim3(256), 0, st, z, r, p, n, per, nwg, scal, rrpart, (const double*)nullptr,
                     c->pin_dev, 0, residual_0, tol, 1);
  // The host runs `ahead` iterations in front of the convergence test it has read: the exit words of iteration `it` are
  // written to host-mapped memory by its kernels and looked at (behind an event) while iteration it + ahead is queued.
  const int ahead = std::max(0, std::min(NSLOT - 2, c->opt.pcg_lookahead));
  int done_code = 0, done_it = 0;
  auto poll = [&](int it) -> int {          // the words of iteration `it`
    const int s = it % NSLOT;
    LRN_HIP(c, hipEventSynchronize(c->pcg_ev[s]));
    if (hs[s * 16] != 0.0) { done_code = (int)hs[s * 16]; done_it = (int)hs[s * 16 + 1]; }
    return LRN_OK;
  };
  int it = 1;
  for (; it <= maxit && done_code == 0; ++it) {
    int nq = 0;
    LRN_TRY(op_apply(c, use_h, p, Ap, qpart, &nq));
    ++nmv;
    const int s = it % NSLOT;
    hipLaunchKernelGGL(cg_b_kernel, dim3(nwg), dim3(256), 0, st, p, Ap, r, x, z, kind, dprec, n, per, scal,
                       nq > 0 ? qpart : (const double*)nullptr, nq, rrpart, zrpart, c->pin_dev + s * 16, it);
    if (kind == 1) LRN_TRY(prec_apply_dev(c, r, z, tmpv));
    hipLaunchKernelGGL(cg_d_kernel, dim3(nwg), dim3(256), 0, st, z, r, p, n, per, nwg, scal, rrpart,
                       kind == 1 ? (const double*)nullptr : zrpart, c->pin_dev + s * 16, it, residual_0, tol, 0);
    LRN_HIP(c, hipEventRecord(c->pcg_ev[s], st));
    if (it - ahead >= 1) LRN_TRY(poll(it - ahead));
  }
  const int last = std::min(it - 1, maxit);
  for (int k = std::max(1, last - ahead + 1); k <= last && done_code == 0; ++k) LRN_TRY(poll(k));
  LRN_HIP(c, hipStreamSynchronize(st));
  LRN_HIP(c, hipGetLastError());
  c->counts["matvec"] += nmv;
  if (done_code != 0) { *exit_code = done_code; *iters = done_it; }
  else { *exit_code = -2; *iters = maxit; }
  c->cg_cur_iters += *iters;
  return LRN_OK;
}

}  // namespace lrn

using namespace lrn;

extern "C" int lrn_matvec(lrn_ctx* c, const double* x, double* Ax) {
  if (!c || !x || !Ax) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_TRY(copy_in(c, c->v0.p, x, (size_t)n * 8));
  // the assembled-matrix operator only when lrn_pcg chose it for this scaling, or when it is forced (option matvec_h = 2)
  bool use_h = false;
  if (c->opt.matvec_h == 2 || (c->hop_use && c->hop_version == c->scal_version)) LRN_TRY(op_select(c, &use_h));
  tic(c);
  LRN_TRY(op_apply(c, use_h, c->v0.as<double>(), c->v1.as<double>()));
  toc(c, "matvec");
  return copy_out(c, Ax, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_matvec_partial(lrn_ctx* c, const double* x, double* Ax_partial) {
  if (!c || !x || !Ax_partial) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_TRY(copy_in(c, c->v0.p, x, (size_t)n * 8));
  LRN_TRY(matvec_partial_dev(c, c->v0.as<double>(), c->v1.as<double>(), c->rank, c->world));
  return copy_out(c, Ax_partial, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_make_rhs(lrn_ctx* c, const double* Rp, const double* const* RdS, double* h) {
  if (!c || !Rp || !h || (c->nlmi > 0 && !RdS)) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_TRY(copy_in(c, c->v1.p, Rp, (size_t)n * 8));
  for (int il = 0; il < c->nlmi; ++il) {
    LmiBlock& b = c->lmi[il];
    if (!b.have_W) return set_error(c, LRN_ERR_STATE, "W not set");
    LRN_TRY(ensure_m(c, b.msz));
    LRN_TRY(copy_in(c, c->m0.p, RdS[il], (size_t)b.msz * b.msz * 8));
    LRN_TRY(wmw(c, b, c->m0.as<double>(), c->m1.as<double>(), c->m2.as<double>()));
    LRN_TRY(aa_times(c, b, c->m2.as<double>(), c->v1.as<double>()));
  }
  return copy_out(c, h, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_prec_setup(lrn_ctx* c, int prec, int erank, int aamat, int* info) {
  if (!c) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  return prec_setup(c, prec, erank, aamat, info);
}

extern "C" int lrn_prec_apply(lrn_ctx* c, const double* x, double* Mx) {
  if (!c || !x || !Mx) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_TRY(copy_in(c, c->v0.p, x, (size_t)n * 8));
  if (c->prec && !c->prec->has_dense && c->opt.prec_dense == 2 && prec_dense_worthwhile(c, c->prec, 0))
    LRN_TRY(prec_dense_build(c, c->prec));
  LRN_TRY(prec_apply_dev(c, c->v0.as<double>(), c->v1.as<double>(), c->v2.as<double>()));
  return copy_out(c, Mx, c->v1.p, (size_t)n * 8);
}

extern "C" int lrn_pcg(lrn_ctx* c, const double* h, double tol, int maxit, double* x, int* exit_code, int* iters) {
  if (!c || !h || !x || !exit_code || !iters) return LRN_ERR_ARG;
  LRN_HIP(c, hipSetDevice(c->device));
  const int n = c->nvar;
  LRN_TRY(copy_in(c, c->v0.p, h, (size_t)n * 8));
  hipEvent_t a0, a1;
  if (c->profile) { (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventRecord(a0, c->stream); }
  const int rc = pcg_dev(c, c->v0.as<double>(), tol, maxit, c->v3.as<double>(), exit_code, iters);
  if (rc != LRN_OK) {
    if (c->profile) { (void)hipEventDestroy(a0); (void)hipEventDestroy(a1); }
    return rc;
  }
  if (c->profile) {
    (void)hipEventRecord(a1, c->stream); (void)hipEventSynchronize(a1);
    float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
    c->timing["pcg"] += ms; c->counts["pcg"] += 1; c->counts["pcg_iters"] += *iters;
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
  }
  return copy_out(c, x, c->v3.p, (size_t)n * 8);
}
