// Internal declarations shared by the HIP translation units of libloraine_hip.so.
// gfx950 (MI355X / CDNA4) only.  Public C ABI: include/loraine_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define LRN_OK 0
#define LRN_ERR_ARG (-1)
#define LRN_ERR_HIP (-2)
#define LRN_ERR_STATE (-3)
#define LRN_ERR_NOMEM (-4)

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace lrn {

// Multi-GPU ownership of Schur column block `blk`: snake (boustrophedon) block-cyclic, so the
// triangular work (long columns first) is balanced across ranks.
__host__ __device__ inline int shard_owner(int blk, int world) {
  int r = blk % world;
  return ((blk / world) & 1) ? world - 1 - r : r;
}
// global block of (rank, local block lb)
__host__ __device__ inline int shard_global_block(int rank, int lb, int world) {
  return lb * world + ((lb & 1) ? world - 1 - rank : rank);
}

// ---------------------------------------------------------------- device buffer
struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---------------------------------------------------------------- GEMM descriptor
// C[m][n] (+)= alpha * sum_k opA[m][k] * opB[k][n], every operand addressed by strides
// (in elements), so any transpose / leading dimension / batch layout is one kernel.
enum : int {
  GEMM_TRI_LOWER = 1,      // compute only tiles that touch m >= n (needs BM == BN)
  GEMM_TRI_UPPER = 2,      // compute only tiles that touch m <= n
  GEMM_OFFDIAG_X2 = 4,     // multiply strictly off-diagonal tiles by 2 (packed-symmetric dot)
  GEMM_SQUARE = 8,         // epilogue c = (alpha*acc)^2 (+ beta*c)   (rank-one Schur)
  GEMM_KSEG_TRI = 16,      // K index = (col c, row r) of an ld x ncols matrix; rows
                           // r < (c / 128) * 128 are skipped (upper tiles of a lower-stored
                           // symmetric operand)
  GEMM_SMALL_TILE = 32,    // force the 64x64 tile
  GEMM_KFROM_N = 64,       // the K loop starts at the first column of the n-tile: op(B)[k][n] = 0 for
                           // k < n (B is the transposed lower Cholesky factor); forces the 128 tile
  GEMM_KFROM_M = 128,      // same with the m-tile (op(A)[m][k] = 0 for k < m)
  GEMM_C_PACKED = 256,     // C is an msz x msz symmetric matrix stored in the packed lower layout (16x16 blocks
                           // on and below the diagonal, packed_lower_offset below; pk_m); needs sCm == 1 on entry
  GEMM_KTO_N = 1024,       // the K loop ends with the n-tile (k < n0 + 128): op(B)[k][n] = 0 for k > n
  GEMM_KTO_M = 2048,       // same with the m-tile (op(A)[m][k] = 0 for k > m)
  GEMM_C_MIRROR = 4096,    // off-diagonal tiles are also stored transposed (C symmetric, computed one-sided)
  GEMM_DIAG_LOWER = 8192,  // K-contiguous kernels (GEMM3 / GEMM3'): in diagonal tiles only the 16x16 blocks on and below
                           // the diagonal (m >= n) are computed and stored (the caller never reads the others)
  GEMM_DIAG_UPPER = 16384, // same for n >= m (what GEMM_DIAG_LOWER becomes when gemm() transposes the problem)
  GEMM_TILE160 = 65536,    // GEMM_KFLAT only: 160 x 160 workgroup tile (gemm_f64_kseg_lds_kernel<true, 5>)
  GEMM_DYN_MASKS = 131072, // measurement only (option "gemm_dyn_masks"): no straight-line K-steps for the common block
                           // patterns, every masked K-step branches per block (the round-2 kernel)
  GEMM_NO_SKIP = 32768,    // measurement only (option "gemm_no_skip"): compute every block of every tile
  GEMM_DIAG_LOWER_Z = 262144, // direct-to-LDS kernel (GEMM1'): in diagonal tiles the 16x16 blocks above the block diagonal
                           // (m / 16 < n / 16) are not computed and STORED AS ZEROS (the caller multiplies them by zeros:
                           // stale NaNs must not be left there); a hint -- any other kernel computes them
  GEMM_LAB_SAME_CHUNK = 1 << 25, // measurement only: the K-contiguous kernel (GEMM3') requests the FIRST chunk of its split at every K-step
                                // (all panel loads hit in L2 after the first: what the kernel does when memory latency is out of the way)
  GEMM_LAB_NO_STORE = 1 << 20,  // measurement only (option "gemm_lab", tools/gemm12_overhead.py): direct-to-LDS kernel without
                                // its epilogue
  GEMM_LAB_NO_KLOOP = 1 << 21,  // ... without its K loop (the first K-step is still loaded and waited for)
  GEMM_LAB_NO_LOAD = 1 << 22,   // ... without the first load either (with the two above: an empty workgroup)
  GEMM_LAB_ONE_WG = 1 << 23,    // ... launched with 24 KB of unused dynamic LDS: one workgroup per CU instead of two
  GEMM_KFLAT = 512,        // both operands K-contiguous, K = flat index of the packed lower layout;
                           // the first kflat_nsd splits cover the diagonal blocks [0, kflat_diag), the
                           // others the strictly-lower blocks [kflat_diag, K)
};

// ---------------------------------------------------------------- packed lower layout
// A symmetric msz x msz matrix of which only the 16x16 blocks on and below the diagonal are kept
// (schur.hip, Cholesky path: At_k = L' A_k L).  S = msz rounded up to 16, column c lies in block column
// q = c / 16.  Flat index space:
//   [0, Kd)   Kd = 16 msz: column c at 16 c holds rows 16q .. 16q+15 (the diagonal block, both triangles)
//   [Kd, Kp)  column c: rows 16(q+1) .. S-1, contiguous
// Every column piece is a multiple of 16 doubles and rows >= msz are never written (stay zero), so a K walk
// in chunks of 16 never straddles the two regions.  <X,Y> = sum over [0,Kd) + 2 * sum over [Kd,Kp).
__host__ __device__ inline int packed_S(int m) { return (m + 15) & ~15; }
__host__ __device__ inline long packed_diag_elems(int m) { return 16L * m; }
// number of [Kd,Kp) elements in the columns before c
__host__ __device__ inline long packed_off_base(int c, int S) {
  const long q = c >> 4;
  return 16L * (q * S - 8L * q * (q + 1)) + (long)(c - 16 * q) * (S - 16 * (q + 1));
}
__host__ __device__ inline long packed_total_elems(int m) { return packed_diag_elems(m) + packed_off_base(m, packed_S(m)); }
// offset of element (r, c), r / 16 >= c / 16
__host__ __device__ inline long packed_lower_offset(int r, int c, int S, long Kd) {
  const int q = c >> 4;
  if ((r >> 4) == q) return 16L * c + (r - 16 * q);
  return Kd + packed_off_base(c, S) + (r - 16 * (q + 1));
}

struct GemmDesc {
  const double* A = nullptr;
  const double* B = nullptr;
  double* C = nullptr;
  double* C2 = nullptr;     // optional: the transposed result as well (same strides as C; square, unbatched products)
  int M = 0, N = 0, K = 0;
  long sAm = 0, sAk = 0, sBk = 0, sBn = 0, sCm = 0, sCn = 0;
  long bA = 0, bB = 0, bC = 0;
  int batch = 1;
  double alpha = 1.0, beta = 0.0;
  int flags = 0;
  // split-K: C must then point to `ksplit` slabs (slab stride sCs elements); the caller
  // reduces the slabs (reduce_slabs) -- deterministic, no atomics.
  int ksplit = 1;
  long sCs = 0;
  // GEMM_KSEG_TRI: K = kseg_ld * kseg_cols, segment list derived from (ld, cols)
  int kseg_ld = 0, kseg_cols = 0;
  // GEMM_C_PACKED: side of the packed matrix.  GEMM_KFLAT: K = kflat_total, diagonal region and its splits
  int pk_m = 0;
  // Chunk-major storage of an ARRAY of packed matrices: chunk q (16 doubles) of matrix k lives at
  // q * pk_cstride + 16 k + (0..15), so the same chunk of 128 consecutive matrices -- one operand panel of the
  // rank-k update per K-step -- is one contiguous 16 KB block (one DRAM page / TLB entry instead of 128 rows 16 MB
  // apart).  GEMM_C_PACKED: bC = 16; GEMM_KFLAT: sAm = sBn = 16, chunks kflat_cstride apart.  16 = one matrix alone.
  long pk_cstride = 16;
  int pk_off = 0;           // GEMM_C_PACKED on a trailing sub-block: kernel indices + pk_off = indices in the packed matrix
  long kflat_cstride = 16;
  // GEMM_KFLAT with explicit splits (multi-GPU partial sums over column ranges of the packed matrices): split s walks
  // the chunks [kflat_kb[s], kflat_ke[s]) (host arrays of `ksplit` entries); null = kflat_nsd splits over the
  // diagonal region and the others over the strictly-lower region of the whole matrix
  const int* kflat_kb = nullptr;
  const int* kflat_ke = nullptr;
  long kflat_total = 0, kflat_diag = 0;
  int kflat_nsd = 0;
  int tile_class = 0;      // 0 all tiles; 1 only tiles whose blocks are all computed; 2 only the tiles with skipped blocks
                           // (edge tiles, GEMM_DIAG_* diagonal tiles) -- see get_tile_list; 3 (GEMM_KFLAT): both in one
                           // launch, class 1 of every split first, class 2 last
  // tile_class 4 / 5 (GEMM_KFLAT, M == N, M % 128 in (0, 32]): 4 = as 3 for the leading M - 128 - M % 128 rows; 5 = the last
  // 128 + M % 128 rows tiled by 128 x 160
  unsigned long long* lab_trace = nullptr;   // measurement only (LRN_MID_TRACE): 8 words per workgroup, clocks of its phases
  int kstagger = 0;        // GEMM_KFLAT: workgroup (tm, tn) starts its K walk ((tm + tn) & 7) * kstagger chunks into
                           // its split and wraps around (see gemm_f64_kseg_lds_kernel)
};

int gemm(hipStream_t st, const GemmDesc& d);
// The result of a product as a sum of split-K slabs: value(e) = p[e] + p[stride + e] + ... (n slabs, fixed order -- the
// sum reduce_slabs would store).  gemm_slabs runs gemm() but leaves the slabs of a mid-size product unreduced for a
// consumer that adds them while it reads (a transpose, a symmetrisation, ...): n == 1 means the product is in d.C as usual.
// The slab memory belongs to the stream: it is valid until the next product on the same stream.
struct SlabSrc {
  const double* p = nullptr;
  long stride = 0;
  int n = 1;
};
__device__ __forceinline__ double slab_sum(const SlabSrc& s, long e) {
  double v = 0.0;
  for (int k = 0; k < s.n; ++k) v += s.p[(long)k * s.stride + e];
  return v;
}
int gemm_slabs(hipStream_t st, const GemmDesc& d, SlabSrc* out);
const char* gemm_last_error();   // reason of this thread's last gemm() failure, or null
// out[i] = beta*out[i] + sum_s slabs[s*stride + i]   (i < n), fixed summation order
int reduce_slabs(hipStream_t st, const double* slabs, long stride, int nslab, double* out,
                 long n, double beta);
// FP64 MFMA issue-rate probe: returns achieved TFLOP/s of a register-only MFMA loop.
int mfma_f64_peak(hipStream_t st, double* tflops);
// debug: one 16x16x4 MFMA with explicit operands (checks the lane maps)
int mfma_f64_probe(hipStream_t st, const double* A16x4, const double* B4x16, double* D16x16);

}  // namespace lrn
