// FP64 MFMA GEMM for gfx950 (MI355X): the contraction engine behind the Schur assembly
// (W*A_k*W products, <A_i, T_j> inner products), the blocked Cholesky trailing update,
// NT-scaling products and the CG mat-vec.
//
// Design (CDNA4-first, see DESIGN.md "K1"):
//  * v_mfma_f64_16x16x4_f64: one wave owns a (BM/2)x(BN/2) sub-tile = (BM/32)x(BN/32) MFMA
//    tiles, 4 waves (2x2) per 256-thread workgroup; all accumulators independent, so the
//    matrix pipe is issue-bound.  f64 C/D lane map: col = lane&15, row = (lane>>4) + 4*reg.
//  * operands staged global -> registers -> LDS, double-buffered, one barrier per K-tile;
//    the next tile's global loads are in flight under the current tile's MFMAs.
//  * LDS images padded so that the ds_read_b64 fragment reads are bank-conflict free:
//    [k][m] images use row stride BM+16 (== 16 mod 32), [m][k] images use stride BK+2.
//  * every operand is addressed by element strides, so all transposes / leading dimensions /
//    batches share this kernel; the launcher orients the problem so that the lane-contiguous
//    MFMA output dimension is the memory-contiguous dimension of C.
//  * blockIdx -> tile map is XCD-aware (8 XCDs, private L2s): each XCD walks a contiguous
//    range of tiles that share an operand panel.
//  * split-K writes per-split slabs that a second kernel reduces in fixed order
//    (deterministic; no float atomics).
#include "lrn_common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace lrn {

static constexpr int BK = 16;
static constexpr int MAX_KSPLIT = 64;

struct GemmParams {
  GemmDesc d;
  int tilesM, tilesN;
  int kchunk;                    // K elements per split (multiple of BK), non-KSEG
  int kcols[MAX_KSPLIT + 1];     // KSEG: column range per split; KFLAT: first chunk of split s
  int kcols2[MAX_KSPLIT + 1];    // KFLAT: end chunk of split s
  const int2* tile_list;         // (tm, tn) per workgroup, super-tile order
  // tile_class 3 (K-contiguous kernels): ONE launch walks the regular tiles of every split first and the tiles with
  // skipped blocks (tile_list2) last -- a 1-D grid in which XCD x = blockIdx.x % 8 takes entries [x q1, (x+1) q1) of
  // tile_list for split 0, 1, ... and then [x q2, (x+1) q2) of tile_list2 for split 0, 1, ...
  const int2* tile_list2;
  int n1, n2;                    // list lengths (multiples of 8); n2 < 0: the plain (tile, split) grid
  int m_org = 0, n_org = 0;      // K-contiguous kernels: origin of tile (0, 0) (a sub-range of C tiled on its own)
};

#define MFMA_F64_ROW(lane, r) (((lane) >> 4) + 4 * (r))
// An MFMA that accumulates IN PLACE behind a wave-uniform `if (block is needed)`.  One masked body serves every K-step
// (a second, unconditional copy of the body for the common all-blocks case made the compiler hoist the fragment loads
// of both copies above the branch: 256 VGPRs and > 100 spilled); the tied "+v" operand keeps each accumulator in its
// registers across the 16 branches of a K-quarter.  Inline asm is invisible to the hazard recogniser: the operands come
// from ds_read (ordered by s_waitcnt, which does see asm operands), a dependent MFMA on the same accumulator needs
// no wait states, and the one VALU read of the accumulators -- the epilogue -- is behind LRN_MFMA_DRAIN.
#define LRN_MFMA_INPLACE(c, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define LRN_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

template <int BM, bool KC>
__device__ __forceinline__ int lds_idx(int m, int k) {
  // KC: k-contiguous image [m][k], stride BK+2 ; else [k][m], stride BM+16
  return KC ? (m * (BK + 2) + k) : (k * (BM + 16) + m);
}

template <int BM, int BN, bool AKC, bool BKC, bool KSEG, bool EPI>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmParams p) {
  constexpr int TM = BM / 32, TN = BN / 32;       // MFMA tiles per wave
  constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
  constexpr int LA = BK * (BM + 16), LB = BK * (BN + 16);
  __shared__ double lds[2 * (LA + LB)];

  const GemmDesc& d = p.d;
  // ---- tile decode (XCD-aware, bijective for any grid size)
  int tm, tn;
  {
    int bid = blockIdx.x, nwg = gridDim.x;
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int2 tt = p.tile_list[swz];
    tm = tt.x;
    tn = tt.y;
  }
  if (tm < 0) return;      // padding entry of the tile list
  const int ks = blockIdx.z % d.ksplit;
  const int bz = blockIdx.z / d.ksplit;
  const double* __restrict__ Ag = d.A + (long)bz * d.bA;
  const double* __restrict__ Bg = d.B + (long)bz * d.bB;
  double* __restrict__ Cg = d.C + (long)bz * d.bC + (long)ks * d.sCs;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- K range
  long kcur, kend;             // non-KSEG
  int segc = 0, segcend = 0, segr = 0;   // KSEG
  if (KSEG) {
    segc = p.kcols[ks];
    segcend = p.kcols[ks + 1];
    segr = (segc / 128) * 128;
    kcur = 0; kend = 0;
  } else {
    kcur = (long)ks * p.kchunk;
    kend = kcur + p.kchunk;
    if (kend > d.K) kend = d.K;
    if (d.flags & GEMM_KFROM_N) kcur = n0;          // (ksplit == 1: checked by the launcher)
    if (d.flags & GEMM_KFROM_M) kcur = m0;
    if ((d.flags & GEMM_KTO_N) && kend > n0 + BN) kend = n0 + BN;
    if ((d.flags & GEMM_KTO_M) && kend > m0 + BM) kend = m0 + BM;
  }

  // ---- per-thread staging geometry
  // A: !AKC -> m fastest over threads ; AKC -> k fastest
  int a_m, a_k, a_dm, a_dk;    // first element (tile-local) and per-i increments
  if (AKC) { a_k = t & 15; a_m = t >> 4; a_dm = 16; a_dk = 0; }
  else     { a_m = t % BM; a_k = t / BM; a_dm = 0; a_dk = 256 / BM; }
  int b_n, b_k, b_dn, b_dk;
  if (BKC) { b_k = t & 15; b_n = t >> 4; b_dn = 16; b_dk = 0; }
  else     { b_n = t % BN; b_k = t / BN; b_dn = 0; b_dk = 256 / BN; }
  const double* pa = Ag + (long)(m0 + a_m) * d.sAm + (long)a_k * d.sAk;
  const double* pb = Bg + (long)(n0 + b_n) * d.sBn + (long)b_k * d.sBk;
  const long a_step = (long)a_dm * d.sAm + (long)a_dk * d.sAk;
  const long b_step = (long)b_dn * d.sBn + (long)b_dk * d.sBk;

  double ra[EA], rb[EB];
  v4f64 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  auto chunk_valid = [&]() -> bool { return KSEG ? (segc < segcend) : (kcur < kend); };
  auto chunk_base = [&]() -> long { return KSEG ? ((long)segc * d.kseg_ld + segr) : kcur; };
  auto chunk_len = [&]() -> int {
    long rem = KSEG ? (long)(d.kseg_ld - segr) : (kend - kcur);
    return rem < BK ? (int)rem : BK;
  };
  auto chunk_next = [&]() {
    if (KSEG) {
      segr += BK;
      if (segr >= d.kseg_ld) { ++segc; segr = (segc / 128) * 128; }
    } else {
      kcur += BK;
    }
  };
  auto gload = [&](long kb, int len) {
    const double* qa = pa + kb * d.sAk;
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      bool ok = (m0 + a_m + i * a_dm < d.M) && (a_k + i * a_dk < len);
      ra[i] = ok ? qa[i * a_step] : 0.0;
    }
    const double* qb = pb + kb * d.sBk;
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      bool ok = (n0 + b_n + i * b_dn < d.N) && (b_k + i * b_dk < len);
      rb[i] = ok ? qb[i * b_step] : 0.0;
    }
  };
  auto lstore = [&](int buf) {
    double* sa = lds + buf * (LA + LB);
    double* sb = sa + LA;
#pragma unroll
    for (int i = 0; i < EA; ++i) sa[lds_idx<BM, AKC>(a_m + i * a_dm, a_k + i * a_dk)] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) sb[lds_idx<BN, BKC>(b_n + i * b_dn, b_k + i * b_dk)] = rb[i];
  };

  if (chunk_valid()) {
    gload(chunk_base(), chunk_len());
    chunk_next();
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  const int fr = lane & 15, fk = lane >> 4;
  bool more = true;
  // number of chunks is uniform across the workgroup (depends only on block indices)
  {
    // first chunk may not exist at all (empty split): then nothing to do
    bool any = KSEG ? (p.kcols[ks] < p.kcols[ks + 1])
                    : ((d.flags & (GEMM_KFROM_N | GEMM_KFROM_M)) ? true : ((long)ks * p.kchunk < d.K));
    more = any;
  }
  while (more) {
    bool have_next = chunk_valid();
    if (have_next) {
      gload(chunk_base(), chunk_len());
      chunk_next();
    }
    const double* sa = lds + cur * (LA + LB);
    const double* sb = sa + LA;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = sa[lds_idx<BM, AKC>(wm * (BM / 2) + i * 16 + fr, kk * 4 + fk)];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = sb[lds_idx<BN, BKC>(wn * (BN / 2) + j * 16 + fr, kk * 4 + fk)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (have_next) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    more = have_next;
  }

  // ---- epilogue
  const bool x2 = EPI && (d.flags & GEMM_OFFDIAG_X2) && (tm != tn);
  const double alpha = x2 ? 2.0 * d.alpha : d.alpha;
  const bool sq = EPI && (d.flags & GEMM_SQUARE);
  // packed lower C (kernel n = row, kernel m = column of the symmetric matrix; tiles with tn >= tm)
  const bool pk = EPI && (d.flags & GEMM_C_PACKED);
  const bool mir = EPI && (d.flags & GEMM_C_MIRROR) && (tm != tn);
  const int pkS = packed_S(d.pk_m);
  const long pkKd = packed_diag_elems(d.pk_m > 0 ? d.pk_m : 1);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm * (BM / 2) + i * 16 + MFMA_F64_ROW(lane, r);
        int n = n0 + wn * (BN / 2) + j * 16 + fr;
        if (m < d.M && n < d.N) {
          double v = alpha * acc[i][j][r];
          if (sq) v = v * v;
          if (pk) {                               // 16x16 blocks on and below the diagonal only
            if ((n >> 4) >= (m >> 4)) {
              const long o = packed_lower_offset(n + d.pk_off, m + d.pk_off, pkS, pkKd);
              Cg[(o >> 4) * d.pk_cstride + (o & 15)] = v;
            }
            continue;
          }
          double* c = Cg + (long)m * d.sCm + (long)n * d.sCn;
          if (d.beta != 0.0) v += d.beta * (*c);
          *c = v;
          if (mir) Cg[(long)n * d.sCm + (long)m * d.sCn] = v;
          if (EPI && d.C2) d.C2[(long)n * d.sCm + (long)m * d.sCn] = v;      // transposed copy (square C, batch 1)
        }
      }
}

template <int BM, int BN, bool KSEG, bool EPI>
static void launch4(hipStream_t st, const GemmParams& p, bool akc, bool bkc, dim3 grid) {
  if (akc && bkc) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, true, KSEG, EPI>), grid, dim3(256), 0, st, p);
  else if (akc) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, false, KSEG, EPI>), grid, dim3(256), 0, st, p);
  else if (bkc) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true, KSEG, EPI>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, false, KSEG, EPI>), grid, dim3(256), 0, st, p);
}

// ------------------------------------------------------------------ direct-to-LDS variant
// Same tile / wave / MFMA structure as above for the case the hot Schur products are arranged
// in: both operands contiguous along their non-K dimension (op(A)[m][k] = A[m + k*lda],
// op(B)[k][n] = B[n + k*ldb], 16-byte aligned, even leading dimensions).  Each k-row of a
// 128-wide tile is exactly one `buffer_load_dwordx4 ... lds` wave-instruction (64 lanes x 16 B
// = 1 KiB) that lands in the padded [k][m] LDS image without touching VGPRs: no staging
// registers, no ds_write pass, no per-element predicates -- rows beyond M/N and k-rows beyond K
// are out of range of the buffer descriptor and read as zero.  Two LDS buffers, one barrier
// per K-tile (the form the CDNA guide recommends when ~2 workgroups share a CU).
// DYN: the round-2 form of the masked K-steps (a wave-uniform branch per block), kept for measurement (GEMM_DYN_MASKS).
template <bool EPI, bool DYN>
__global__ __launch_bounds__(256, 2) void gemm_f64_lds_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, TM = 4, TN = 4;
  constexpr int LROW = BM + 16;                 // doubles per k-row of an image
  constexpr int LA = BK * LROW;                 // doubles per image
  __shared__ double lds[2 * 2 * LA];
  const GemmDesc& d = p.d;
  int tm, tn;
  {
    int bid = blockIdx.x, nwg = gridDim.x;
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // triangular-K products: the tiles of one matrix differ in K length, and with nwg % 8 == 0 every
    // XCD would get the same run of the list for every batch element -- rotate the runs over the XCDs
    if ((d.flags & (GEMM_KFROM_N | GEMM_KFROM_M | GEMM_KTO_N | GEMM_KTO_M)) && r == 0) {
      xcd = (xcd + (int)blockIdx.z) & 7;
      swz = xcd * q + (bid >> 3);
    }
    int2 tt = p.tile_list[swz];
    tm = tt.x;
    tn = tt.y;
  }
  if (tm < 0) return;      // padding entry of the tile list
  const int bz = blockIdx.z;
  const double* Ag = d.A + (long)bz * d.bA;
  const double* Bg = d.B + (long)bz * d.bB;
  double* __restrict__ Cg = d.C + (long)bz * d.bC;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lda = (int)d.sAk, ldb = (int)d.sBk;
  // buffer descriptors: whole operand matrix of this batch element, K k-rows
  // (an odd edge is rounded up to the pair: ld is even and >= M, so that element exists)
  const int Me = d.M + (d.M & 1) <= lda ? d.M + (d.M & 1) : d.M;
  const int Ne = d.N + (d.N & 1) <= ldb ? d.N + (d.N & 1) : d.N;
  const unsigned bytesA = (unsigned)(((long)(d.K - 1) * lda + Me) * 8);
  const unsigned bytesB = (unsigned)(((long)(d.K - 1) * ldb + Ne) * 8);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bg, 0, bytesB, 0x00020000);
  // per-lane in-row byte offset; lanes whose pair starts beyond the matrix edge go out of range
  const int ma = m0 + 2 * lane, nb = n0 + 2 * lane;
  const unsigned offA = ma < d.M ? (unsigned)ma * 8u : 0x80000000u;
  const unsigned offB = nb < d.N ? (unsigned)nb * 8u : 0x80000000u;
  int kend = d.K;       // triangular operand: the K loop ends with the tile (tile origins are multiples of 128)
  if ((d.flags & GEMM_KTO_N) && kend > n0 + BN) kend = n0 + BN;
  if ((d.flags & GEMM_KTO_M) && kend > m0 + BM) kend = m0 + BM;
  const int nk = (kend + BK - 1) / BK;

  v4f64 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  auto issue = [&](int kt, int buf) {
    double* sa = lds + buf * (2 * LA);
    double* sb = sa + LA;
#pragma unroll
    for (int j = 0; j < BK / 4; ++j) {
      const int kr = w + 4 * j;                        // k-row of the tile handled by this wave
      const unsigned krow = (unsigned)(kt * BK + kr);
      // k-rows beyond K start at >= K*ld*8 > num_records -> zero fill
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + kr * LROW), 16,
                                               (int)(offA + krow * (unsigned)lda * 8u), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + kr * LROW), 16,
                                               (int)(offB + krow * (unsigned)ldb * 8u), 0, 0, 0);
    }
  };

  const int fr = lane & 15, fk = lane >> 4;
  // ---- which 16x16 blocks this wave computes.  Wave (wm, wn) owns the blocks (bm, bn) = (2 i + wm, 2 j + wn),
  // i, j < 4, of the 8 x 8 blocks of the tile -- INTERLEAVED, so that whatever part of a tile need not be computed
  // (blocks beyond M / N at the edges, the blocks above the diagonal of a packed diagonal tile, and, K-step by
  // K-step, the blocks a triangular operand leaves zero) is shared evenly by the four waves: a tile costs its
  // busiest wave.  Bit i * 4 + j of a mask = block (i, j) of this wave; masks are wave-uniform (SGPR).
  unsigned smask = 0;
  unsigned zmask = 0;        // blocks that are not computed but stored (as zeros): GEMM_DIAG_LOWER_Z
  {
    const bool diag_pk = EPI && (d.flags & GEMM_C_PACKED) && tm == tn;   // stored: blocks with bn >= bm only
    const bool diag_lz = (d.flags & GEMM_DIAG_LOWER_Z) && tm == tn;      // computed: blocks with bm >= bn only
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int bm = 2 * i + wm, bn = 2 * j + wn;
        bool need = (m0 + 16 * bm < d.M) && (n0 + 16 * bn < d.N);
        if (diag_pk && bn < bm) need = false;
        if (diag_lz && bm < bn && !(d.flags & GEMM_NO_SKIP)) {
          if (need) zmask |= 1u << (i * 4 + j);
          need = false;
        }
        if (d.flags & GEMM_NO_SKIP) need = true;
        if (need) smask |= 1u << (i * 4 + j);
      }
  }
  const bool from_n = d.flags & GEMM_KFROM_N, from_m = d.flags & GEMM_KFROM_M;
  const bool to_n = d.flags & GEMM_KTO_N, to_m = d.flags & GEMM_KTO_M;
  // K loop from the tile origin when an operand is triangular (tile origins are multiples of 128)
  const int kt0 = from_n ? n0 / BK : (from_m ? m0 / BK : 0);
  // Three loops over K with two bodies.  `fast`: all 16 blocks, the compiler's own schedule (builtin MFMAs, fragment
  // reads of the next quarter hoisted over the current one).  `masked`: a wave-uniform branch per block -- only for
  // the K-steps that cross the tile's own diagonal block of a triangular operand (the first 8 of KFROM, the last 8 of
  // KTO), and for every K-step of a tile with blocks to skip (edges, packed diagonal tiles).  Kept as SEPARATE loops:
  // as two branches of one loop body the compiler merges their identical fragment loads above the branch and spills.
  auto fast_step = [&](int kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const double* sa = lds + cur * (2 * LA);
    const double* sb = sa + LA;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = sa[(kk * 4 + fk) * LROW + (2 * i + wm) * 16 + fr];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = sb[(kk * 4 + fk) * LROW + (2 * j + wn) * 16 + fr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  // blocks of K-step kt: op(B)[k][n] = 0 for k < n (KFROM_N) means block column bn starts contributing at the K-step
  // that reaches its first column, i.e. kt - n0/16 >= bn; KTO_N (zero for k > n): kt - n0/16 <= bn.
  auto step_mask = [&](int kt) __attribute__((always_inline)) -> unsigned {
    unsigned mask = smask;
    const int sn = kt - n0 / BK, sm = kt - m0 / BK;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int bn = 2 * j + wn;
      if ((from_n && sn < bn) || (to_n && sn > bn)) mask &= ~(0x1111u << j);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int bm = 2 * i + wm;
      if ((from_m && sm < bm) || (to_m && sm > bm)) mask &= ~(0xfu << (4 * i));
    }
    return mask;
  };
  auto masked_step = [&](int kt, unsigned mask) __attribute__((always_inline)) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const double* sa = lds + cur * (2 * LA);
    const double* sb = sa + LA;
    if (mask != 0u) {
      // all fragments of the K-step first (one exposed LDS latency instead of four: nothing is scheduled across the
      // branches below), then the needed blocks
      double fa[BK / 4][TM], fb[BK / 4][TN];
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kk][i] = sa[(kk * 4 + fk) * LROW + (2 * i + wm) * 16 + fr];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kk][j] = sb[(kk * 4 + fk) * LROW + (2 * j + wn) * 16 + fr];
      }
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (mask & (1u << (i * 4 + j))) LRN_MFMA_INPLACE(acc[i][j], fa[kk][i], fb[kk][j]);
    }
    __syncthreads();
  };
  // A K-step whose set of blocks is one of the common patterns, as straight-line code (round 3): the compiler's own
  // schedule of fragment reads and builtin MFMAs, only the fragments the pattern needs.  The patterns: a prefix of the
  // wave's block rows or columns (the triangular head of a K range, step by step; the edge tiles of msz % 128 != 0),
  // and the blocks on and above the diagonal of a packed diagonal tile.  Any other set of blocks runs the smallest
  // pattern that contains it: a block outside the set multiplies explicit zeros (the other triangle of a triangular
  // operand is stored as zeros, rows beyond M / N read as zeros) or is never stored (packed diagonal tiles) -- what
  // GEMM_NO_SKIP does for every block.  The branch-per-block body above (round 2; with it next to these bodies the
  // register allocator spills the accumulators) ran the blocks it executed at 0.76 of the unmasked loop's rate.
  auto static_step = [&](auto mc, int kt) __attribute__((always_inline)) {
    constexpr unsigned MK = decltype(mc)::value;
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const double* sa = lds + cur * (2 * LA);
    const double* sb = sa + LA;
    // (a distinct marker per pattern: identical fragment reads at the head of the branches below must not be merged
    // above them -- see the note on the two loop bodies)
    asm volatile("; static K-step, block pattern %0" ::"n"(MK) : "memory");
    if (MK != 0u) {
      // in-place MFMAs (with builtin MFMAs the accumulators of the many bodies meet in PHI copies and spill); the
      // statements keep their order, so the fragments of quarter kk + 1 are requested before the MFMAs of quarter kk
      double fa[2][TM], fb[2][TN];
      auto frags = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          if (MK & (0xfu << (4 * i))) fa[kk & 1][i] = sa[(kk * 4 + fk) * LROW + (2 * i + wm) * 16 + fr];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (MK & (0x1111u << j)) fb[kk & 1][j] = sb[(kk * 4 + fk) * LROW + (2 * j + wn) * 16 + fr];
      };
      frags(0);
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        if (kk + 1 < BK / 4) frags(kk + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (MK & (1u << (i * 4 + j))) LRN_MFMA_INPLACE(acc[i][j], fa[kk & 1][i], fb[kk & 1][j]);
      }
    }
    __syncthreads();
  };
  if (!(d.flags & GEMM_LAB_NO_LOAD)) issue(kt0, kt0 & 1);
  __syncthreads();
  int kt = (d.flags & GEMM_LAB_NO_KLOOP) ? nk : kt0;
  if constexpr (DYN) {
    int head_end = kt0, tail_begin = nk;          // K-steps [kt0, head_end) and [tail_begin, nk) run block by block
    if (smask != 0xffffu) {
      head_end = nk;
    } else {
      if (from_n | from_m) head_end = kt0 + 8 < nk ? kt0 + 8 : nk;
      if (to_n) tail_begin = n0 / BK;
      if (to_m && m0 / BK < tail_begin) tail_begin = m0 / BK;
      if (d.flags & GEMM_NO_SKIP) { head_end = kt0; tail_begin = nk; }
      if (tail_begin < head_end) tail_begin = head_end;
      if (tail_begin > nk) tail_begin = nk;
    }
    for (; kt < head_end; ++kt) masked_step(kt, step_mask(kt));
    for (; kt < tail_begin; ++kt) fast_step(kt);
    for (; kt < nk; ++kt) masked_step(kt, step_mask(kt));
  } else {
    // A SEQUENCE of simple loops, one per pattern, each over its range of K-steps (empty for most tiles): as branches of
    // one loop body the accumulators of the bodies meet in copies and spill.
    //  * a tile whose 16 blocks are all needed, triangular operand (KFROM): wave column wn meets block column bn = 2 j + wn
    //    at K-step kt0 + bn -- no block during the first wn steps, then j < 1, 2, 3 for two steps each, then the
    //    unmasked loop; KFROM_M the same with block rows;
    //  * a tile with skipped blocks (edge, packed diagonal tile): the smallest pattern that contains its blocks, for all
    //    its K-steps (a block of the head computed too early multiplies stored zeros);
    //  * the K-steps of a KTO tail run unmasked (stored zeros again).
    int e[12];                                    // end of the K-steps of pattern q (see the loops below)
#pragma unroll
    for (int q = 0; q < 12; ++q) e[q] = kt0;
    if (smask == 0xffffu) {
      if ((from_n | from_m) && !(d.flags & GEMM_NO_SKIP)) {
        const int w0 = kt0 + (from_n ? wn : wm);
        e[0] = w0;
        if (from_n) { e[1] = w0 + 2; e[2] = w0 + 4; e[3] = w0 + 6; }
        else { e[4] = w0 + 2; e[5] = w0 + 4; e[6] = w0 + 6; }
      }
    } else if (!(d.flags & GEMM_NO_SKIP)) {
      // (smallest first)
      if ((smask & ~0x1111u) == 0u) e[1] = nk;
      else if ((smask & ~0x000fu) == 0u) e[4] = nk;
      else if ((smask & ~0x08ceu) == 0u) e[7] = nk;
      else if ((smask & ~0x7310u) == 0u) e[9] = nk;
      else if ((smask & ~0x3333u) == 0u) e[2] = nk;
      else if ((smask & ~0x00ffu) == 0u) e[5] = nk;
      else if ((smask & ~0x8cefu) == 0u) e[8] = nk;
      else if ((smask & ~0xf731u) == 0u) e[10] = nk;
      else if ((smask & ~0x7777u) == 0u) e[3] = nk;
      else if ((smask & ~0x0fffu) == 0u) e[6] = nk;
    }
#define LRN_PATTERN_LOOP(q, m)                                                             \
    {                                                                                      \
      const int end = e[q] < nk ? e[q] : nk;                                               \
      for (; kt < end; ++kt) static_step(std::integral_constant<unsigned, m>{}, kt);       \
    }
    LRN_PATTERN_LOOP(0, 0x0000u)
    LRN_PATTERN_LOOP(1, 0x1111u) LRN_PATTERN_LOOP(2, 0x3333u) LRN_PATTERN_LOOP(3, 0x7777u)     // block columns 0 .. NJ-1 of the wave
    LRN_PATTERN_LOOP(4, 0x000fu) LRN_PATTERN_LOOP(5, 0x00ffu) LRN_PATTERN_LOOP(6, 0x0fffu)     // block rows 0 .. NI-1
    LRN_PATTERN_LOOP(7, 0x08ceu) LRN_PATTERN_LOOP(8, 0x8cefu)                                   // packed diagonal tile: bn > bm, bn >= bm
    LRN_PATTERN_LOOP(9, 0x7310u) LRN_PATTERN_LOOP(10, 0xf731u)                                  // GEMM_DIAG_LOWER_Z: bm > bn, bm >= bn
#undef LRN_PATTERN_LOOP
    for (; kt < nk; ++kt) fast_step(kt);
  }
  LRN_MFMA_DRAIN();
  if (d.flags & GEMM_LAB_NO_STORE) return;

  const bool x2 = EPI && (d.flags & GEMM_OFFDIAG_X2) && (tm != tn);
  const double alpha = x2 ? 2.0 * d.alpha : d.alpha;
  const bool sq = EPI && (d.flags & GEMM_SQUARE);
  // packed lower C (kernel n = row, kernel m = column of the symmetric matrix; tiles with tn >= tm)
  const bool pk = EPI && (d.flags & GEMM_C_PACKED);
  const bool mir = EPI && (d.flags & GEMM_C_MIRROR) && (tm != tn);
  const int pkS = packed_S(d.pk_m);
  const long pkKd = packed_diag_elems(d.pk_m > 0 ? d.pk_m : 1);
  if (pk && !sq) {
    // Packed lower C, chunk-major: a 16x16 block (column block q, row block rb >= q of the symmetric matrix) is 16
    // chunks of 16 doubles, chunk(c, rb) = c for the diagonal block and Kd/16 + [q S - 8 q (q+1)] + cc (S/16 - q - 1) +
    // (rb - q - 1) below it (packed_lower_offset / 16 with c = 16 q + cc) -- block-uniform terms hoisted, two integer
    // operations per element (the generic offset function per element cost GEMM2' 3 % in its epilogue)
    const long S16 = pkS >> 4, kd16 = pkKd >> 4;
    const int qb = (m0 + d.pk_off) >> 4, rbb = (n0 + d.pk_off) >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long q = qb + 2 * i + wm;
      const long below = kd16 + q * pkS - 8 * q * (q + 1) - q - 1;      // + cc * (S16 - q - 1) + rb
      const long percol = S16 - q - 1;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!(smask & (1u << (i * 4 + j)))) continue;
        const long rb = rbb + 2 * j + wn;
        if (rb < q) continue;                     // 16x16 blocks on and below the diagonal only
        const int n = n0 + (2 * j + wn) * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = MFMA_F64_ROW(lane, r);
          const int m = m0 + (2 * i + wm) * 16 + cc;
          if (m < d.M && n < d.N) {
            const long chunk = rb == q ? 16 * q + cc : below + cc * percol + rb;
            Cg[chunk * d.pk_cstride + fr] = alpha * acc[i][j][r];
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!((smask | zmask) & (1u << (i * 4 + j)))) continue;      // (zmask: the accumulators were never touched -- zeros)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + (2 * i + wm) * 16 + MFMA_F64_ROW(lane, r);
        int n = n0 + (2 * j + wn) * 16 + fr;
        if (m < d.M && n < d.N) {
          double v = alpha * acc[i][j][r];
          if (sq) v = v * v;
          if (pk) {                               // (only with GEMM_SQUARE: the generic offset function)
            if ((n >> 4) >= (m >> 4)) {
              const long o = packed_lower_offset(n + d.pk_off, m + d.pk_off, pkS, pkKd);
              Cg[(o >> 4) * d.pk_cstride + (o & 15)] = v;
            }
            continue;
          }
          double* c = Cg + (long)m * d.sCm + (long)n * d.sCn;
          if (d.beta != 0.0) v += d.beta * (*c);
          *c = v;
          if (mir) Cg[(long)n * d.sCm + (long)m * d.sCn] = v;
          if (EPI && d.C2) d.C2[(long)n * d.sCm + (long)m * d.sCn] = v;      // transposed copy (square C, batch 1)
        }
      }
    }
}

// ------------------------------------------------------------------ mid-size products (round 4)
// One plain product whose 64 x 64 tiles do not fill the chip by themselves (msz 400 .. 1400: Newton-Schulz, the Lyapunov CG
// and the step-length products of C2 / C3, 70-90 per IP iteration): split-K over 2-4 workgroups per tile, slabs added by
// reduce_slabs.  On the register-staged kernel above (one K-tile of global loads in flight per workgroup) the product
// 801^3 took 33.5 us with the MFMA pipe 37 % busy -- every K-tile waits a global-load latency.  Here the operand panels go
// global -> LDS by DMA into FOUR LDS stages: the panels of K-tile t + 3 are requested before the MFMAs of K-tile t, one
// barrier per K-tile, the wait counts set by hand (s_waitcnt vmcnt(8): the DMA instructions of the two stages after this
// one may still be in flight).  16-byte DMA (`buffer_load_dwordx4 ... lds`) from rows that are only 8-byte aligned
// (msz = 801): measured correct on gfx950 (tools/probe_unaligned_dma.py; the first version of this kernel moved 4 bytes
// per lane and was bound by its 64 DMA instructions per K-tile: 41.5 us).  One DMA instruction fills two k-rows of the
// unpadded [k][64] image (lanes 0-31 / 32-63); the bank conflicts of the fragment reads are removed by an XOR of the 16-byte
// unit index with 8 (k & 3), applied to the source address and to the read.  Both operands contiguous along their non-K
// dimension, C contiguous along the kernel's n; 64 x 64 tile, 4 waves 2 x 2, 64 KB of LDS: two workgroups per CU.
__global__ __launch_bounds__(256, 2) void gemm_f64_mid_kernel(GemmParams p) {
  constexpr int BM = 64, BN = 64, NST = 3;
  constexpr int LA = BK * BM;                   // doubles per image
  __shared__ double lds[NST * 2 * LA];
  const GemmDesc& d = p.d;
  // 1-D grid, a multiple of 8 workgroups: XCD x = blockIdx.x % 8 takes the items [x q, (x + 1) q) of the (tile, split) list
  // (tile-major: the splits of a tile and the tiles that share panels stay in one L2); p.n1 = tiles x splits items.  No
  // workgroup that exits at once sits in the middle of the grid: the dispatcher then fills the CUs evenly (with a (tiles
  // padded to 8) x splits grid 14 CUs received three workgroups and 19 one: 29 us instead of 21)
  int tm, tn, ks;
  {
    const int q = gridDim.x >> 3;
    const int item = (blockIdx.x & 7) * q + (blockIdx.x >> 3);
    if (item >= p.n1) return;
    const int2 tt = p.tile_list[item / d.ksplit];
    tm = tt.x;
    tn = tt.y;
    ks = item % d.ksplit;
  }
  const int t = threadIdx.x, lane = t & 63;
  unsigned long long* trace = d.lab_trace ? d.lab_trace + 8 * ((long)blockIdx.x + (long)gridDim.x * blockIdx.z) : nullptr;
  if (trace && t == 0) {
    trace[0] = wall_clock64();
    trace[1] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lda = (int)d.sAk, ldb = (int)d.sBk;
  const unsigned bytesA = (unsigned)(((long)(d.K - 1) * lda + d.M) * 8);
  const unsigned bytesB = (unsigned)(((long)(d.K - 1) * ldb + d.N) * 8);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, bytesB, 0x00020000);
  // DMA instruction jj (0, 1) of this wave and stage fills the k-rows 2 (w + 4 jj) + (lane >> 5); lane & 31 is the 16-byte
  // unit of the LDS row, unit ^ 8 (k & 3) the unit of the source row ((2 w + (lane >> 5)) & 3 = k & 3 for both jj)
  const int krow = 2 * w + (lane >> 5);
  const int usrc = (lane & 31) ^ (8 * (krow & 3));
  const int ma = m0 + 2 * usrc, nb = n0 + 2 * usrc;
  const unsigned offA = ma < d.M ? (unsigned)ma * 8u : 0x80000000u;
  const unsigned offB = nb < d.N ? (unsigned)nb * 8u : 0x80000000u;
  const int k_lo = ks * p.kchunk;
  int k_hi = k_lo + p.kchunk;
  if (k_hi > d.K) k_hi = d.K;
  const int nkt = k_hi > k_lo ? (k_hi - k_lo + BK - 1) / BK : 0;
  // K-tile `it` of this split -> stage buf.  k-rows at or beyond k_hi belong to the next split (or lie beyond K): their
  // offsets are sent out of range -- zero fill.
  auto issue = [&](int it, int buf) {
    double* sa = lds + buf * (2 * LA);
    double* sb = sa + LA;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int k = k_lo + it * BK + 8 * jj + krow;
      const unsigned oob = k < k_hi ? 0u : 0x80000000u;
      const unsigned ka = (unsigned)k * (unsigned)lda * 8u, kb = (unsigned)k * (unsigned)ldb * 8u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + (2 * w + 8 * jj) * BM), 16,
                                               (int)((offA + ka) | oob), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + (2 * w + 8 * jj) * BN), 16,
                                               (int)((offB + kb) | oob), 0, 0, 0);
    }
  };
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const int fr = lane & 15, fk = lane >> 4;
  // fragment (k = 4 kk + fk, m = 16 b + fr) of an image: row k, unit ((m >> 1) ^ 8 (k & 3)) = doubles ((16 b) ^ (16 fk)) + fr
  int fa_off[2], fb_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fa_off[i] = fk * BM + (((wm * 32 + i * 16)) ^ (16 * fk)) + fr;
    fb_off[i] = fk * BN + (((wn * 32 + i * 16)) ^ (16 * fk)) + fr;
  }
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nkt) issue(s0, s0);
  for (int it = 0; it < nkt; ++it) {
    // stage `it` has landed (this wave's part; the barrier extends that to the workgroup) -- the 4 DMA instructions of each
    // of the stages it + 1, it + 2, issued after it, may still be in flight
    if (NST == 4 && it + 2 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (it + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (trace && t == 0 && it < 4) trace[2 + it] = wall_clock64();
    // every wave is past the MFMAs of K-tile it - 1: its stage is free for K-tile it + 3
    if (it + NST - 1 < nkt) issue(it + NST - 1, (it + NST - 1) % NST);
    const double* sa = lds + (it % NST) * (2 * LA);
    const double* sb = sa + LA;
    double fa[BK / 4][2], fb[BK / 4][2];
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[kk][i] = sa[kk * 4 * BM + fa_off[i]];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[kk][j] = sb[kk * 4 * BN + fb_off[j]];
    }
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
  }
  if (trace && t == 0) trace[6] = wall_clock64();
  double* __restrict__ Cg = d.C + (long)ks * d.sCs;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + MFMA_F64_ROW(lane, r);
        const int n = n0 + wn * 32 + j * 16 + fr;
        if (m < d.M && n < d.N) Cg[(long)m * d.sCm + n] = d.alpha * acc[i][j][r];
      }
  if (trace && t == 0) trace[7] = wall_clock64();
}

// ------------------------------------------------------------------ direct-to-LDS, K-contiguous
// GEMM3 of the Schur assembly: H[i,j] = <A_i, T_j>, both operands contiguous along K (the vec
// index of an msz x msz matrix), K walked in the lower-tile segments of GEMM_KSEG_TRI.
// One `global_load_lds_dwordx4` wave-instruction fills 8 rows x 16 k (128 B each) of the
// [m][k] image.  The image is unpadded (the DMA writes 1 KiB linearly), so the bank conflicts
// of the fragment reads are removed by an XOR swizzle of the 16-byte k-pair index with
// (row>>1)&7, applied to the SOURCE address and to the read (both-sides rule).  Rows beyond
// M/N are clamped to the last valid row (their results are never stored).
// FLAT (GEMM_KFLAT): K is the flat index of the packed lower-tile layout, walked in chunks of 16;
// kcols[] then holds chunk indices.
// TB = 16x16 blocks per wave and dimension: 4 -> the 128 x 128 tile; 5 -> a 160 x 160 tile (25 accumulators = 200 VGPRs per
// lane; two images of 160 x 16 doubles, double-buffered = 80 KB per workgroup -- two workgroups fill the CU's 160 KB of
// LDS exactly).  The larger tile moves 0.8 of the panel bytes per flop, spends 100 instead of 64 MFMAs per wave between
// barriers, and nvar = 4000 = 25 x 160 has no edge tiles at all.
// TBM x TBN = 4 x 5: the 128 x 160 tiles of a last tile row of height 160 (nvar = 4000 = 30 x 128 + 160: no edge tiles).
template <bool FLAT, int TBM, int TBN>
__global__ __launch_bounds__(256, 2) void gemm_f64_kseg_lds_kernel(GemmParams p) {
  constexpr int BM = 32 * TBM, BN = 32 * TBN, TM = TBM, TN = TBN;
  constexpr int LA = BM * BK, LB = BN * BK;     // doubles per image (unpadded)
  extern __shared__ double lds[];               // 2 * (LA + LB) doubles
  const GemmDesc& d = p.d;
  int tm, tn, ks, bz;
  if (p.n2 >= 0) {
    // regular tiles of all splits first, short tiles last (see GemmParams): the equally long workgroups keep their
    // lock-step through K, and the short ones fill the slots of the last round instead of a launch of their own
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int q1 = p.n1 >> 3, q2 = p.n2 >> 3;
    int2 tt;
    if (l < d.ksplit * q1) { ks = l / q1; tt = p.tile_list[xcd * q1 + l % q1]; }
    else { const int l2 = l - d.ksplit * q1; ks = l2 / q2; tt = p.tile_list2[xcd * q2 + l2 % q2]; }
    tm = tt.x;
    tn = tt.y;
    bz = 0;
  } else {
    int bid = blockIdx.x, nwg = gridDim.x;
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int2 tt = p.tile_list[swz];
    tm = tt.x;
    tn = tt.y;
    ks = blockIdx.z % d.ksplit;
    bz = blockIdx.z / d.ksplit;
  }
  if (tm < 0) return;      // padding entry of the tile list
  const double* Ag = d.A + (long)bz * d.bA;
  const double* Bg = d.B + (long)bz * d.bB;
  double* __restrict__ Cg = d.C + (long)bz * d.bC + (long)ks * d.sCs;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int m0 = p.m_org + tm * BM, n0 = p.n_org + tn * BN;
  // staging geometry: instruction j of wave w covers image rows 8*(w + 4j) .. +7
  const int lrow = lane >> 3, lpair = lane & 7;
  const double* pa[TBM];
  const double* pb[TBN];
#pragma unroll
  for (int j = 0; j < TBM; ++j) {
    int row = 8 * (w + 4 * j) + lrow;                  // tile-local row 0 .. BM-1
    int src_pair = lpair ^ ((row >> 1) & 7);           // swizzle on the source
    int ra = m0 + row;
    if (ra >= d.M) ra = d.M - 1;
    pa[j] = Ag + (long)ra * d.sAm + 2 * src_pair;
  }
#pragma unroll
  for (int j = 0; j < TBN; ++j) {
    int row = 8 * (w + 4 * j) + lrow;
    int src_pair = lpair ^ ((row >> 1) & 7);
    int rb = n0 + row;
    if (rb >= d.N) rb = d.N - 1;
    pb[j] = Bg + (long)rb * d.sBn + 2 * src_pair;
  }
  int segc = p.kcols[ks], segcend = FLAT ? p.kcols2[ks] : p.kcols[ks + 1];
  int segr = FLAT ? 0 : (segc / 128) * 128;
  const int ld = d.kseg_ld;
  auto chunk_base = [&]() -> long { return FLAT ? (long)segc * d.kflat_cstride : (long)segc * ld + segr; };
  // FLAT: chunks of this split not yet issued.  Workgroups that share a panel (same tm or same tn) run in
  // lock-step -- all MFMA-bound at the same rate -- and would ask L2 for the same line within the miss latency
  // of the first request; starting each one ((tm + tn) & 7) * kstagger chunks into the split (a Latin square:
  // distinct along a row and along a column of the super-tile) and wrapping around spreads them in time.
  // Measured at C4: the L2 hit rate FALLS (11 % -> 4 %) and the kernel slows by 3-6 %; delaying the workgroups by
  // fractions of a K-step instead (s_sleep at the start) changes nothing.  Off by default, kept as a knob.
  const int seg0 = segc;
  int left = segcend - segc;
  if (FLAT && d.kstagger > 0 && left > 0) segc = seg0 + (int)(((long)((tm + tn) & 7) * d.kstagger) % left);

  v4f64 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  // a diagonal tile of a symmetric rank-k update (same operand, same rows): ONE panel serves as both images -- half
  // the loads of the tiles that, computing 10 of their 16 blocks, would otherwise be bound by the panel traffic
  const bool same_panel = (TBM == TBN) && (Ag == Bg) && (m0 == n0) && (d.sAm == d.sBn) && (d.M == d.N);
  const long lab_kb = (d.flags & GEMM_LAB_SAME_CHUNK) ? chunk_base() : -1;       // (measurement only)
  auto issue = [&](long kb, int buf) {
    if (lab_kb >= 0) kb = lab_kb;
    double* sa = lds + buf * (LA + LB);
    double* sb = sa + LA;
#pragma unroll
    for (int j = 0; j < (TBM > TBN ? TBM : TBN); ++j) {
      const int r8 = 8 * (w + 4 * j);
      if (j < TBM)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[j < TBM ? j : 0] + kb),
                                         (__attribute__((address_space(3))) void*)(sa + r8 * BK), 16, 0, 0);
      if (j < TBN && !same_panel)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[j < TBN ? j : 0] + kb),
                                         (__attribute__((address_space(3))) void*)(sb + r8 * BK), 16, 0, 0);
    }
  };
  auto next_chunk = [&]() {
    if (FLAT) {
      --left;
      if (++segc == segcend) segc = seg0;
      return;
    }
    segr += BK;
    if (segr >= ld) { ++segc; segr = (segc / 128) * 128; }
  };

  const int fr = lane & 15, fk = lane >> 4;
  // blocks of this wave: (bm, bn) = (2 i + wm, 2 j + wn), interleaved like in gemm_f64_lds_kernel, so that the
  // blocks beyond M / N (nvar = 4000: the last tile row holds 32 of 128 rows) and, with GEMM_DIAG_LOWER, the blocks
  // above the diagonal of a diagonal tile are skipped evenly by the four waves
  unsigned smask = 0;
  {
    const bool diag_lo = (d.flags & GEMM_DIAG_LOWER) && tm == tn;
    const bool diag_up = (d.flags & GEMM_DIAG_UPPER) && tm == tn;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int bm = 2 * i + wm, bn = 2 * j + wn;
        bool need = (m0 + 16 * bm < d.M) && (n0 + 16 * bn < d.N);
        if ((diag_lo && bm < bn) || (diag_up && bn < bm)) need = false;
        if (d.flags & GEMM_NO_SKIP) need = true;
        if (need) smask |= 1u << (i * TN + j);
      }
  }
  bool more = FLAT ? left > 0 : segc < segcend;
  if (more) {
    issue(chunk_base(), 0);
    next_chunk();
  }
  __syncthreads();
  int cur = 0;
  // two copies of the K loop (see gemm_f64_lds_kernel): the unmasked one for tiles whose 16 blocks are all needed,
  // the masked one for edge and diagonal tiles
  if (smask == (1u << (TM * TN)) - 1u) {
    while (more) {
      const bool have_next = FLAT ? left > 0 : segc < segcend;
      if (have_next) {
        issue(chunk_base(), cur ^ 1);
        next_chunk();
      }
      const double* sa = lds + cur * (LA + LB);
      const double* sb = same_panel ? sa : sa + LA;
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        double fa[TM], fb[TN];
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          int row = (2 * i + wm) * 16 + fr;
          fa[i] = sa[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          int row = (2 * j + wn) * 16 + fr;
          fb[j] = sb[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      cur ^= 1;
      more = have_next;
    }
  } else {
    while (more) {
      const bool have_next = FLAT ? left > 0 : segc < segcend;
      if (have_next) {
        issue(chunk_base(), cur ^ 1);
        next_chunk();
      }
      const double* sa = lds + cur * (LA + LB);
      const double* sb = same_panel ? sa : sa + LA;
      if (smask != 0u) {
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
          double fa[TM], fb[TN];
          const int k = kk * 4 + fk;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            int row = (2 * i + wm) * 16 + fr;
            fa[i] = sa[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)];
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            int row = (2 * j + wn) * 16 + fr;
            fb[j] = sb[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)];
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              if (smask & (1u << (i * TN + j))) LRN_MFMA_INPLACE(acc[i][j], fa[i], fb[j]);
        }
      }
      __syncthreads();
      cur ^= 1;
      more = have_next;
    }
  }
  LRN_MFMA_DRAIN();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!(smask & (1u << (i * TN + j)))) continue;       // (GEMM_DIAG_LOWER: those slab entries are never read)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + (2 * i + wm) * 16 + MFMA_F64_ROW(lane, r);
        int n = n0 + (2 * j + wn) * 16 + fr;
        if (m < d.M && n < d.N) Cg[(long)m * d.sCm + (long)n * d.sCn] = d.alpha * acc[i][j][r];
      }
    }
}

static bool kseg_lds_path_ok(const GemmDesc& d) {
  if (d.sAk != 1 || d.sBk != 1 || d.beta != 0.0) return false;
  if ((d.flags & GEMM_KFLAT) ? ((d.kflat_total & 15) || (d.kflat_diag & 15)) : (d.kseg_ld & 15)) return false;
  if ((d.sAm & 1) || (d.sBn & 1) || (d.bA & 1) || (d.bB & 1)) return false;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return false;
  if (d.flags & (GEMM_OFFDIAG_X2 | GEMM_SQUARE)) return false;
  return true;
}

static bool lds_path_ok(const GemmDesc& d) {
  if (d.sAm != 1 || d.sBn != 1 || d.ksplit != 1) return false;
  // (round 4: rows that are only 8-byte aligned -- odd leading dimensions -- are fine: the 16-byte DMA was measured correct
  // from them on gfx950, tools/probe_unaligned_dma.py and test_gpu_blocks.py; LRN_LDS_ALIGNED=1 restores the old rule)
  static const bool aligned_only = getenv("LRN_LDS_ALIGNED") != nullptr;
  if (d.sAk < d.M || d.sBk < d.N || (aligned_only && ((d.sAk & 1) || (d.sBk & 1)))) return false;
  if (aligned_only && (((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15) || (d.bA & 1) || (d.bB & 1))) return false;
  // short-K products (Cholesky panel / trailing updates) stay on the generic kernel: the DMA
  // pipeline needs a long K loop to pay off, and it keeps this kernel's profile = the assembly GEMMs
  if (d.K < 256) return false;
  // 32-bit byte offsets inside one operand matrix
  if ((double)d.K * (double)d.sAk * 8.0 >= 2.0e9 || (double)d.K * (double)d.sBk * 8.0 >= 2.0e9) return false;
  return true;
}

// (tm, tn) enumeration: 8x8 super-tiles (tm fastest inside), only the tiles a TRI flag keeps.
// Consecutive list entries share operand panels, and the XCD swizzle hands each XCD a
// contiguous run of the list, so co-resident workgroups of one L2 re-use panels.
// `short_sel`: 0 all tiles; 1 only the tiles whose 16 blocks are all computed ("regular"); 2 only the others -- the
// tiles of the last tile row / column when `edge_m` / `edge_n` (M, N not multiples of the tile) and the diagonal
// tiles when `diag` (GEMM_DIAG_*).  The K-contiguous rank-k update runs them as two launches: its workgroups are all
// equally long and advance in lock-step through K (that is where its L2 hits come from); shorter workgroups mixed in
// break the step -- measured at C4: 496 -> 520 ms although 7 % of the MFMAs were skipped.
static const int2* get_tile_list(int tilesM, int tilesN, int tri, int* count, int short_sel = 0, bool edge_m = false,
                                 bool edge_n = false, bool diag = false, int korder = 0) {
  struct Key {
    int a, b, c;
    bool operator<(const Key& o) const { return a != o.a ? a < o.a : (b != o.b ? b < o.b : c < o.c); }
  };
  tri |= short_sel << 8 | (edge_m ? 1 << 12 : 0) | (edge_n ? 1 << 13 : 0) | (diag ? 1 << 14 : 0) | korder << 16;
  struct Val { int2* dev; int n; };
  static std::map<Key, Val> cache[16];
  static std::mutex mu;                       // contexts on several host threads share the cache
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto& cm = cache[dev & 15];
  Key k{tilesM, tilesN, tri};
  auto it = cm.find(k);
  if (it != cm.end()) { *count = it->second.n; return it->second.dev; }
  std::vector<int2> v;
  const int GS = 8;
  for (int sn = 0; sn < tilesN; sn += GS)
    for (int sm = 0; sm < tilesM; sm += GS)
      for (int tn = sn; tn < sn + GS && tn < tilesN; ++tn)
        for (int tm = sm; tm < sm + GS && tm < tilesM; ++tm) {
          if ((tri & 3) == GEMM_TRI_LOWER && tn > tm) continue;
          if ((tri & 3) == GEMM_TRI_UPPER && tm > tn) continue;
          const bool shrt = (edge_m && tm == tilesM - 1) || (edge_n && tn == tilesN - 1) || (diag && tm == tn);
          if ((short_sel == 1 && shrt) || (short_sel == 2 && !shrt)) continue;
          v.push_back(make_int2(tm, tn));
        }
  if (korder) {
    // (measurement, LRN_TILE_ORDER: the tiles of a triangular-K product by K length -- korder 1 / 2: K grows with tn / tm
    // downwards -- longest, shortest, second longest, second shortest, ...)
    std::vector<int2> srt = v, mix;
    std::stable_sort(srt.begin(), srt.end(), [&](const int2& a, const int2& b) { return korder == 1 ? a.y < b.y : a.x < b.x; });
    for (size_t i = 0, j = srt.size(); i < j;) {
      mix.push_back(srt[i++]);
      if (i < j) mix.push_back(srt[--j]);
    }
    v.swap(mix);
  }
  if (short_sel == 2 && diag) {
    // every XCD walks one contiguous run of the list: deal the diagonal tiles (10 of 16 blocks per wave) and the
    // edge tiles (a quarter of the rows) alternately, so that the runs are equally long
    std::vector<int2> dg, ed, mix;
    for (auto& t : v) (t.x == t.y ? dg : ed).push_back(t);
    for (size_t i = 0; i < std::max(dg.size(), ed.size()); ++i) {
      if (i < dg.size()) mix.push_back(dg[i]);
      if (i < ed.size()) mix.push_back(ed[i]);
    }
    v.swap(mix);
  }
  // The kernels take blockIdx.x % 8 for the XCD of a workgroup (flat id % 8 in hardware): true for every z-slice of
  // the grid only if the list length is a multiple of 8 -- pad with entries whose workgroups exit at once.
  const int n_real = (int)v.size();
  while (n_real > 0 && (v.size() & 7)) v.push_back(make_int2(-1, -1));
  Val val{nullptr, (int)v.size()};
  if (hipMalloc(&val.dev, sizeof(int2) * (v.size() + 1)) != hipSuccess) { *count = 0; return nullptr; }
  (void)hipMemcpy(val.dev, v.data(), sizeof(int2) * v.size(), hipMemcpyHostToDevice);
  cm[k] = val;
  *count = val.n;
  return val.dev;
}

// gemm() has no context to report into: the reason of its last failure on this thread, appended by
// lrn_last_error (api.hip) so that a bare LRN_ERR_ARG from deep inside a driver is not a stale message
static thread_local const char* tls_gemm_error = nullptr;
const char* gemm_last_error() { return tls_gemm_error; }
static int gemm_fail(int code, const char* why) { tls_gemm_error = why; return code; }

static int gemm_impl(hipStream_t st, const GemmDesc& din);
// hipFuncSetAttribute is per device: one process may drive several (one context per GPU)
static bool big_tile_attr_ok() {
  static bool done[64] = {false}, ok[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!done[dev]) {
    ok[dev] = hipFuncSetAttribute((const void*)gemm_f64_kseg_lds_kernel<true, 5, 5>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 160 * BK * 8) == hipSuccess &&
              hipFuncSetAttribute((const void*)gemm_f64_kseg_lds_kernel<true, 4, 5>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 160) * BK * 8) == hipSuccess;
    done[dev] = true;
  }
  return ok[dev];
}

// ---- mid-size products (round 3): a plain product whose 64 x 64 tiles do not fill the chip (msz ~ 800: 169 tiles on 256 CUs,
// one workgroup per CU, one wave per SIMD) is bound by the latency of its global loads -- 45 us for 801^3 where the MFMAs
// need 21.  Splitting K over 2-4 workgroups per tile puts several workgroups on every CU (their loads overlap each other's
// MFMAs); the slabs are added in a fixed order by reduce_slabs.  Slab memory: one buffer per stream (products on different
// streams run concurrently), grown on demand, kept for the life of the process.
#include <map>
#include <mutex>
static double* split_slabs(hipStream_t st, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  auto& e = cache[{dev, st}];
  if (e.second < bytes) {
    if (e.first) { (void)hipStreamSynchronize(st); (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
    if (hipMalloc(&e.first, bytes) != hipSuccess) { e.first = nullptr; return nullptr; }
    e.second = bytes;
  }
  return static_cast<double*>(e.first);
}

static int auto_split_factor(const GemmDesc& d) {
  static const int forced = getenv("LRN_GEMM_SPLIT") ? atoi(getenv("LRN_GEMM_SPLIT")) : -1;    // 0 / 1: off; k: k slabs
  if (forced == 0 || forced == 1) return 1;
  // (flags: none, or GEMM_TRI_LOWER alone -- a symmetric product of which only the lower 64-tiles are wanted, gemm_slabs)
  const bool tril = d.flags == GEMM_TRI_LOWER && d.M == d.N;
  if (d.ksplit > 1 || d.batch != 1 || (d.flags != 0 && !tril) || d.C2 || d.K < 256 || d.M < 128 || d.N < 128) return 1;
  const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
  if (t128 >= 256) return 1;                                   // the 128-tile kernels fill the chip by themselves
  const long tm64 = (d.M + 63) / 64;
  const long t64 = tril ? tm64 * (tm64 + 1) / 2 : tm64 * ((d.N + 63) / 64);
  // C dense and contiguous (reduce_slabs adds flat vectors)
  const long a = d.sCm < 0 ? -d.sCm : d.sCm, b = d.sCn < 0 ? -d.sCn : d.sCn;
  if (!((a == 1 && b == d.M) || (b == 1 && a == d.N))) return 1;
  if (forced > 1) return forced > 8 ? 8 : forced;
  // Round 4: the factor that minimises a small cost model of gemm_f64_mid_kernel (us; calibrated at msz 640 .. 1400,
  // tools/gemm_nt_times.py): a CU shares its MFMA pipe between the workgroups it holds (three fit), one K-tile of one
  // workgroup costs 0.515 us of it (1.5 x that when the workgroup is alone on its CU), plus launch and epilogue, plus -- for
  // slabs -- the pass that adds them ((ks + 1) M N doubles at 4 TB/s and a launch).
  int best = 1;
  double best_us = 1e300;
  for (int ks = 1; ks <= (tril ? 6 : 4); ++ks) {
    if (ks > 1 && d.K / ks < 96) break;
    const long wgs = t64 * ks;
    const long per_cu = (wgs + 255) / 256;
    const double ktiles = std::ceil((double)d.K / 16.0 / ks);
    double us = (double)per_cu * ktiles * 0.515 * (per_cu == 1 ? 1.5 : 1.0) + 5.0;
    if (ks > 1) us += 4.0 + (double)(ks + 1) * (double)d.M * (double)d.N * 8.0 / 4.0e6;
    if (us < best_us) { best_us = us; best = ks; }
  }
  return best;
}

static int gemm_maybe_slabs(hipStream_t st, const GemmDesc& din, SlabSrc* out) {
  tls_gemm_error = nullptr;            // (a stale reason must not be appended to a later, unrelated error)
  const int ks = auto_split_factor(din);
  if (din.flags == GEMM_TRI_LOWER && out && (ks < 2 || din.beta != 0.0)) {
    out->p = nullptr; out->stride = 0; out->n = 0;     // (lower tiles as slabs or not at all: the caller takes its other route)
    return LRN_OK;
  }
  if (ks > 1) {
    const size_t mn = (size_t)din.M * din.N;
    double* slabs = split_slabs(st, mn * ks * 8);
    if (slabs) {
      GemmDesc d2 = din;
      d2.C = slabs; d2.sCs = (long)mn; d2.ksplit = ks; d2.beta = 0.0;
      d2.flags |= GEMM_SMALL_TILE;
      const int rc2 = gemm_impl(st, d2);
      if (rc2 != LRN_OK) { if (!tls_gemm_error) tls_gemm_error = "kernel launch failed"; return rc2; }
      if (out && din.beta == 0.0) {
        out->p = slabs; out->stride = (long)mn; out->n = ks;
        return LRN_OK;
      }
      if (out) { out->p = din.C; out->stride = 0; out->n = 1; }
      return reduce_slabs(st, slabs, (long)mn, ks, din.C, (long)mn, din.beta);
    }
  }
  if (out) { out->p = din.C; out->stride = 0; out->n = 1; }
  const int rc = gemm_impl(st, din);
  if (rc != LRN_OK && !tls_gemm_error) tls_gemm_error = "kernel launch failed";
  return rc;
}

int gemm(hipStream_t st, const GemmDesc& din) { return gemm_maybe_slabs(st, din, nullptr); }
int gemm_slabs(hipStream_t st, const GemmDesc& din, SlabSrc* out) { return gemm_maybe_slabs(st, din, out); }

static int gemm_impl(hipStream_t st, const GemmDesc& din) {
  GemmParams p;
  p.d = din;
  GemmDesc& d = p.d;
  if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return LRN_OK;
  if (d.ksplit < 1) d.ksplit = 1;
  if (d.ksplit > MAX_KSPLIT) return gemm_fail(LRN_ERR_ARG, "gemm: d.ksplit > MAX_KSPLIT");
  // Orient so that the kernel's n (lane-contiguous in the MFMA result) is the contiguous
  // dimension of C: C^T = B^T A^T.
  long asCm = d.sCm < 0 ? -d.sCm : d.sCm, asCn = d.sCn < 0 ? -d.sCn : d.sCn;
  bool swapped = false;
  if (asCm < asCn) {
    swapped = true;
    if (d.flags & GEMM_KFROM_N) d.flags = (d.flags & ~GEMM_KFROM_N) | GEMM_KFROM_M;
    else if (d.flags & GEMM_KFROM_M) d.flags = (d.flags & ~GEMM_KFROM_M) | GEMM_KFROM_N;
    if (d.flags & GEMM_KTO_N) d.flags = (d.flags & ~GEMM_KTO_N) | GEMM_KTO_M;
    else if (d.flags & GEMM_KTO_M) d.flags = (d.flags & ~GEMM_KTO_M) | GEMM_KTO_N;
    std::swap(d.A, d.B);
    std::swap(d.bA, d.bB);
    long sAm = d.sBn, sAk = d.sBk, sBk = d.sAk, sBn = d.sAm;
    d.sAm = sAm; d.sAk = sAk; d.sBk = sBk; d.sBn = sBn;
    std::swap(d.sCm, d.sCn);
    std::swap(d.M, d.N);
    if (d.flags & GEMM_TRI_LOWER) d.flags = (d.flags & ~GEMM_TRI_LOWER) | GEMM_TRI_UPPER;
    else if (d.flags & GEMM_TRI_UPPER) d.flags = (d.flags & ~GEMM_TRI_UPPER) | GEMM_TRI_LOWER;
    if (d.flags & GEMM_DIAG_LOWER) d.flags = (d.flags & ~GEMM_DIAG_LOWER) | GEMM_DIAG_UPPER;
    else if (d.flags & GEMM_DIAG_UPPER) d.flags = (d.flags & ~GEMM_DIAG_UPPER) | GEMM_DIAG_LOWER;
    d.flags &= ~GEMM_DIAG_LOWER_Z;                 // (a hint for the unswapped orientation only)
  }
  const bool tri = d.flags & (GEMM_TRI_LOWER | GEMM_TRI_UPPER);
  const bool kflat = d.flags & GEMM_KFLAT;
  const bool kseg = (d.flags & GEMM_KSEG_TRI) || kflat;
  const bool kfrom = d.flags & (GEMM_KFROM_N | GEMM_KFROM_M | GEMM_KTO_N | GEMM_KTO_M);   // triangular K ranges
  if ((d.flags & GEMM_C_MIRROR) && (d.M != d.N || !tri || d.beta != 0.0)) return gemm_fail(LRN_ERR_ARG, "gemm: (d.flags & GEMM_C_MIRROR) && (d.M != d.N || !tri || d.beta != 0.0)");
  // triangular K ranges: the operand that is triangular spans K (a trailing sub-block may be narrower in the other
  // dimension: columns [c0, c1) of a product with the trailing block of the factor)
  if (kfrom && (d.ksplit != 1 || kseg)) return gemm_fail(LRN_ERR_ARG, "gemm: kfrom && (d.ksplit != 1 || kseg)");
  if ((d.flags & (GEMM_KFROM_N | GEMM_KTO_N)) && d.N > d.K) return gemm_fail(LRN_ERR_ARG, "gemm: (d.flags & (GEMM_KFROM_N | GEMM_KTO_N)) && d.N > d.K");
  if ((d.flags & (GEMM_KFROM_M | GEMM_KTO_M)) && d.M > d.K) return gemm_fail(LRN_ERR_ARG, "gemm: (d.flags & (GEMM_KFROM_M | GEMM_KTO_M)) && d.M > d.K");
  if ((d.flags & GEMM_C_PACKED) && (d.pk_off & 15)) return gemm_fail(LRN_ERR_ARG, "gemm: GEMM_C_PACKED needs pk_off % 16 == 0 (block width of the packed layout)");
  if ((d.flags & GEMM_C_PACKED) && (!swapped || d.pk_m <= 0 || d.beta != 0.0)) return gemm_fail(LRN_ERR_ARG, "gemm: (d.flags & GEMM_C_PACKED) && (!swapped || d.pk_m <= 0 || d.beta != 0.0)");
  // tile choice: 128x128 unless the problem is too small to fill the chip with it
  long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch * d.ksplit;
  static const int mid_mode = getenv("LRN_GEMM_MID") ? atoi(getenv("LRN_GEMM_MID")) : 1;      // (measurement knob; 0: off)
  bool small = (d.flags & GEMM_SMALL_TILE) ||
               (t128 < 256 && !(d.flags & (GEMM_OFFDIAG_X2 | GEMM_C_PACKED | GEMM_C_MIRROR)) && !kseg && !kfrom);
  // (measurement, LRN_GEMM_MID=2: plain products of up to 1024 128-tiles on the 64-tile DMA kernel as well)
  if (mid_mode == 2 && t128 < 1024 && d.flags == 0 && !d.C2 && d.batch == 1 && d.beta == 0.0) small = true;
  // Round 4: one plain product of 256 .. 1023 128-tiles (msz 2000 .. 4000) does not fill whole rounds of the 512 workgroup
  // slots with 128-tiles -- msz 3000: 576 tiles, the 64 of the second round run alone on their CUs, 1226 us where the
  // 64-tile DMA kernel (2209 tiles on 768 slots) takes 1011.  Both kernels priced by the round model that fits
  // tools/gemm_nt_times.py (its decisions match the measurements at msz 2100 .. 3800); the 128-tile kernel keeps the tie.
  if (mid_mode == 1 && !small && t128 >= 256 && t128 < 1024 && d.flags == 0 && !d.C2 && d.batch == 1 && d.ksplit == 1 &&
      d.beta == 0.0 && d.sAm == 1 && d.sBn == 1 && d.sCn == 1 && d.M == d.N) {
    const double ks = (double)((d.K + BK - 1) / BK);
    const long full = t128 / 512, rem = t128 % 512;
    const double big_us = ks * 1.85 * (2.0 * full + (rem == 0 ? 0.0 : (rem <= 256 ? 1.25 : 2.0)));
    const long t64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
    const long mfull = t64 / 768, mrem = t64 % 768, per_cu = (mrem + 255) / 256;
    const double mid_us = mfull * 3.0 * ks * 0.515 + (mrem ? per_cu * ks * 0.515 * (per_cu == 1 ? 1.5 : 1.0) : 0.0);
    if (mid_us * 1.08 < big_us * 1.03) small = true;
  }
  const bool big = kflat && (d.flags & GEMM_TILE160);       // 160 x 160 tile of the K-contiguous rank-k update
  const int BMv = small ? 64 : (big ? 160 : 128);
  p.tilesM = (d.M + BMv - 1) / BMv;
  p.tilesN = (d.N + BMv - 1) / BMv;
  if (kflat) {
    // chunk (16 doubles) boundaries per split: kflat_nsd splits over the diagonal region, the rest over
    // the strictly-lower region (the caller weights their slabs by 2)
    if (d.kflat_total <= 0 || d.kflat_diag <= 0 || d.kflat_diag > d.kflat_total || d.kflat_nsd < 1 ||
        d.kflat_nsd > d.ksplit || d.kflat_cstride < 16 || (d.kflat_cstride & 1) || !kseg_lds_path_ok(d)) return gemm_fail(LRN_ERR_ARG, "gemm: d.kflat_total <= 0 || d.kflat_diag <= 0 || d.kflat_diag > d.kflat_total || d.kflat_nsd < 1 || d.kflat_nsd > d.ksplit || d.kflat_cstride < 16 || (d.kflat_cstride & 1) || !kseg_lds_path_ok(d)");
      d.K = (int)(d.kflat_total > 0x7fffffff ? 0x7fffffff : d.kflat_total);
    const long cd = d.kflat_diag / BK, ct = d.kflat_total / BK;
    // split s walks the chunks [kcols[s], kcols2[s])
    if (d.kflat_kb && d.kflat_ke) {
      for (int s = 0; s < d.ksplit; ++s) {
        if (d.kflat_kb[s] < 0 || d.kflat_ke[s] < d.kflat_kb[s] || d.kflat_ke[s] > ct) return gemm_fail(LRN_ERR_ARG, "gemm: d.kflat_kb[s] < 0 || d.kflat_ke[s] < d.kflat_kb[s] || d.kflat_ke[s] > ct");
        p.kcols[s] = d.kflat_kb[s];
        p.kcols2[s] = d.kflat_ke[s];
      }
    } else {
      const int nsd = d.kflat_nsd, nso = d.ksplit - nsd;
      if ((nso == 0) != (ct == cd)) return gemm_fail(LRN_ERR_ARG, "gemm: (nso == 0) != (ct == cd)");
      for (int s = 0; s < nsd; ++s) { p.kcols[s] = (int)(cd * s / nsd); p.kcols2[s] = (int)(cd * (s + 1) / nsd); }
      for (int s = 0; s < nso; ++s) {
        p.kcols[nsd + s] = (int)(cd + (ct - cd) * s / nso);
        p.kcols2[nsd + s] = (int)(cd + (ct - cd) * (s + 1) / nso);
      }
    }
    p.kchunk = 0;
  } else if (kseg) {
    if (d.kseg_ld <= 0 || d.kseg_cols <= 0) return gemm_fail(LRN_ERR_ARG, "gemm: d.kseg_ld <= 0 || d.kseg_cols <= 0");
    d.K = d.kseg_ld * d.kseg_cols;
    // balance splits by segment length sum
    double total = 0;
    for (int c = 0; c < d.kseg_cols; ++c) total += d.kseg_ld - (c / 128) * 128;
    double accw = 0;
    int s = 1;
    p.kcols[0] = 0;
    for (int c = 0; c < d.kseg_cols && s < d.ksplit; ++c) {
      accw += d.kseg_ld - (c / 128) * 128;
      if (accw >= total * s / d.ksplit) p.kcols[s++] = c + 1;
    }
    for (; s <= d.ksplit; ++s) p.kcols[s] = d.kseg_cols;
    p.kchunk = 0;
  } else {
    long per = (d.K + d.ksplit - 1) / d.ksplit;
    per = ((per + BK - 1) / BK) * BK;
    if (per < BK) per = BK;
    p.kchunk = (int)per;
  }
  int ntile = 0;
  if (d.tile_class >= 4 && !kflat) return gemm_fail(LRN_ERR_ARG, "gemm: tile_class 4 / 5 are for the K-contiguous rank-k update");
  if (d.tile_class != 0 && small) return gemm_fail(LRN_ERR_ARG, "gemm: tile_class needs the 128 tile");
  p.tile_list2 = nullptr;
  p.n1 = 0;
  p.n2 = -1;
  if (d.tile_class == 3) {
    if (!kflat || d.batch != 1) return gemm_fail(LRN_ERR_ARG, "gemm: tile_class 3 is for the K-contiguous rank-k update");
    const int trif = d.flags & (GEMM_TRI_LOWER | GEMM_TRI_UPPER);
    const bool em = (d.M % BMv) != 0, en = (d.N % BMv) != 0, dg = (d.flags & (GEMM_DIAG_LOWER | GEMM_DIAG_UPPER)) != 0;
    int c1 = 0, c2 = 0;
    p.tile_list = get_tile_list(p.tilesM, p.tilesN, trif, &c1, 1, em, en, dg);
    p.tile_list2 = get_tile_list(p.tilesM, p.tilesN, trif, &c2, 2, em, en, dg);
    if (!p.tile_list || !p.tile_list2) return gemm_fail(LRN_ERR_NOMEM, "gemm: tile list allocation failed");
    p.n1 = c1;
    p.n2 = c2;
    if (c1 + c2 == 0) return LRN_OK;
    if (c1 == 0) { p.tile_list = p.tile_list2; p.n1 = c2; p.n2 = 0; }     // (only short tiles: they are the first list)
    const long wgs = 8L * d.ksplit * ((p.n1 >> 3) + (p.n2 >> 3));
    if (wgs > 0x7fffffffL) return gemm_fail(LRN_ERR_ARG, "gemm: grid too large");
    dim3 grid1((unsigned)wgs, 1, 1);
    if (big) {
      if (!big_tile_attr_ok()) return gemm_fail(LRN_ERR_HIP, "gemm: 80 KB of dynamic LDS refused");
      hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<true, 5, 5>), grid1, dim3(256), 2 * 2 * 160 * BK * 8, st, p);
    } else {
      hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<true, 4, 4>), grid1, dim3(256), 2 * 2 * 128 * BK * 8, st, p);
    }
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  if (d.tile_class == 4 || d.tile_class == 5) {
    // The symmetric rank-k update with a LAST TILE ROW OF HEIGHT 128 + r, r = M % 128 in (0, 32] (nvar = 4000 = 30 x 128 +
    // 160): instead of a row of edge tiles that hold r of 128 rows -- each costs 0.7 of a full tile, bound by its panel
    // traffic -- the last 128 + r rows are tiled on their own by 128 x 160 tiles (<true, 4, 5>), the diagonal corner
    // included (two of those tiles, the second one with r of its 128 rows).  Orientation after the swap above: upper
    // tiles (tn >= tm), the strip is the last tile COLUMN.  tile_class 4 launches the leading part, 5 the strip (two
    // calls, so that the caller can time them apart).  Two launches in the caller's stream; side by side on two
    // streams the leading part loses its lock-step (measured at C4: 502 against 488 ms as one launch with edge tiles,
    // 486 as two launches in sequence with a corner launch of its own, which this form saves).
    const int rem = d.M % 128;
    if (!kflat || d.batch != 1 || big || d.M != d.N || rem == 0 || rem > 32 || d.M < 288 ||
        (d.flags & (GEMM_TRI_LOWER | GEMM_TRI_UPPER)) != GEMM_TRI_UPPER || !(d.flags & GEMM_DIAG_UPPER) ||
        (d.flags & GEMM_NO_SKIP))
      return gemm_fail(LRN_ERR_ARG, "gemm: tile_class 4 needs the symmetric K-contiguous update with M % 128 in (0, 32]");
    if (!big_tile_attr_ok()) return gemm_fail(LRN_ERR_HIP, "gemm: 80 KB of dynamic LDS refused");
    const int Mm = d.M - 128 - rem, tmain = Mm / 128;
    if (d.tile_class == 4) {   // leading part: regular tiles of every split first, its diagonal tiles last (as tile_class 3)
      GemmParams pm = p;
      pm.d.M = Mm; pm.d.N = Mm;
      pm.tilesM = tmain; pm.tilesN = tmain;
      int c1 = 0, c2 = 0;
      pm.tile_list = get_tile_list(tmain, tmain, GEMM_TRI_UPPER, &c1, 1, false, false, true);
      pm.tile_list2 = get_tile_list(tmain, tmain, GEMM_TRI_UPPER, &c2, 2, false, false, true);
      if (!pm.tile_list || !pm.tile_list2) return gemm_fail(LRN_ERR_NOMEM, "gemm: tile list allocation failed");
      pm.n1 = c1;
      pm.n2 = c2;
      if (c1 == 0) { pm.tile_list = pm.tile_list2; pm.n1 = c2; pm.n2 = 0; }
      const long wgs = 8L * d.ksplit * ((pm.n1 >> 3) + (pm.n2 >> 3));
      if (wgs > 0x7fffffffL) return gemm_fail(LRN_ERR_ARG, "gemm: grid too large");
      if (wgs > 0)
        hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<true, 4, 4>), dim3((unsigned)wgs, 1, 1), dim3(256),
                           2 * 2 * 128 * BK * 8, st, pm);
    } else {   // strip: all rows x the last 128 + r columns (entries below the diagonal of the corner are computed and never read)
      GemmParams ps = p;
      ps.d.flags &= ~(GEMM_TRI_LOWER | GEMM_TRI_UPPER | GEMM_DIAG_LOWER | GEMM_DIAG_UPPER);
      ps.m_org = 0; ps.n_org = Mm;
      ps.tilesM = tmain + 2; ps.tilesN = 1;
      int cnt = 0;
      ps.tile_list = get_tile_list(tmain + 2, 1, 0, &cnt);
      if (!ps.tile_list || cnt <= 0) return gemm_fail(LRN_ERR_NOMEM, "gemm: tile list allocation failed");
      hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<true, 4, 5>), dim3(cnt, 1, d.ksplit), dim3(256),
                         2 * (128 + 160) * BK * 8, st, ps);
    }
    return hipGetLastError() == hipSuccess ? LRN_OK : gemm_fail(LRN_ERR_HIP, "gemm: launch failed");
  }
  static const int tile_order = getenv("LRN_TILE_ORDER") ? atoi(getenv("LRN_TILE_ORDER")) : 0;      // (measurement knob)
  const int korder = (tile_order && d.tile_class == 0 && d.batch > 1) ? ((d.flags & GEMM_KFROM_N) ? 1 : (d.flags & GEMM_KFROM_M) ? 2 : 0) : 0;
  p.tile_list = get_tile_list(p.tilesM, p.tilesN, d.flags & (GEMM_TRI_LOWER | GEMM_TRI_UPPER), &ntile, d.tile_class,
                              d.tile_class != 0 && (d.M % BMv) != 0, d.tile_class != 0 && (d.N % BMv) != 0,
                              d.tile_class != 0 && (d.flags & (GEMM_DIAG_LOWER | GEMM_DIAG_UPPER)) != 0, korder);
  if (d.tile_class != 0 && p.tile_list && ntile == 0) return LRN_OK;      // nothing of that class
  if (!p.tile_list || ntile <= 0) return gemm_fail(LRN_ERR_NOMEM, "gemm: tile list allocation failed");
  (void)tri;
  const bool akc = (d.sAk == 1 && d.sAm != 1);
  const bool bkc = (d.sBk == 1 && d.sBn != 1);
  dim3 grid(ntile, 1, d.batch * d.ksplit);
  if (grid.z > 65535) return gemm_fail(LRN_ERR_ARG, "gemm: grid.z > 65535");
  const bool epi = (d.flags & (GEMM_OFFDIAG_X2 | GEMM_SQUARE | GEMM_C_PACKED | GEMM_C_MIRROR)) || d.C2;
  if (d.C2 && (d.M != d.N || d.batch != 1 || d.ksplit != 1 || kseg)) return gemm_fail(LRN_ERR_ARG, "gemm: C2 needs a square, unbatched, unsplit product");
  if (kflat) {
    if (big) {
      if (!big_tile_attr_ok()) return gemm_fail(LRN_ERR_HIP, "gemm: 80 KB of dynamic LDS refused");
      hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<true, 5, 5>), grid, dim3(256), 2 * 2 * 160 * BK * 8, st, p);
    } else {
      hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<true, 4, 4>), grid, dim3(256), 2 * 2 * 128 * BK * 8, st, p);
    }
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  if (kseg && kseg_lds_path_ok(d)) {
    hipLaunchKernelGGL((gemm_f64_kseg_lds_kernel<false, 4, 4>), grid, dim3(256), 2 * 2 * 128 * BK * 8, st, p);
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  if (!small && !kseg && lds_path_ok(d)) {
    const unsigned dyn = (d.flags & GEMM_LAB_ONE_WG) ? 24576u : 0u;        // (measurement only)
    if (dyn) {
      (void)hipFuncSetAttribute((const void*)gemm_f64_lds_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
      (void)hipFuncSetAttribute((const void*)gemm_f64_lds_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    }
    if (d.flags & GEMM_DYN_MASKS) {
      if (epi) hipLaunchKernelGGL((gemm_f64_lds_kernel<true, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((gemm_f64_lds_kernel<false, true>), grid, dim3(256), 0, st, p);
    } else {
      if (epi) hipLaunchKernelGGL((gemm_f64_lds_kernel<true, false>), grid, dim3(256), dyn, st, p);
      else hipLaunchKernelGGL((gemm_f64_lds_kernel<false, false>), grid, dim3(256), dyn, st, p);
    }
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  if (small && mid_mode != 0 && !kseg && !epi && d.batch == 1 && d.beta == 0.0 && d.sAm == 1 && d.sBn == 1 && d.K >= 64 &&
      d.sCn == 1 && d.sAk >= d.M && d.sBk >= d.N && !(d.flags & ~(GEMM_SMALL_TILE | GEMM_TRI_LOWER | GEMM_TRI_UPPER)) &&
      (!tri || (d.M == d.N && d.ksplit > 1)) &&
      (double)d.K * (double)std::max(d.sAk, d.sBk) * 8.0 < 2.0e9) {
    // a plain mid-size product, or its split-K slabs (gemm() below): three-stage LDS DMA pipeline
    static const char* trace_path = getenv("LRN_MID_TRACE");       // (measurement: clocks of the workgroups of launch #40)
    static int trace_launch = 0;
    unsigned long long* tb = nullptr;
    // (every tile of the grid, or its lower / upper triangle; the real entries come first in the list)
    p.n1 = (tri ? p.tilesM * (p.tilesM + 1) / 2 : p.tilesM * p.tilesN) * d.ksplit;
    grid = dim3((unsigned)((p.n1 + 7) & ~7), 1, 1);
    const size_t tw = 8 * (size_t)grid.x * grid.z;
    if (trace_path && ++trace_launch == 40 && hipMalloc(&tb, tw * 8) == hipSuccess) {
      (void)hipMemsetAsync(tb, 0, tw * 8, st);
      p.d.lab_trace = tb;
    }
    hipLaunchKernelGGL(gemm_f64_mid_kernel, grid, dim3(256), 0, st, p);
    if (tb) {
      std::vector<unsigned long long> h(tw);
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h.data(), tb, tw * 8, hipMemcpyDeviceToHost);
      if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, tw, f); fclose(f); }
      (void)hipFree(tb);
    }
    return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
  }
  if (small) {
    if (kseg) launch4<64, 64, true, false>(st, p, akc, bkc, grid);
    else if (epi) launch4<64, 64, false, true>(st, p, akc, bkc, grid);
    else launch4<64, 64, false, false>(st, p, akc, bkc, grid);
  } else {
    if (kseg) launch4<128, 128, true, false>(st, p, akc, bkc, grid);
    else if (epi) launch4<128, 128, false, true>(st, p, akc, bkc, grid);
    else launch4<128, 128, false, false>(st, p, akc, bkc, grid);
  }
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// ------------------------------------------------------------------ slab reduction
__global__ void reduce_slabs_kernel(const double* __restrict__ slabs, long stride, int nslab,
                                    double* __restrict__ out, long n, double beta) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long step = (long)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += slabs[(long)k * stride + i];
    out[i] = (beta != 0.0 ? beta * out[i] : 0.0) + s;
  }
}

int reduce_slabs(hipStream_t st, const double* slabs, long stride, int nslab, double* out,
                 long n, double beta) {
  if (n <= 0) return LRN_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slabs, stride,
                     nslab, out, n, beta);
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

// ------------------------------------------------------------------ MFMA probes
__global__ void mfma_probe_kernel(const double* A, const double* B, double* D) {
  int lane = threadIdx.x;
  double a = A[(lane & 15) * 4 + (lane >> 4)];   // A[row][k], row-major 16x4
  double b = B[(lane >> 4) * 16 + (lane & 15)];  // B[k][col], row-major 4x16
  v4f64 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[MFMA_F64_ROW(lane, r) * 16 + (lane & 15)] = c[r];
}

int mfma_f64_probe(hipStream_t st, const double* A, const double* B, double* D) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, st, A, B, D);
  return hipGetLastError() == hipSuccess ? LRN_OK : LRN_ERR_HIP;
}

__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  // inline asm keeps the accumulators in VGPRs (the builtin form made hipcc shuttle them
  // through AGPRs every iteration, which measured the copies, not the matrix pipe)
#define LRN_MFMA(c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
  for (int i = 0; i < iters; ++i) {
    LRN_MFMA(c0); LRN_MFMA(c1); LRN_MFMA(c2); LRN_MFMA(c3);
    LRN_MFMA(c4); LRN_MFMA(c5); LRN_MFMA(c6); LRN_MFMA(c7);
  }
#undef LRN_MFMA
  LRN_MFMA_DRAIN();            // (the asm MFMAs are invisible to the hazard recogniser: retire them before the VALU reads)
  v4f64 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[0] = s[0];
}

int mfma_f64_peak(hipStream_t st, double* tflops) {
  double* dummy = nullptr;
  if (hipMalloc(&dummy, 8) != hipSuccess) return LRN_ERR_NOMEM;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // (round 4: 4000 iterations -- 0.9 ms -- measured the clock's ramp, not the pipe: 71-72 TFLOP/s where a run of 40 ms
  // sustains 77.3-77.8 at 2.38-2.39 GHz, 64.0 shader cycles per MFMA of a SIMD; tools/lab/mfma_clock.hip)
  const int iters = 100000;
  const int blocks = 256 * 2;       // 2 workgroups of 4 waves per CU -> 2 waves per SIMD
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, dummy, 20000);
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, dummy, iters);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * 8 * 2048.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(dummy);
  return LRN_OK;
}

}  // namespace lrn
