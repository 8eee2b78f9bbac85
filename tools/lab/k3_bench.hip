// Lab (round 4): does a third workgroup per CU pay for GEMM1'-like work?  A batched lower-tile product P_k = A_k L with K from
// the tile's column origin (every block computed: the gemm_no_skip form), 128 x 128 tiles, direct-to-LDS staging:
//   k3<16, 2>: K-steps of 16, 73.7 KB of LDS, 184 registers -> two workgroups per CU (the production kernel's shape)
//   k3<8, 3>:  K-steps of 8, 36.9 KB of LDS, 154 registers -> three workgroups per CU
// Build and run on the GPU box: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/k3_bench.hip -o /tmp/k3_bench && /tmp/k3_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
struct P { const double* A; const double* B; double* C; int M, N, K, lda, ldb, ldc; long bA, bB, bC; int kfrom; const int2* tiles; int mode; unsigned long long* clk; };
template <int BKL, int WPE>
__global__ __launch_bounds__(256, WPE) void k3(P p) {
  constexpr int LROW = 144, LA = BKL * LROW;
  __shared__ double lds[2 * 2 * LA];
  const bool rec = p.clk && blockIdx.x == 0 && blockIdx.z == gridDim.z / 2 && threadIdx.x == 0;      // a workgroup in the middle of the run
  unsigned long long s0 = 0, w0 = 0;
  if (rec) { s0 = clock64(); w0 = wall_clock64(); }
  const int2 tt = p.tiles[blockIdx.x];
  const int tm = tt.x, tn = tt.y, bz = blockIdx.z;
  const double* Ag = p.A + (long)bz * p.bA;
  const double* Bg = p.B + (long)bz * p.bB;
  double* Cg = p.C + (long)bz * p.bC;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int m0 = tm * 128, n0 = tn * 128;
  const unsigned bytesA = (unsigned)(((long)(p.K - 1) * p.lda + p.M) * 8);
  const unsigned bytesB = (unsigned)(((long)(p.K - 1) * p.ldb + p.N) * 8);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bg, 0, bytesB, 0x00020000);
  const int ma = m0 + 2 * lane, nb = n0 + 2 * lane;
  const unsigned offA = ma < p.M ? (unsigned)ma * 8u : 0x80000000u;
  const unsigned offB = nb < p.N ? (unsigned)nb * 8u : 0x80000000u;
  const int nk = (p.K + BKL - 1) / BKL;
  const int kt0 = p.kfrom ? n0 / BKL : 0;
  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  auto issue = [&](int kt, int buf) {
    double* sa = lds + buf * (2 * LA);
    double* sb = sa + LA;
#pragma unroll
    for (int j = 0; j < BKL / 4; ++j) {
      const int kr = w + 4 * j;
      const unsigned krow = (unsigned)(kt * BKL + kr);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + kr * LROW), 16, (int)(offA + krow * (unsigned)p.lda * 8u), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + kr * LROW), 16, (int)(offB + krow * (unsigned)p.ldb * 8u), 0, 0, 0);
    }
  };
  const int fr = lane & 15, fk = lane >> 4;
  issue(kt0, kt0 & 1);
  __syncthreads();
  if (p.mode == 0) {
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
      const double* sa = lds + cur * (2 * LA);
      const double* sb = sa + LA;
#pragma unroll
      for (int kk = 0; kk < BKL / 4; ++kk) {
        double fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = sa[(kk * 4 + fk) * LROW + (2 * i + wm) * 16 + fr];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = sb[(kk * 4 + fk) * LROW + (2 * j + wn) * 16 + fr];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  } else {
    // dissection (wrong results on purpose): mode bit 0 = no DMA inside the loop, bit 1 = the fragments are read once,
    // bit 2 = no barrier
    double fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = lds[fk * LROW + (2 * i + wm) * 16 + fr];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = lds[LA + fk * LROW + (2 * j + wn) * 16 + fr];
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (!(p.mode & 1) && kt + 1 < nk) issue(kt + 1, cur ^ 1);
      const double* sa = lds + cur * (2 * LA);
      const double* sb = sa + LA;
#pragma unroll
      for (int kk = 0; kk < BKL / 4; ++kk) {
        if (!(p.mode & 2)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = sa[(kk * 4 + fk) * LROW + (2 * i + wm) * 16 + fr];
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = sb[(kk * 4 + fk) * LROW + (2 * j + wn) * 16 + fr];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      if (!(p.mode & 4)) __syncthreads();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (rec) { p.clk[0] = clock64() - s0; p.clk[1] = wall_clock64() - w0; }
  double* cbase = Cg + (long)(m0 + wm * 16 + (lane >> 4)) * p.ldc + (n0 + wn * 16 + fr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (2 * i + wm) * 16 + (lane >> 4) + 4 * r, n = n0 + (2 * j + wn) * 16 + fr;
        if (m < p.M && n < p.N) cbase[(long)(32 * i + 4 * r) * p.ldc + 32 * j] = acc[i][j][r];
      }
}
// k5: k3<16, 2> with the barrier of a K-step moved in front of its LAST quarter: the fragments of the last quarter are in
// registers before the barrier, after it the DMA of the K-step after next goes into the buffer nobody reads any more and the
// first fragments of the next K-step are requested -- the barrier's skew and the first LDS latency pass under 16 MFMAs
__global__ __launch_bounds__(256, 2) void k5(P p) {
  constexpr int BKL = 16, LROW = 144, LA = BKL * LROW;
  __shared__ double lds[2 * 2 * LA];
  const int2 tt = p.tiles[blockIdx.x];
  const int tm = tt.x, tn = tt.y, bz = blockIdx.z;
  const double* Ag = p.A + (long)bz * p.bA;
  const double* Bg = p.B + (long)bz * p.bB;
  double* Cg = p.C + (long)bz * p.bC;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int m0 = tm * 128, n0 = tn * 128;
  const unsigned bytesA = (unsigned)(((long)(p.K - 1) * p.lda + p.M) * 8);
  const unsigned bytesB = (unsigned)(((long)(p.K - 1) * p.ldb + p.N) * 8);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bg, 0, bytesB, 0x00020000);
  const int ma = m0 + 2 * lane, nb = n0 + 2 * lane;
  const unsigned offA = ma < p.M ? (unsigned)ma * 8u : 0x80000000u;
  const unsigned offB = nb < p.N ? (unsigned)nb * 8u : 0x80000000u;
  const int nk = (p.K + BKL - 1) / BKL;
  const int kt0 = p.kfrom ? n0 / BKL : 0;
  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  auto issue = [&](int kt, int buf) {
    double* sa = lds + buf * (2 * LA);
    double* sb = sa + LA;
#pragma unroll
    for (int j = 0; j < BKL / 4; ++j) {
      const int kr = w + 4 * j;
      const unsigned krow = (unsigned)(kt * BKL + kr);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + kr * LROW), 16, (int)(offA + krow * (unsigned)p.lda * 8u), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + kr * LROW), 16, (int)(offB + krow * (unsigned)p.ldb * 8u), 0, 0, 0);
    }
  };
  const int fr = lane & 15, fk = lane >> 4;
  const int aoff = fk * LROW + wm * 16 + fr, boff = LA + fk * LROW + wn * 16 + fr;     // + kk * 4 * LROW + 32 i
  double fa[2][4], fb[2][4];
  auto frags = [&](const double* sbuf, int kk, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[set][i] = sbuf[aoff + kk * 4 * LROW + 32 * i];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[set][j] = sbuf[boff + kk * 4 * LROW + 32 * j];
  };
  auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
  };
  issue(kt0, kt0 & 1);
  __syncthreads();
  if (kt0 + 1 < nk) issue(kt0 + 1, (kt0 + 1) & 1);
  frags(lds + (kt0 & 1) * (2 * LA), 0, 0);
  for (int kt = kt0; kt < nk; ++kt) {
    const double* sc = lds + (kt & 1) * (2 * LA);
    frags(sc, 1, 1); mfmas(0);
    frags(sc, 2, 0); mfmas(1);
    frags(sc, 3, 1); mfmas(0);
    if (kt + 1 < nk) {
      __syncthreads();                                   // (vmcnt(0): K-step kt + 1 has landed; every wave holds its last fragments of kt)
      if (kt + 2 < nk) issue(kt + 2, kt & 1);
      frags(lds + ((kt + 1) & 1) * (2 * LA), 0, 0);
    }
    mfmas(1);
  }
  double* cbase = Cg + (long)(m0 + wm * 16 + (lane >> 4)) * p.ldc + (n0 + wn * 16 + fr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (2 * i + wm) * 16 + (lane >> 4) + 4 * r, n = n0 + (2 * j + wn) * 16 + fr;
        if (m < p.M && n < p.N) cbase[(long)(32 * i + 4 * r) * p.ldc + 32 * j] = acc[i][j][r];
      }
}

// k4: 256 x 128 tile, 512 threads (8 waves, 4 x 2, 64 x 64 each), K-steps of 16, 106 KB of LDS: ONE workgroup per CU, the same two
// waves per SIMD -- 0.75 of the panel bytes per flop, but all eight waves meet at one barrier per K-step
__global__ __launch_bounds__(512, 2) void k4(P p) {
  constexpr int BKL = 16, LRA = 272, LRB = 144, LA = BKL * LRA, LB = BKL * LRB;
  extern __shared__ double lds[];               // 2 * (LA + LB) doubles
  const int2 tt = p.tiles[blockIdx.x];
  const int tm = tt.x, tn = tt.y, bz = blockIdx.z;      // rows [256 tm, +256), columns [128 tn, +128)
  const double* Ag = p.A + (long)bz * p.bA;
  const double* Bg = p.B + (long)bz * p.bB;
  double* Cg = p.C + (long)bz * p.bC;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 3, wn = w >> 2;
  const int m0 = tm * 256, n0 = tn * 128;
  const unsigned bytesA = (unsigned)(((long)(p.K - 1) * p.lda + p.M) * 8);
  const unsigned bytesB = (unsigned)(((long)(p.K - 1) * p.ldb + p.N) * 8);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bg, 0, bytesB, 0x00020000);
  unsigned offA[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { const int ma = m0 + 128 * h + 2 * lane; offA[h] = ma < p.M ? (unsigned)ma * 8u : 0x80000000u; }
  const int nb = n0 + 2 * lane;
  const unsigned offB = nb < p.N ? (unsigned)nb * 8u : 0x80000000u;
  const int nk = (p.K + BKL - 1) / BKL;
  const int kt0 = p.kfrom ? n0 / BKL : 0;
  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  auto issue = [&](int kt, int buf) {
    double* sa = lds + buf * (LA + LB);
    double* sb = sa + LA;
#pragma unroll
    for (int j = 0; j < BKL / 8; ++j) {
      const int kr = w + 8 * j;                        // eight waves: two k-rows each
      const unsigned krow = (unsigned)(kt * BKL + kr);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sa + kr * LRA + 128 * h), 16, (int)(offA[h] + krow * (unsigned)p.lda * 8u), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(sb + kr * LRB), 16, (int)(offB + krow * (unsigned)p.ldb * 8u), 0, 0, 0);
    }
  };
  const int fr = lane & 15, fk = lane >> 4;
  issue(kt0, kt0 & 1);
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const double* sa = lds + cur * (LA + LB);
    const double* sb = sa + LA;
#pragma unroll
    for (int kk = 0; kk < BKL / 4; ++kk) {
      double fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = sa[(kk * 4 + fk) * LRA + wm * 64 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = sb[(kk * 4 + fk) * LRB + wn * 64 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  double* cbase = Cg + (long)(m0 + wm * 64 + (lane >> 4)) * p.ldc + (n0 + wn * 64 + fr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r, n = n0 + wn * 64 + j * 16 + fr;
        if (m < p.M && n < p.N) cbase[(long)(16 * i + 4 * r) * p.ldc + 16 * j] = acc[i][j][r];
      }
}

__global__ void fill(double* x, long n, unsigned long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned long z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048, nb = argc > 2 ? atoi(argv[2]) : 300;
  const long mm = (long)n * n;
  double *A, *L, *C8, *C16;
  hipMalloc(&A, mm * nb * 8); hipMalloc(&L, mm * 8); hipMalloc(&C8, mm * nb * 8); hipMalloc(&C16, mm * nb * 8);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, mm * nb, 1ul);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, L, mm, 77ul);
  hipMemset(C8, 0, mm * nb * 8); hipMemset(C16, 0, mm * nb * 8);
  const int nt = (n + 127) / 128;
  std::vector<int2> tl;                       // 8 x 8 super-tiles, tm fastest, lower tiles (the production order)
  for (int sn = 0; sn < nt; sn += 8) for (int sm = 0; sm < nt; sm += 8)
    for (int tn = sn; tn < sn + 8 && tn < nt; ++tn) for (int tm = sm; tm < sm + 8 && tm < nt; ++tm) if (tm >= tn) tl.push_back(make_int2(tm, tn));
  int2* dtl; hipMalloc(&dtl, tl.size() * sizeof(int2)); hipMemcpy(dtl, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice);
  for (int kfrom = 1; kfrom >= 0; --kfrom) {
    P p{A, L, nullptr, n, n, n, n, n, n, mm, 0, mm, kfrom, dtl, 0, nullptr};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((unsigned)tl.size(), 1, nb);
    for (int variant = 0; variant < 2; ++variant) {
      p.C = variant ? C8 : C16;
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        if (variant) hipLaunchKernelGGL((k3<8, 3>), grid, dim3(256), 0, 0, p);
        else hipLaunchKernelGGL((k3<16, 2>), grid, dim3(256), 0, 0, p);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
      }
      double ksteps = 0; for (auto& t : tl) ksteps += (kfrom ? n - 128 * t.y : n) / 16.0;
      printf("n %d batch %d kfrom %d %s: %.2f ms = %.3f ns per 128x128x16 tile-step, hip error %d\n", n, nb, kfrom,
             variant ? "BK 8, three workgroups per CU" : "BK 16, two workgroups per CU ", best, best * 1e6 / (ksteps * nb), (int)hipGetLastError());
    }
    {
      p.C = C8;
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k5, grid, dim3(256), 0, 0, p);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
      }
      double ksteps = 0; for (auto& t : tl) ksteps += (kfrom ? n - 128 * t.y : n) / 16.0;
      printf("n %d batch %d kfrom %d k5 (barrier before the last quarter)   : %.2f ms = %.3f ns per 128x128x16 tile-step, hip error %d\n", n, nb, kfrom, best,
             best * 1e6 / (ksteps * nb), (int)hipGetLastError());
      if (kfrom == 0) {
        std::vector<double> h5(mm), h16(mm);
        hipMemcpy(h5.data(), C8 + mm * (nb - 1), mm * 8, hipMemcpyDeviceToHost); hipMemcpy(h16.data(), C16 + mm * (nb - 1), mm * 8, hipMemcpyDeviceToHost);
        long bad = 0; for (long i = 0; i < mm; ++i) bad += h5[i] != h16[i];
        printf("k5 against k3<16, 2>, last matrix: %ld entries differ\n", bad);
      }
    }
  }
  {   // dissection of the K loop of k3<16, 2>, whole K range (equal tiles)
    unsigned long long* dclk; hipMalloc(&dclk, 16);
    P p{A, L, C8, n, n, n, n, n, n, mm, 0, mm, 0, dtl, 0, dclk};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[8] = {"production loop", "no DMA in the loop", "fragments read once", "no DMA, fragments once", "no barrier", "no DMA, no barrier",
                            "fragments once, no barrier", "MFMAs only"};
    for (int mode = 0; mode < 8; ++mode) {
      p.mode = mode;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k3<16, 2>), dim3((unsigned)tl.size(), 1, nb), dim3(256), 0, 0, p);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
      }
      unsigned long long hc[2]; hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost);
      printf("dissection %d (%s): %.2f ms = %.3f ns per tile-step = %.1f TFLOP/s; shader clock of a workgroup in mid-run %.0f MHz, its K loop %.1f us\n", mode, names[mode], best,
             best * 1e6 / (tl.size() * (n / 16.0) * nb), 524288.0 / (best * 1e6 / (tl.size() * (n / 16.0) * nb)) / 1e3, hc[0] / (hc[1] / 100.0), hc[1] / 100.0);
    }
  }
  {   // k4: 256 x 128 tiles (tm2, tn), any part on or below the diagonal: tn <= 2 tm2 + 1
    std::vector<int2> t2;
    const int nt2 = (n + 255) / 256;
    for (int sn = 0; sn < nt; sn += 8) for (int sm = 0; sm < nt2; sm += 4)
      for (int tn = sn; tn < sn + 8 && tn < nt; ++tn) for (int tm = sm; tm < sm + 4 && tm < nt2; ++tm) if (tn <= 2 * tm + 1) t2.push_back(make_int2(tm, tn));
    int2* d2; hipMalloc(&d2, t2.size() * sizeof(int2)); hipMemcpy(d2, t2.data(), t2.size() * sizeof(int2), hipMemcpyHostToDevice);
    const size_t ldsb = 2 * (16 * 272 + 16 * 144) * 8;
    hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kfrom = 1; kfrom >= 0; --kfrom) {
      P p{A, L, C8, n, n, n, n, n, n, mm, 0, mm, kfrom, d2, 0, nullptr};
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k4, dim3((unsigned)t2.size(), 1, nb), dim3(512), ldsb, 0, p);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
      }
      double ksteps = 0; for (auto& t : t2) ksteps += 2.0 * (kfrom ? n - 128 * t.y : n) / 16.0;
      printf("n %d batch %d kfrom %d 256 x 128 tile, one workgroup of eight waves per CU: %.2f ms = %.3f ns per executed 128x128x16 tile-step (%zu tiles), hip error %d\n",
             n, nb, kfrom, best, best * 1e6 / (ksteps * nb), t2.size(), (int)hipGetLastError());
    }
  }
  // the two variants sum in the same order: compare
  std::vector<double> h8(mm), h16(mm);
  hipMemcpy(h8.data(), C8 + mm * (nb - 1), mm * 8, hipMemcpyDeviceToHost); hipMemcpy(h16.data(), C16 + mm * (nb - 1), mm * 8, hipMemcpyDeviceToHost);
  double d = 0, s = 0; for (long i = 0; i < mm; ++i) { d += (h8[i] - h16[i]) * (h8[i] - h16[i]); s += h16[i] * h16[i]; }
  printf("difference of the two variants (last matrix): %.3e relative\n", s > 0 ? sqrt(d / s) : -1.0);
  return 0;
}
