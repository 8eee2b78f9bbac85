// Lab (round 4): the K loop of GEMM3' (both operands K-contiguous, [m][k] LDS images filled by global_load_lds_dwordx4, XOR
// swizzle) -- fragment-read variants.  C = X X' for X[4096][K] row-major, 128 x 128 tiles (32 x 32 = 1024 workgroups = two
// rounds of the 512 slots), every tile the whole K range.
//   V = 0: production (round 3): 8-byte fragment reads of k = 4 kk + fk, pair index ^ ((row >> 1) & 7)   [2-way bank conflicts]
//   V = 1: 16-byte reads of the pairs fk, fk + 4, pair index ^ (row & 7), the compiler's schedule
//   V = 2: the same, both pair groups requested first (sched_barrier), then the 64 MFMAs
//   V = 3: group 1 requested after the first 16 MFMAs of group 0
//   V = 4: 8-byte reads as V = 0 but k = 4 kk + fk swizzled with row & 7 on a [m][k] image whose pairs hold (k, k + 8): see below
// hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/g3_bench.hip -o /tmp/g3_bench && /tmp/g3_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int BK = 16;
struct P { const double* X; double* C; int M, K; long ld; };
template <int V>
__global__ __launch_bounds__(256, 2) void g3(P p) {
  constexpr int LA = 128 * BK;
  __shared__ double lds[2 * 2 * LA];
  const int tm = blockIdx.x, tn = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int m0 = tm * 128, n0 = tn * 128;
  const int lrow = lane >> 3, lpair = lane & 7;
  const double* pa[4];
  const double* pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * (w + 4 * j) + lrow;
    const int sp = V == 0 ? (lpair ^ ((row >> 1) & 7)) : (lpair ^ (row & 7));
    pa[j] = p.X + (long)(m0 + row) * p.ld + 2 * sp;
    pb[j] = p.X + (long)(n0 + row) * p.ld + 2 * sp;
  }
  auto issue = [&](long kb, int buf) {
    double* sa = lds + buf * (2 * LA);
    double* sb = sa + LA;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r8 = 8 * (w + 4 * j);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[j] + kb), (__attribute__((address_space(3))) void*)(sa + r8 * BK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[j] + kb), (__attribute__((address_space(3))) void*)(sb + r8 * BK), 16, 0, 0);
    }
  };
  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const int fr = lane & 15, fk = lane >> 4;
  const int nk = p.K / BK;
  issue(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue((long)(kt + 1) * BK, cur ^ 1);
    const double* sa = lds + cur * (2 * LA);
    const double* sb = sa + LA;
    if constexpr (V == 4 || V == 5) {
      // the production reads (8 bytes, 2-way conflicts), but requested ahead of the MFMAs: V = 4 all four quarters first,
      // V = 5 two quarters ahead (fragments of quarter kk + 2 requested before the MFMAs of quarter kk)
      double fa[4][4], fb[4][4];
      auto rd = [&](int kk) __attribute__((always_inline)) {
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = (2 * i + wm) * 16 + fr; fa[kk][i] = sa[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int row = (2 * j + wn) * 16 + fr; fb[kk][j] = sb[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)]; }
      };
      auto mm = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
      };
      if constexpr (V == 4) {
        rd(0); __builtin_amdgcn_sched_barrier(0); rd(1); __builtin_amdgcn_sched_barrier(0); rd(2); __builtin_amdgcn_sched_barrier(0); rd(3);
        __builtin_amdgcn_sched_barrier(0); mm(0); mm(1); mm(2); mm(3);
      } else {
        rd(0); __builtin_amdgcn_sched_barrier(0); rd(1); __builtin_amdgcn_sched_barrier(0); mm(0); __builtin_amdgcn_sched_barrier(0);
        rd(2); __builtin_amdgcn_sched_barrier(0); mm(1); __builtin_amdgcn_sched_barrier(0); rd(3); __builtin_amdgcn_sched_barrier(0); mm(2); mm(3);
      }
    } else if constexpr (V == 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double fa[4], fb[4];
        const int k = kk * 4 + fk;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = (2 * i + wm) * 16 + fr; fa[i] = sa[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int row = (2 * j + wn) * 16 + fr; fb[j] = sb[row * BK + 2 * ((k >> 1) ^ ((row >> 1) & 7)) + (k & 1)]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
      double2 fa[2][4], fb[2][4];
      auto rd = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = (2 * i + wm) * 16 + fr; fa[h][i] = *reinterpret_cast<const double2*>(sa + row * BK + 2 * ((fk + 4 * h) ^ (row & 7))); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int row = (2 * j + wn) * 16 + fr; fb[h][j] = *reinterpret_cast<const double2*>(sb + row * BK + 2 * ((fk + 4 * h) ^ (row & 7))); }
      };
      auto mm = [&](int h, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(half ? fa[h][i].y : fa[h][i].x, half ? fb[h][j].y : fb[h][j].x, acc[i][j], 0, 0, 0);
      };
      if constexpr (V == 1) { rd(0); mm(0, 0); mm(0, 1); rd(1); mm(1, 0); mm(1, 1); }
      if constexpr (V == 2) { rd(0); __builtin_amdgcn_sched_barrier(0); rd(1); __builtin_amdgcn_sched_barrier(0); mm(0, 0); mm(0, 1); mm(1, 0); mm(1, 1); }
      if constexpr (V == 3) { rd(0); __builtin_amdgcn_sched_barrier(0); mm(0, 0); __builtin_amdgcn_sched_barrier(0); rd(1); __builtin_amdgcn_sched_barrier(0); mm(0, 1); mm(1, 0); mm(1, 1); }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (2 * i + wm) * 16 + (lane >> 4) + 4 * r, n = n0 + (2 * j + wn) * 16 + fr;
        p.C[(long)m * p.M + n] = acc[i][j][r];
      }
}
__global__ void fill(double* x, long n, unsigned long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned long z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}
template <int V> float run(P p) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((g3<V>), dim3(p.M / 128, p.M / 128), dim3(256), 0, 0, p);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
  }
  return best;
}
int main(int argc, char** argv) {
  const int M = 4096, K = argc > 1 ? atoi(argv[1]) : 8192;
  double *X, *C, *C0;
  hipMalloc(&X, (long)M * K * 8); hipMalloc(&C, (long)M * M * 8); hipMalloc(&C0, (long)M * M * 8);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, X, (long)M * K, 5ul);
  P p{X, C0, M, K, K};
  const double steps = 1024.0 * (K / 16);
  float t0 = run<0>(p);
  printf("V0 production (8-byte reads, 2-way conflicts): %.2f ms = %.3f ns per tile-step = %.1f TFLOP/s\n", t0, t0 * 1e6 / steps, 524288.0 / (t0 * 1e6 / steps) / 1e3);
  std::vector<double> h0((long)M * M), h((long)M * M);
  hipMemcpy(h0.data(), C0, (long)M * M * 8, hipMemcpyDeviceToHost);
  p.C = C;
  float tv[3] = {run<1>(p), 0, 0};
  hipMemcpy(h.data(), C, (long)M * M * 8, hipMemcpyDeviceToHost);
  double d = 0, s = 0; for (long i = 0; i < (long)M * M; ++i) { d += (h[i] - h0[i]) * (h[i] - h0[i]); s += h0[i] * h0[i]; }
  tv[1] = run<2>(p); tv[2] = run<3>(p);
  {
    p.C = C;
    float t4 = run<4>(p);
    hipMemcpy(h.data(), C, (long)M * M * 8, hipMemcpyDeviceToHost);
    long bad = 0; for (long i = 0; i < (long)M * M; ++i) bad += h[i] != h0[i];
    float t5 = run<5>(p);
    printf("V4 production reads, all four quarters requested first: %.2f ms = %.3f ns per tile-step = %.1f TFLOP/s (%ld entries differ from V0)\n", t4, t4 * 1e6 / steps, 524288.0 / (t4 * 1e6 / steps) / 1e3, bad);
    printf("V5 production reads, two quarters ahead: %.2f ms = %.3f ns per tile-step = %.1f TFLOP/s\n", t5, t5 * 1e6 / steps, 524288.0 / (t5 * 1e6 / steps) / 1e3);
  }
  const char* nm[3] = {"V1 16-byte reads, compiler's schedule", "V2 both groups first", "V3 group 1 after 16 MFMAs"};
  for (int v = 0; v < 3; ++v) printf("%s: %.2f ms = %.3f ns per tile-step = %.1f TFLOP/s\n", nm[v], tv[v], tv[v] * 1e6 / steps, 524288.0 / (tv[v] * 1e6 / steps) / 1e3);
  printf("V1 against V0: %.2e relative (another grouping of the k of a step)\n", sqrt(d / s));
  return 0;
}
