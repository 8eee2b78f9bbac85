// Lab (round 4): what limits the FP64 MFMA probe at 71-72 of the nominal 78.6 TFLOP/s -- the clock the chip sustains under the
// load, or cycles between two v_mfma_f64_16x16x4_f64 of a SIMD beyond the 64 of its 16 passes?  Every workgroup runs the
// register-only loop of lrn_mfma_f64_peak; one lane reads the shader clock (s_memtime) and the 100 MHz wall clock before and after.
// hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
  v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const unsigned long long s0 = clock64(), w0 = wall_clock64();
#define M(c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
  for (int i = 0; i < iters; ++i) { M(c0); M(c1); M(c2); M(c3); M(c4); M(c5); M(c6); M(c7); }
#undef M
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long s1 = clock64(), w1 = wall_clock64();
  v4f64 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[7] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s1 - s0; out[1] = w1 - w0; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu)
    for (int iters : {20000, 200000}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(probe, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, 1000);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      const double us = h[1] / 100.0, mhz = h[0] / us;
      const double mfma_per_simd = (double)iters * 8 * wgs_per_cu;            // one wave of every workgroup per SIMD
      printf("%d workgroup(s) per CU, %d iterations: %.2f ms, %.1f TFLOP/s; shader clock %.0f MHz; %.1f shader cycles per MFMA of a SIMD\n", wgs_per_cu,
             iters, ms, 256.0 * wgs_per_cu * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12, mhz, h[0] / mfma_per_simd);
    }
  return 0;
}
