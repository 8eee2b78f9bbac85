// Lab (round 4, second session): what does a barrier between the workgroups of a RESIDENT kernel cost when all of them sit
// on ONE XCD (one L2) -- against the same number of workgroups spread over the eight XCDs?
// Background: a Lanczos step of the step-length rule is one launch of 51 workgroups, 7-8 us
// per step; the resident multi-step kernel of round 4 (a device-scope release / acquire barrier per step across eight L2s)
// was slower than one launch per step.  Workgroups are dealt to the XCDs round-robin by their index, so the workgroups
// with blockIdx % 8 == 0 of a 256-workgroup launch share XCD 0: 32 CUs, 16 MB of registers -- enough to keep an 800 x 800
// matrix (5 MB) in registers for the whole run.
//   variant A: __threadfence() + atomic add + acquire spin (what a generic grid barrier does)
//   variant B: relaxed agent-scope atomics only; the payload (a 800-vector per step) travels through relaxed atomic stores and
//              loads as well, so no cache needs writing back or invalidating
// Every spin is bounded (the kernel cannot hang: after 2^22 polls a workgroup raises the abort word and all leave).
// hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <bool FENCES>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* flag, double* payload, unsigned long long* out, int nbar, int local,
                                                      int P, int n) {
  const int b = blockIdx.x;
  int me;
  if (local) { if (b % 8 != 0 || b / 8 >= P) return; me = b / 8; }
  else { if (b >= P) return; me = b; }
  const int t = threadIdx.x;
  if (t == 0) out[8 + me] = xcc_id();
  __shared__ int ok;
  if (t == 0) ok = 1;
  __syncthreads();
  double acc = 0.0;
  const unsigned long long w0 = wall_clock64();
  for (int s = 1; s <= nbar; ++s) {
    // payload of the step: this workgroup's share of an n-vector
    const int per = (n + P - 1) / P;
    for (int i = t; i < per; i += 256) {
      const int g = me * per + i;
      if (g < n) {
        const double v = (double)s + acc * 1e-30;
        if (FENCES) payload[(size_t)(s & 1) * n + g] = v;
        else __hip_atomic_store(payload + (size_t)(s & 1) * n + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (t == 0) {
      if (FENCES) __threadfence();
      __hip_atomic_fetch_add(flag, 1u, FENCES ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)s * (unsigned)P;
      int polls = 0;
      while (__hip_atomic_load(flag, FENCES ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++polls > (1 << 22) || __hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store(flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
      if (FENCES) __threadfence();
    }
    __syncthreads();
    if (!ok) break;
    // every workgroup reads the whole vector of the step
    for (int i = t; i < n; i += 256) {
      double v;
      if (FENCES) v = payload[(size_t)(s & 1) * n + i];
      else v = __hip_atomic_load(payload + (size_t)(s & 1) * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += v;
    }
  }
  const unsigned long long w1 = wall_clock64();
  if (t == 0 && me == 0) { out[0] = w1 - w0; out[1] = (unsigned long long)ok; }
  if (acc == 12345.678 && t == 0) out[2] = 1;        // (keeps the reads alive)
}

int main() {
  const int n = 801, nbar = 2000;
  unsigned* flag;
  double* payload;
  unsigned long long* out;
  hipMalloc(&flag, 64);
  hipMalloc(&payload, (size_t)2 * n * 8);
  hipMalloc(&out, 8 * 128);
  std::vector<unsigned long long> h(128);
  for (int P : {32, 51})
    for (int local = 1; local >= 0; --local) {
      if (local && P > 32) continue;
      for (int fences = 1; fences >= 0; --fences) {
        hipMemset(flag, 0, 64);
        hipMemset(payload, 0, (size_t)2 * n * 8);
        hipMemset(out, 0, 8 * 128);
        if (fences) hipLaunchKernelGGL(barrier_kernel<true>, dim3(256), dim3(256), 0, 0, flag, payload, out, nbar, local, P, n);
        else hipLaunchKernelGGL(barrier_kernel<false>, dim3(256), dim3(256), 0, 0, flag, payload, out, nbar, local, P, n);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h.data(), out, 8 * 128, hipMemcpyDeviceToHost);
        int xcds[16] = {0};
        for (int i = 0; i < P; ++i) xcds[h[8 + i] & 15]++;
        printf("%2d workgroups %s, %s: %.2f us per step (barrier + %d-vector exchanged), completed %d, hip %d | workgroups per XCD:", P,
               local ? "on ONE XCD (blockIdx % 8 == 0)" : "spread over the XCDs       ",
               fences ? "fences + acquire/release" : "relaxed atomics only    ", (double)h[0] / 100.0 / nbar, n, (int)h[1], (int)e);
        for (int x = 0; x < 8; ++x) printf(" %d", xcds[x]);
        printf("\n");
      }
    }
  return 0;
}
