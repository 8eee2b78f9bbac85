// Lab (round 4, second session): what does a chain of DEPENDENT launches cost per link on this machine, and would replaying
// it as a hipGraph be cheaper?  The Lanczos step of the step-length rule is such a chain: one launch of 51 workgroups per
// step, 7.3-8.0 us per step by the kernel stats for a few microseconds of work.
//   (a) plain launches on one stream, (b) the same 16 launches captured once and replayed with hipGraphLaunch,
// for an empty kernel and for a kernel that does what a Lanczos step does in shape (51 workgroups, every one reads three
// 800-vectors and reduces them, then streams its 16 columns of an 800 x 800 matrix).
// hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/launch_chain.hip -o /tmp/launch_chain && /tmp/launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void empty_kernel(double* p) {
  if (p == nullptr && threadIdx.x == 12345) p[0] = 1.0;
}

__global__ __launch_bounds__(256) void step_like_kernel(const double* __restrict__ M, int n, const double* __restrict__ vin,
                                                        double* __restrict__ vout) {
  __shared__ double qs[1024];
  __shared__ double sh[16];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < n; i += 256) { const double v = vin[i] - 0.5 * vin[n + i] - 0.25 * vin[2 * n + i]; qs[i] = v; s += v * v; }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((t & 63) == 0) sh[t >> 6] = s;
  __syncthreads();
  const double r = 1.0 / sqrt(sh[0] + sh[1] + sh[2] + sh[3] + 1.0);
  const int lane = t & 63, w = t >> 6;
  const int c0 = blockIdx.x * 16 + 4 * w;
  double a[4] = {0, 0, 0, 0};
  for (int k = lane; k < n; k += 64) {
    const double q = qs[k] * r;
    for (int u = 0; u < 4; ++u) a[u] += M[(size_t)min(c0 + u, n - 1) * n + k] * q;
  }
  for (int u = 0; u < 4; ++u) {
    for (int off = 32; off > 0; off >>= 1) a[u] += __shfl_down(a[u], off, 64);
    if (lane == 0 && c0 + u < n) vout[c0 + u] = a[u];
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const int n = 801, nwg = (n + 15) / 16, LINKS = 16, REPS = 200;
  double *M, *v;
  hipMalloc(&M, (size_t)n * n * 8);
  hipMalloc(&v, (size_t)8 * n * 8);
  hipMemset(M, 0, (size_t)n * n * 8);
  hipMemset(v, 0, (size_t)8 * n * 8);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int kind = 0; kind < 2; ++kind) {
    auto launch = [&](int j) {
      if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(256), 0, st, v);
      else hipLaunchKernelGGL(step_like_kernel, dim3(nwg), dim3(256), 0, st, M, n, v + (size_t)(j % 2) * 3 * n, v + (size_t)((j + 1) % 2) * 3 * n);
    };
    // (a) plain launches
    for (int j = 0; j < 64; ++j) launch(j);
    hipStreamSynchronize(st);
    double t0 = now_us();
    for (int rep = 0; rep < REPS; ++rep)
      for (int j = 0; j < LINKS; ++j) launch(j);
    double t_issue = now_us() - t0;
    hipStreamSynchronize(st);
    double t_plain = now_us() - t0;
    // (b) graph of LINKS launches
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int j = 0; j < LINKS; ++j) launch(j);
    hipStreamEndCapture(st, &g);
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { printf("hipGraphInstantiate failed: %s\n", hipGetErrorString(e)); return 1; }
    for (int rep = 0; rep < 4; ++rep) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = now_us();
    for (int rep = 0; rep < REPS; ++rep) hipGraphLaunch(ge, st);
    double tg_issue = now_us() - t0;
    hipStreamSynchronize(st);
    double t_graph = now_us() - t0;
    // (c) one batch at a time with a host round trip after it (what the Lanczos driver does): plain vs graph
    t0 = now_us();
    for (int rep = 0; rep < REPS; ++rep) { for (int j = 0; j < LINKS; ++j) launch(j); hipStreamSynchronize(st); }
    double t_plain_sync = now_us() - t0;
    t0 = now_us();
    for (int rep = 0; rep < REPS; ++rep) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    double t_graph_sync = now_us() - t0;
    printf("%s, %d workgroups: plain %.2f us per launch (host issue %.2f) | graph of %d: %.2f us per launch (host issue %.2f per node) | "
           "batch + sync: plain %.2f, graph %.2f us per launch\n",
           kind == 0 ? "empty kernel" : "step-like kernel", nwg, t_plain / (REPS * LINKS), t_issue / (REPS * LINKS), LINKS,
           t_graph / (REPS * LINKS), tg_issue / (REPS * LINKS), t_plain_sync / (REPS * LINKS), t_graph_sync / (REPS * LINKS));
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  hipError_t e = hipGetLastError();
  printf("hip error %d\n", (int)e);
  return 0;
}
