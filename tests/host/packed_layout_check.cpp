// Host-side check of the packed lower layout (loraine.jl_amd/csrc/lrn_common.h) used by the Cholesky path of the
// Schur assembly: offsets are a bijection onto [0, Kp), the diagonal 16-blocks fill [0, Kd), column pieces are
// contiguous multiples of 16.  Built by
// tests/test_packed_layout_cpu.py with hipcc (host code only, no GPU needed).
#include <cstdio>
#include <vector>

#include "lrn_common.h"
using namespace lrn;

int main() {
  int bad = 0;
  const int sizes[] = {1, 2, 5, 15, 16, 17, 31, 32, 33, 100, 127, 128, 129, 150, 255, 256, 257, 300, 333, 801, 1000, 2000};
  for (int m : sizes) {
    const int S = packed_S(m);
    const long Kd = packed_diag_elems(m), Kp = packed_total_elems(m);
    if (S % 16 || S < m || S >= m + 16 || Kd % 16 || Kp % 16 || Kd > Kp) {
      std::printf("m=%d: S=%d Kd=%ld Kp=%ld\n", m, S, Kd, Kp);
      ++bad;
    }
    std::vector<char> seen((size_t)Kp, 0);
    for (int c = 0; c < m; ++c)
      for (int r = (c / 16) * 16; r < S; ++r) {
        const long o = packed_lower_offset(r, c, S, Kd);
        if (o < 0 || o >= Kp || seen[(size_t)o]) { ++bad; continue; }
        seen[(size_t)o] = 1;
        if (((r / 16) == (c / 16)) != (o < Kd)) ++bad;                 // region
        if (r + 1 < S && packed_lower_offset(r + 1, c, S, Kd) != o + 1 && (r + 1) / 16 != c / 16 + 1) ++bad;   // contiguity
      }
    long covered = 0;
    for (char v : seen) covered += v;
    if (covered != Kp) { std::printf("m=%d: covered %ld of %ld\n", m, covered, Kp); ++bad; }
    // weights: sum over the diagonal region + 2 * sum over the rest == full Frobenius inner product of a
    // symmetric matrix (small integer entries, exact sums)
    long full = 0, d = 0, off = 0;
    for (int c = 0; c < m; ++c)
      for (int r = 0; r < m; ++r) {
        const long x = ((long)(r + 1) * (c + 1)) % 7 + 1;
        full += x * x;
        if (r / 16 == c / 16) d += x * x;
        else if (r / 16 > c / 16) off += x * x;
      }
    if (full != d + 2 * off) { std::printf("m=%d: weights\n", m); ++bad; }
  }
  std::printf("bad=%d\n", bad);
  return bad ? 1 : 0;
}
