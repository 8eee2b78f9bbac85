// Host-side eigenvalues of the Lanczos tridiagonal matrices (lanczos.hip, ipstep.hip).
//
// Every batch of Lanczos steps ends with the host looking at T_m = tridiag(b, a, b): the smallest (or k-th) eigenvalue by
// bisection on the Sturm count, then the residual bound of its Ritz pair.  While the host does that the stream is empty,
// so the analysis is part of the step-length rule's and the H_alpha setup's wall time (reference: eigmin(XXX),
// src/predictor_corrector.jl:272,285; eigen(W), src/Solvers.jl:642,706).  Rounds 1-4 counted with the pivots of the LDL'
// factorisation, d_i = a_i - x - b_{i-1}^2 / d_{i-1}: one division on the dependency chain per row, ~8 ns per row, ~75
// bisection steps from the Gershgorin interval down to one ulp -- 60 us per eigenvalue at m = 100, 150 us at m = 240, per
// batch, and growing with the run (maxG11: 7.4 us per Lanczos step in runs of 64 steps, 13.4 us in runs of 120).  Here:
//  * the count runs on the characteristic polynomials of the leading blocks, p_i = (a_i - x) p_{i-1} - b_{i-1}^2 p_{i-2}
//    (a multiply and an FMA on the chain, rescaled by powers of two, which leaves the signs alone);
//  * the bracket starts from what the caller knows: the eigenvalue of the previous batch's T is an upper bound of the
//    same eigenvalue of this batch's (Cauchy interlacing: T_m is a leading block of T_{m+16}); the lower end is found by
//    stepping down from there.  Every bound is CHECKED by a count before it is used, so a wrong hint costs evaluations,
//    never the result.
// The result is the same number as before up to the last bits: bisection down to neighbouring doubles.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace lrn {

struct TriEig {
  int m = 0;
  const double* a = nullptr;
  std::vector<double> b2;          // (b_i / sigma)^2
  double glo = 0.0, ghi = 0.0;     // Gershgorin interval
  double inv = 1.0;                // 1 / sigma, sigma = the power of two at the matrix' norm: the recurrence runs on T / sigma
  long evals = 0;                  // Sturm counts evaluated (measurement)

  TriEig(const double* a_, const double* b_, int m_) : m(m_), a(a_), b2((size_t)std::max(0, m_ - 1)) {
    glo = ghi = m > 0 ? a[0] : 0.0;
    for (int i = 0; i < m; ++i) {
      const double r = (i > 0 ? std::fabs(b_[i - 1]) : 0.0) + (i < m - 1 ? std::fabs(b_[i]) : 0.0);
      glo = std::min(glo, a[i] - r);
      ghi = std::max(ghi, a[i] + r);
    }
    const double nrm = std::max(std::fabs(glo), std::fabs(ghi));
    if (nrm > 0.0 && nrm <= 1.79e308) inv = std::scalbn(1.0, -std::ilogb(nrm));
    for (int i = 0; i + 1 < m; ++i) { const double t = b_[i] * inv; b2[i] = t * t; }
  }

  // number of eigenvalues < x: sign changes of p_0 = 1, p_1, .., p_m.  An exact zero is replaced by a tiny value of the
  // sign opposite to its predecessor's -- the pivot p_i / p_{i-1} = 0 counted as negative, LAPACK's pivmin rule (dlaebz):
  // at x equal to an eigenvalue of a leading block the count includes it, and a block-diagonal T (b_i = 0: the restarts of
  // lanczos_extremes) is counted block by block.
  int below(double x) {
    ++evals;
    // T / sigma has entries <= 2: a row changes the magnitude of p by a factor between rounding level and ~4
    constexpr double BIG = 0x1p+400, SMALL = 0x1p-400, PIVMIN = 0x1p-300;
    double pm = 1.0, p = (a[0] - x) * inv;
    if (p == 0.0) p = -PIVMIN;
    int cnt = p < 0.0 ? 1 : 0;
    for (int i = 1; i < m; ++i) {
      double pn = ((a[i] - x) * inv) * p - b2[i - 1] * pm;
      if (pn == 0.0) pn = -PIVMIN * p;
      cnt += (pn < 0.0) != (p < 0.0);
      pm = p;
      p = pn;
      const double ap = std::fabs(p), am = std::fabs(pm);
      const double mx = ap > am ? ap : am;
      if (mx > BIG) { p *= SMALL; pm *= SMALL; }           // (powers of two: the signs, and every later ratio, stay as they are)
      else if (mx < SMALL) { p *= BIG; pm *= BIG; }
    }
    return cnt;
  }

  // k-th smallest eigenvalue (k = 0 .. m-1).  upper (may be null): a value believed to be >= that eigenvalue (the same
  // eigenvalue of a leading block); width: the distance the caller expects it to have moved (<= 0: unknown).
  double kth(int k, const double* upper = nullptr, double width = 0.0) {
    if (m <= 0) return 0.0;
    double lo = glo, hi = ghi;
    if (!(lo == lo) || !(hi == hi)) return 0.5 * (lo + hi);      // NaN in the matrix: the callers look at their scale
    const double span = hi - lo;
    if (upper && *upper > lo && *upper <= hi && span > 0.0) {
      // a few ulps above the hint: the hint itself came out of a bisection
      double h = *upper + 8.0 * 2.220446049250313e-16 * std::max(std::fabs(*upper), std::fabs(lo) + std::fabs(hi));
      if (h < hi && below(h) >= k + 1) {
        hi = h;
        double w = width > 0.0 ? width : 1e-8 * span;
        w = std::max(w, 64.0 * 2.220446049250313e-16 * (std::fabs(lo) + std::fabs(hi)));
        for (int probe = 0; probe < 64; ++probe) {
          const double t = hi - w;
          if (!(t > lo)) break;
          if (below(t) >= k + 1) hi = t; else { lo = t; break; }
          w *= 8.0;
        }
      }
    }
    for (int it = 0; it < 200; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (mid == lo || mid == hi) break;
      if (below(mid) >= k + 1) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
  }
};

inline double tri_eig_kth(const std::vector<double>& a, const std::vector<double>& b, int m, int k,
                          const double* upper = nullptr, double width = 0.0) {
  TriEig t(a.data(), b.data(), m);
  return t.kth(k, upper, width);
}

// largest eigenvalue: - smallest of -T (same off-diagonal); lower (may be null): a value believed to be <= it
inline double tri_eig_max(const std::vector<double>& a, const std::vector<double>& b, int m, const double* lower = nullptr,
                          double width = 0.0) {
  std::vector<double> an((size_t)m);
  for (int i = 0; i < m; ++i) an[i] = -a[i];
  TriEig t(an.data(), b.data(), m);
  double up = 0.0;
  if (lower) up = -*lower;
  return -t.kth(0, lower ? &up : nullptr, width);
}

}  // namespace lrn
