/**
 * @file math.h  (mplx shim of <mpl_basis/math.h>): real roots of polynomials up to degree 5,
 * `solve(a, b, c, d, e, f)` = roots of a t^5 + b t^4 + c t^3 + d t^2 + e t + f (primitive_geometry_utils.h:37,83,160).
 *
 * [UNVERIFIED recollection] upstream picks the formula by the first non-zero leading coefficient: linear, the
 * quadratic formula ((-c - sqrt(D)) / (2b) first), Cardano (acos / cos / cbrt) for cubics, Ferrari for quartics
 * and an Eigen companion-matrix eigen-solve for quintics.  Here: the same selection; degree <= 2 with upstream's
 * formulas (the only degrees the ACC / VEL lattices produce: a = b = c = 0); degree >= 3 by derivative-chain
 * isolation + bisection/Newton from + - * / only (deviation D1 of oracle/mpl_oracle.h), roots in ascending order.
 */
#ifndef MPLX_SHIM_MATH_H
#define MPLX_SHIM_MATH_H
#include <mpl_basis/data_type.h>

namespace mplx_shim {
inline decimal_t poly_eval(const decimal_t *a, int n, decimal_t x) {
  decimal_t r = a[n];
  for (int i = n - 1; i >= 0; i--) r = r * x + a[i];
  return r;
}
/// all real roots of sum a[i] x^i (degree n >= 1, a[n] != 0), ascending
inline std::vector<decimal_t> real_roots(const decimal_t *a, int n) {
  std::vector<decimal_t> out;
  if (n == 1) { out.push_back(-a[0] / a[1]); return out; }
  decimal_t m = 0;
  for (int i = 0; i < n; i++) m = std::fabs(a[i] / a[n]) > m ? std::fabs(a[i] / a[n]) : m;
  const decimal_t bound = 1.0 + m;  // Cauchy
  std::vector<decimal_t> d(n);
  for (int i = 1; i <= n; i++) d[i - 1] = a[i] * i;
  int nd = n - 1;
  while (nd > 0 && d[nd] == 0.0) nd--;
  std::vector<decimal_t> crit = nd >= 1 ? real_roots(d.data(), nd) : std::vector<decimal_t>();
  std::vector<decimal_t> xs;
  xs.push_back(-bound);
  for (decimal_t c : crit) if (c > -bound && c < bound) xs.push_back(c);
  xs.push_back(bound);
  for (size_t k = 0; k + 1 < xs.size(); k++) {
    decimal_t lo = xs[k], hi = xs[k + 1], flo = poly_eval(a, n, lo), fhi = poly_eval(a, n, hi);
    if (flo == 0.0) { if (out.empty() || out.back() != lo) out.push_back(lo); continue; }
    if (fhi == 0.0 || (flo < 0) == (fhi < 0)) { if (fhi == 0.0 && k + 2 == xs.size()) out.push_back(hi); continue; }
    for (int it = 0; it < 200 && hi - lo > 4e-16 * std::fabs(hi + lo); it++) {
      const decimal_t mid = 0.5 * (lo + hi), fm = poly_eval(a, n, mid);
      if (fm == 0.0) { lo = hi = mid; break; }
      if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; } else hi = mid;
    }
    out.push_back(0.5 * (lo + hi));
  }
  return out;
}
}  // namespace mplx_shim

/// roots of b t^2 + c t + d (b != 0)
inline std::vector<decimal_t> quad(decimal_t b, decimal_t c, decimal_t d) {
  std::vector<decimal_t> dts;
  const decimal_t p = c * c - 4 * b * d;
  if (p < 0) return dts;
  dts.push_back((-c - sqrt(p)) / (2 * b));
  dts.push_back((-c + sqrt(p)) / (2 * b));
  return dts;
}
inline std::vector<decimal_t> cubic(decimal_t a, decimal_t b, decimal_t c, decimal_t d) {
  const decimal_t co[4] = {d, c, b, a};
  return mplx_shim::real_roots(co, 3);
}
inline std::vector<decimal_t> quartic(decimal_t a, decimal_t b, decimal_t c, decimal_t d, decimal_t e) {
  const decimal_t co[5] = {e, d, c, b, a};
  return mplx_shim::real_roots(co, 4);
}
/// a t^4 + b t^3 + c t^2 + d t + e = 0
inline std::vector<decimal_t> solve(decimal_t a, decimal_t b, decimal_t c, decimal_t d, decimal_t e) {
  std::vector<decimal_t> ts;
  if (a != 0) return quartic(a, b, c, d, e);
  if (b != 0) return cubic(b, c, d, e);
  if (c != 0) return quad(c, d, e);
  if (d != 0) { ts.push_back(-e / d); return ts; }
  return ts;
}
/// a t^5 + b t^4 + c t^3 + d t^2 + e t + f = 0
inline std::vector<decimal_t> solve(decimal_t a, decimal_t b, decimal_t c, decimal_t d, decimal_t e, decimal_t f) {
  if (a == 0) return solve(b, c, d, e, f);
  const decimal_t co[6] = {f, e, d, c, b, a};
  return mplx_shim::real_roots(co, 5);
}
#endif
