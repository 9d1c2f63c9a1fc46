// Host-side (CPU) bit-exactness of the moving-obstacle environment's general root finder (mpl_ros_amd/csrc/mplx_poly_dev.h:
// solve_any6, poly_max_abs -- the host build of the device code) against the statement it restates: solve() and
// Primitive1D::max_abs of include/mpl_shim (the header the compiled-reference checker of tests/test_poly_map.py links).
#include <mpl_basis/primitive.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../mpl_ros_amd/csrc/mplx_poly_dev.h"

static uint64_t sm = 0x9E3779B97F4A7C15ull;
static uint64_t next() { uint64_t z = (sm += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double uni(double lo, double hi) { return lo + (hi - lo) * (double)(next() >> 11) / 9007199254740992.0; }
static bool same(double a, double b) { return memcmp(&a, &b, 8) == 0; }

int main() {
  long fails = 0, cases = 0, roots = 0;
  for (int it = 0; it < 400000; it++) {
    double c[6];
    const int mode = (int)(next() % 7);
    for (int i = 0; i < 6; i++) c[i] = uni(-3, 3);
    if (mode == 1) c[0] = 0;                                     // quartic
    if (mode == 2) c[0] = c[1] = 0;                              // cubic (a JRK primitive against a hyperplane)
    if (mode == 3) c[0] = c[1] = c[2] = 0;                       // quadratic
    if (mode == 4) { c[0] = c[1] = 0; for (int i = 2; i < 6; i++) c[i] = (double)((long)(next() % 9) - 4) / 2; }  // lattice coefficients: exact roots, roots at interval ends
    if (mode == 5) { const double r = (double)((long)(next() % 7) - 3); c[0] = c[1] = 0; c[2] = 1; c[3] = -3 * r; c[4] = 3 * r * r; c[5] = -r * r * r; }  // triple root
    if (mode == 6) { c[0] = c[1] = 0; c[2] = uni(-1e-6, 1e-6); }  // nearly quadratic: a huge Cauchy bound
    const std::vector<decimal_t> want = solve(c[0], c[1], c[2], c[3], c[4], c[5]);
    double ts[8];
    const int n = mplx::solve_any6(c[0], c[1], c[2], c[3], c[4], c[5], ts);
    bool ok = n == (int)want.size();
    for (int i = 0; ok && i < n; i++) ok = same(ts[i], want[i]);
    cases++;
    roots += n;
    if (!ok && fails++ < 5) {
      printf("solve mismatch (mode %d): %.17g %.17g %.17g %.17g %.17g %.17g -> %d roots vs %zu\n", mode, c[0], c[1], c[2], c[3], c[4], c[5], n, want.size());
      for (int i = 0; i < n; i++) printf("  got  %.17g\n", ts[i]);
      for (double w : want) printf("  want %.17g\n", w);
    }
    // max |vel| / |acc| / |jrk| over [0, T] of the same coefficients read as a Primitive1D
    Vec6f v6;
    for (int i = 0; i < 6; i++) v6(i) = c[i];
    const Primitive1D pr(v6);
    const double T = uni(0.1, 2.0);
    for (int k = 1; k <= 3; k++) {
      const double a = mplx::poly_max_abs(c, k, T), b = pr.max_abs(k, T);
      if (!same(a, b) && fails++ < 5) printf("max_abs(%d) mismatch: %.17g vs %.17g\n", k, a, b);
    }
  }
  printf("%ld polynomials, %ld roots; %s (%ld failures)\n", cases, roots, fails ? "FAILED" : "OK", fails);
  return fails ? 1 : 0;
}
