/**
 * @file poly_solver.h  (mplx shim of <mpl_traj_solver/poly_solver.h>)
 * Minimum-derivative piecewise polynomial through a sequence of waypoints with given segment times: the refinement
 * step after the search (map_planner_node.cpp:224-227: TrajSolver3D(Control::JRK), setWaypoints, setDts, solve).
 * Host code, like the reference's (it is not part of the hot path).
 *
 * [UNVERIFIED against upstream: mpl_traj_solver lives in the absent motion_primitive_library submodule.]  What is
 * solved is the standard unconstrained QP of polynomial trajectory generation (Richter, Bry, Roy 2013): with smoothness
 * order s (derivatives 0..s continuous at every waypoint) and minimised derivative r, every segment is a polynomial of
 * degree N - 1, N = 2 (s + 1), fixed by the derivatives 0..s at its two ends; the cost sum_seg int |d^r p|^2 is a
 * quadratic form in the waypoint derivatives; the ones a waypoint's use_pos / use_vel / use_acc / use_jrk flags fix are
 * constants, the others minimise the form: R_pp d_p = -R_pf d_f, per axis.
 */
#ifndef MPLX_SHIM_POLY_SOLVER_H
#define MPLX_SHIM_POLY_SOLVER_H
#include <mpl_basis/trajectory.h>

#include <cmath>
#include <vector>

namespace mplx_shim {
/// dense row-major matrix with the two operations the solver needs
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int rows, int cols) : r(rows), c(cols), a((size_t)rows * cols, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};
/// solve A X = B by LU with partial pivoting (A square, B with any number of columns); false when singular
inline bool lu_solve(Mat A, Mat B, Mat &X) {
  const int n = A.r, m = B.c;
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A(i, k)) > std::fabs(A(p, k))) p = i;
    if (A(p, k) == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(A(k, j), A(p, j));
      for (int j = 0; j < m; j++) std::swap(B(k, j), B(p, j));
    }
    for (int i = k + 1; i < n; i++) {
      const double f = A(i, k) / A(k, k);
      if (f == 0.0) continue;
      for (int j = k; j < n; j++) A(i, j) -= f * A(k, j);
      for (int j = 0; j < m; j++) B(i, j) -= f * B(k, j);
    }
  }
  X = Mat(n, m);
  for (int j = 0; j < m; j++)
    for (int i = n - 1; i >= 0; i--) {
      double s = B(i, j);
      for (int k = i + 1; k < n; k++) s -= A(i, k) * X(k, j);
      X(i, j) = s / A(i, i);
    }
  return true;
}
}  // namespace mplx_shim

template <int Dim>
class PolySolver {
 public:
  /// smooth_derivative_order s: derivatives 0..s are continuous; minimize_derivative r: int |d^r p|^2 is minimised
  PolySolver(unsigned int smooth_derivative_order, unsigned int minimize_derivative, bool debug = false)
      : s_((int)smooth_derivative_order), N_(2 * ((int)smooth_derivative_order + 1)), R_((int)minimize_derivative), debug_(debug) {}

  /// monomial coefficients (ascending, N per segment and axis) of the last solve
  const std::vector<std::vector<Vecf<Dim>>> &coefficients() const { return coeff_; }
  const std::vector<decimal_t> &times() const { return dts_; }

  bool solve(const vec_E<Waypoint<Dim>> &waypoints, const std::vector<decimal_t> &dts) {
    using mplx_shim::Mat;
    coeff_.clear();
    dts_ = dts;
    const int W = (int)waypoints.size(), S = W - 1, K = s_ + 1;
    if (W < 2 || (int)dts.size() != S) return false;
    for (decimal_t t : dts)
      if (!(t > 0)) return false;
    // per segment: A (end derivatives from coefficients), its inverse, and H = A^-T Q A^-1
    std::vector<Mat> Ainv(S);
    Mat Rg(W * K, W * K);
    for (int i = 0; i < S; i++) {
      const double T = dts[i];
      Mat A(N_, N_), Q(N_, N_), I(N_, N_);
      for (int n = 0; n < N_; n++) {
        I(n, n) = 1.0;
        if (n < K) A(n, n) = falling(n, n);                       // d^n p (0) = n! a_n
        for (int k = 0; k < K; k++)
          if (k <= n) A(K + k, n) = falling(n, k) * ipow(T, n - k);  // d^k p (T)
        for (int r = 0; r < N_; r++)
          if (r >= R_ && n >= R_) Q(r, n) = falling(r, R_) * falling(n, R_) * ipow(T, r + n - 2 * R_ + 1) / (double)(r + n - 2 * R_ + 1);
      }
      if (!mplx_shim::lu_solve(A, I, Ainv[i])) return false;
      // H = Ainv^T Q Ainv, added to the rows / columns of the two waypoints' derivatives
      Mat QA(N_, N_);
      for (int a = 0; a < N_; a++)
        for (int b = 0; b < N_; b++) {
          double s = 0;
          for (int k = 0; k < N_; k++) s += Q(a, k) * Ainv[i](k, b);
          QA(a, b) = s;
        }
      for (int a = 0; a < N_; a++)
        for (int b = 0; b < N_; b++) {
          double s = 0;
          for (int k = 0; k < N_; k++) s += Ainv[i](k, a) * QA(k, b);
          Rg(i * K + a, i * K + b) += s;  // (end derivatives of segment i = global indices i K .. i K + 2 K - 1)
        }
    }
    // fixed / free derivatives
    std::vector<int> fixed, freed;
    for (int w = 0; w < W; w++)
      for (int k = 0; k < K; k++) {
        const Waypoint<Dim> &p = waypoints[w];
        const bool f = (k == 0 && p.use_pos) || (k == 1 && p.use_vel) || (k == 2 && p.use_acc) || (k == 3 && p.use_jrk);
        (f ? fixed : freed).push_back(w * K + k);
      }
    Mat D(W * K, Dim);
    for (int g : fixed) {
      const Waypoint<Dim> &p = waypoints[g / K];
      const Vecf<Dim> &v = g % K == 0 ? p.pos : g % K == 1 ? p.vel : g % K == 2 ? p.acc : p.jrk;
      for (int d = 0; d < Dim; d++) D(g, d) = v(d);
    }
    if (!freed.empty()) {
      const int nf = (int)fixed.size(), np = (int)freed.size();
      Mat Rpp(np, np), rhs(np, Dim), X;
      for (int a = 0; a < np; a++) {
        for (int b = 0; b < np; b++) Rpp(a, b) = Rg(freed[a], freed[b]);
        for (int d = 0; d < Dim; d++) {
          double s = 0;
          for (int b = 0; b < nf; b++) s += Rg(freed[a], fixed[b]) * D(fixed[b], d);
          rhs(a, d) = -s;
        }
      }
      if (!mplx_shim::lu_solve(Rpp, rhs, X)) return false;
      for (int a = 0; a < np; a++)
        for (int d = 0; d < Dim; d++) D(freed[a], d) = X(a, d);
    }
    // coefficients of every segment from its end derivatives
    coeff_.resize(S);
    for (int i = 0; i < S; i++) {
      coeff_[i].resize(N_);
      for (int n = 0; n < N_; n++)
        for (int d = 0; d < Dim; d++) {
          double s = 0;
          for (int k = 0; k < N_; k++) s += Ainv[i](n, k) * D(i * K + k, d);
          coeff_[i][n](d) = s;
        }
    }
    if (debug_) printf("[PolySolver] %d segments, %zu fixed and %zu free derivatives\n", S, fixed.size(), freed.size());
    return true;
  }

  /// the segments as Primitives: p(t) = sum a_n t^n = c0/120 t^5 + ... + c5, i.e. c(5 - n) = n! a_n; false when the
  /// polynomials have more than six coefficients (a Primitive holds a quintic)
  bool toPrimitives(vec_E<Primitive<Dim>> &prs) const {
    prs.clear();
    if (N_ > 6) return false;
    for (size_t i = 0; i < coeff_.size(); i++) {
      vec_E<Vec6f> cs(Dim);
      for (int n = 0; n < N_; n++)
        for (int d = 0; d < Dim; d++) cs[d](5 - n) = falling(n, n) * coeff_[i][n](d);
      prs.push_back(Primitive<Dim>(cs, dts_[i], Control::SNP));
    }
    return true;
  }

 private:
  static double falling(int n, int k) {  // n (n - 1) ... (n - k + 1)
    double v = 1;
    for (int m = 0; m < k; m++) v *= (double)(n - m);
    return v;
  }
  static double ipow(double t, int n) {
    double v = 1;
    for (int i = 0; i < n; i++) v *= t;
    return v;
  }
  int s_, N_, R_;
  bool debug_;
  std::vector<std::vector<Vecf<Dim>>> coeff_;
  std::vector<decimal_t> dts_;
};
#endif
