/**
 * @file primitive.h  (mplx shim of <mpl_basis/primitive.h>)
 * Primitive<Dim>: six coefficients per axis, p(t) = c0/120 t^5 + c1/24 t^4 + c2/6 t^3 + c3/2 t^2 +
 * c4 t + c5 (primitive_geometry_utils.h:12-26).  The search itself never touches this class -- the
 * device builds and samples its own primitives -- it carries results to the caller
 * (pr(i).coeff(), t(), control(), evaluate for display / message conversion:
 * primitive_ros_utils.h:12-33, trajectory_extractor.hpp:10-30).
 */
#ifndef MPLX_SHIM_PRIMITIVE_H
#define MPLX_SHIM_PRIMITIVE_H
#include <mpl_basis/math.h>

#include <algorithm>
#include <mpl_basis/waypoint.h>

class Primitive1D {
 public:
  Primitive1D() {}
  explicit Primitive1D(const Vec6f &coeff) : c(coeff) {}
  Vec6f coeff() const { return c; }
  decimal_t p(decimal_t t) const {
    return c(0) / 120 * t * t * t * t * t + c(1) / 24 * t * t * t * t + c(2) / 6 * t * t * t + c(3) / 2 * t * t + c(4) * t + c(5);
  }
  decimal_t v(decimal_t t) const { return c(0) / 24 * t * t * t * t + c(1) / 6 * t * t * t + c(2) / 2 * t * t + c(3) * t + c(4); }
  decimal_t a(decimal_t t) const { return c(0) / 6 * t * t * t + c(1) / 2 * t * t + c(2) * t + c(3); }
  decimal_t j(decimal_t t) const { return c(0) / 2 * t * t + c(1) * t + c(2); }
  /// integral over [0, t] of the squared k-th derivative (k = 1 VEL .. 4 SNP): double sum over the derivative's
  /// monomial coefficients in ascending order -- the expression the device and the oracle evaluate (deviation D2)
  decimal_t J(decimal_t t, Control::Control control) const {
    const int k = (control & 15) == Control::VEL ? 1 : (control & 15) == Control::ACC ? 2 : (control & 15) == Control::JRK ? 3 : 4;
    const decimal_t fact[6] = {1, 1, 2, 6, 24, 120};
    decimal_t q[6];
    const int nq = 6 - k;
    for (int m = k; m <= 5; m++) q[m - k] = c(5 - m) / fact[m - k];
    decimal_t s = 0.0;
    for (int i = 0; i < nq; i++)
      for (int jj = 0; jj < nq; jj++) {
        decimal_t pw = t;
        for (int r = 1; r < i + jj + 1; r++) pw = pw * t;
        s += q[i] * q[jj] * pw / (decimal_t)(i + jj + 1);
      }
    return s;
  }
  /// times in (0, t) where the k-th derivative is stationary (roots of derivative k + 1), as `solve` returns them
  std::vector<decimal_t> extrema(int k, decimal_t t) const {
    std::vector<decimal_t> ts = k == 1 ? solve(0, c(0) / 6, c(1) / 2, c(2), c(3)) : k == 2 ? solve(0, 0, c(0) / 2, c(1), c(2)) : solve(0, 0, 0, c(0), c(1));
    std::vector<decimal_t> out;
    for (decimal_t x : ts)
      if (x > 0 && x < t) out.push_back(x);
    return out;
  }
  /// max over [0, t] of |d^k p| for k = 1 (vel), 2 (acc), 3 (jrk): end points and interior stationary points
  decimal_t max_abs(int k, decimal_t t) const {
    auto f = [&](decimal_t x) { return k == 1 ? v(x) : k == 2 ? a(x) : j(x); };
    decimal_t m = std::max(std::fabs(f(0)), std::fabs(f(t)));
    for (decimal_t x : extrema(k, t)) m = std::max(m, std::fabs(f(x)));
    return m;
  }

 private:
  Vec6f c;
};

template <int Dim>
class Primitive {
 public:
  Primitive() {}
  /// from coefficient vectors (primitive_ros_utils.h:128,146)
  Primitive(const vec_E<Vec6f> &cs, decimal_t t, Control::Control control) : t_(t), control_(control) {
    for (int i = 0; i < Dim; i++) prs_[i] = Primitive1D(cs[i]);
    if (cs.size() > (size_t)Dim) pr_yaw_ = Primitive1D(cs.back());  // 4th entry of the message form = yaw
  }
  /// from a state and a control input (test_primitive_collide.cpp:15, obstacle_config.hpp:24)
  /// a use_yaw state adds the yaw channel yaw(t) = yaw0 + u(Dim) t (Vec4f inputs, map_planner_node.cpp:125-126)
  Primitive(const Waypoint<Dim> &p, const VecDf &u, decimal_t t) : t_(t), control_(p.control) {
    const int kind = (int)p.control & 15;
    for (int i = 0; i < Dim; i++) {
      Vec6f c;
      if (kind == Control::VEL) { c(4) = u(i); c(5) = p.pos(i); }
      else if (kind == Control::ACC) { c(3) = u(i); c(4) = p.vel(i); c(5) = p.pos(i); }
      else if (kind == Control::JRK) { c(2) = u(i); c(3) = p.acc(i); c(4) = p.vel(i); c(5) = p.pos(i); }
      else if (kind == Control::SNP) { c(1) = u(i); c(2) = p.jrk(i); c(3) = p.acc(i); c(4) = p.vel(i); c(5) = p.pos(i); }
      else printf("Null Primitive, check the control set-up of the Waypoint!\n");
      prs_[i] = Primitive1D(c);
    }
    if (p.use_yaw) {
      Vec6f c;
      c(4) = (int)u.size() > Dim ? u(Dim) : 0.0;
      c(5) = p.yaw;
      pr_yaw_ = Primitive1D(c);
    }
  }
  Primitive1D pr(int k) const { return prs_[k]; }
  Primitive1D pr_yaw() const { return pr_yaw_; }
  decimal_t t() const { return t_; }
  Control::Control control() const { return control_; }
  Waypoint<Dim> evaluate(decimal_t t) const {
    Waypoint<Dim> p(control_);
    for (int k = 0; k < Dim; k++) {
      p.pos(k) = prs_[k].p(t);
      p.vel(k) = prs_[k].v(t);
      p.acc(k) = prs_[k].a(t);
      p.jrk(k) = prs_[k].j(t);
    }
    if (p.use_yaw) {  // [UNVERIFIED upstream: the yaw of a state is kept in [-pi, pi]]
      const decimal_t pi = 3.141592653589793;
      decimal_t q = pr_yaw_.p(t);
      while (q > pi) q -= 2 * pi;
      while (q < -pi) q += 2 * pi;
      p.yaw = q;
    }
    return p;
  }
  /// sum over the axes of the integral of the squared derivative `control` selects (env_poly_map.h:72, map_planner_node.cpp:213)
  decimal_t J(const Control::Control &control) const {
    decimal_t j = 0;
    for (int k = 0; k < Dim; k++) j += prs_[k].J(t_, control);
    return j;
  }
  decimal_t Jyaw() const { return pr_yaw_.J(t_, Control::VEL); }
  /// max |velocity| / |acceleration| / |jerk| along axis k (ellipsoid_util.h:68)
  decimal_t max_vel(int k) const { return prs_[k].max_abs(1, t_); }
  decimal_t max_acc(int k) const { return prs_[k].max_abs(2, t_); }
  decimal_t max_jrk(int k) const { return prs_[k].max_abs(3, t_); }
  vec_E<Waypoint<Dim>> sample(int N) const {
    vec_E<Waypoint<Dim>> ps(N + 1);
    decimal_t dt = t_ / N;
    for (int i = 0; i <= N; i++) ps[i] = evaluate(i * dt);
    return ps;
  }

 private:
  Primitive1D prs_[Dim];
  Primitive1D pr_yaw_;
  decimal_t t_{0};
  Control::Control control_{Control::NONE};
};
typedef Primitive<2> Primitive2D;
typedef Primitive<3> Primitive3D;

/// validate_primitive(pr, v_max, a_max, j_max[, yaw_max]) (env_poly_map.h:58, env_cloud.h:62): a control kind checks the
/// derivatives below its own order (ACC: vel; JRK: vel, acc; SNP: vel, acc, jrk); a limit <= 0 is not checked
/// [UNVERIFIED rule, same as the device's].  The yaw threshold is applied by the device search (validate_yaw of
/// mplx_math.h); this host-side helper checks the derivative limits only.
template <int Dim>
bool validate_primitive(const Primitive<Dim> &pr, decimal_t mv = 0, decimal_t ma = 0, decimal_t mj = 0, decimal_t myaw = 0) {
  const int c = pr.control() & 15;
  if (c == Control::ACC || c == Control::JRK || c == Control::SNP)
    for (int i = 0; i < Dim; i++)
      if (mv > 0 && pr.max_vel(i) > mv) return false;
  if (c == Control::JRK || c == Control::SNP)
    for (int i = 0; i < Dim; i++)
      if (ma > 0 && pr.max_acc(i) > ma) return false;
  if (c == Control::SNP)
    for (int i = 0; i < Dim; i++)
      if (mj > 0 && pr.max_jrk(i) > mj) return false;
  (void)myaw;
  return true;
}
template <int Dim>
void print(const Primitive<Dim> &p) {
  for (int i = 0; i < Dim; i++) {
    const Vec6f c = p.pr(i).coeff();
    printf("dim[%d]: %f %f %f %f %f %f\n", i, c(0), c(1), c(2), c(3), c(4), c(5));
  }
  printf("t: %f\n", p.t());
}
#endif
