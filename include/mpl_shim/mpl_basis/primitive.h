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
  Primitive(const Waypoint<Dim> &p, const VecDf &u, decimal_t t) : t_(t), control_(p.control) {
    for (int i = 0; i < Dim; i++) {
      Vec6f c;
      if (p.control == Control::VEL) { c(4) = u(i); c(5) = p.pos(i); }
      else if (p.control == Control::ACC) { c(3) = u(i); c(4) = p.vel(i); c(5) = p.pos(i); }
      else if (p.control == Control::JRK) { c(2) = u(i); c(3) = p.acc(i); c(4) = p.vel(i); c(5) = p.pos(i); }
      else if (p.control == Control::SNP) { c(1) = u(i); c(2) = p.jrk(i); c(3) = p.acc(i); c(4) = p.vel(i); c(5) = p.pos(i); }
      else printf("Null Primitive, check the control set-up of the Waypoint!\n");
      prs_[i] = Primitive1D(c);
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
    return p;
  }
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
#endif
