/**
 * @file trajectory.h  (mplx shim of <mpl_basis/trajectory.h>)
 * Trajectory<Dim>: the piecewise primitive sequence returned by getTraj().  Public members as the
 * ROS glue needs them (segs, taus, Ts, total_t_, lambda_: primitive_ros_utils.h:77-89,154-179).  Nothing in-tree
 * ever creates a time re-scaling: Lambda is carried (and applied when a message brings one) so that the glue compiles
 * and round-trips.
 */
#ifndef MPLX_SHIM_TRAJECTORY_H
#define MPLX_SHIM_TRAJECTORY_H
#include <mpl_basis/primitive.h>

/// one sample of a trajectory: what TrajectoryExtractor turns into a TrajectoryCommand message
/// (planning_ros_utils/src/planning_utils/trajectory_extractor.hpp:13-30 reads pos, vel, acc, jrk, yaw, yaw_dot, t)
template <int Dim>
struct Command {
  Vecf<Dim> pos, vel, acc, jrk;
  decimal_t yaw{0}, yaw_dot{0};
  decimal_t t{0};
};

/// One segment of the time re-scaling (planning_ros_msgs/msg/LambdaSeg.msg: dT, ti, tf, ca[4]; primitive_ros_utils.h:
/// 80-89,166-176).  [UNVERIFIED upstream lambda.h] the virtual-time rate over [ti, tf] is the cubic a(0) tau^3 + a(1) tau^2
/// + a(2) tau + a(3); real time advances by its integral.
struct LambdaSeg {
  Vec4f a;
  decimal_t ti{0}, tf{0}, dT{0};
  decimal_t getT(decimal_t tau) const { return a(0) / 4 * tau * tau * tau * tau + a(1) / 3 * tau * tau * tau + a(2) / 2 * tau * tau + a(3) * tau; }
};
class Lambda {
 public:
  bool exist() const { return !segs.empty(); }
  /// real time at virtual time tau
  decimal_t getT(decimal_t tau) const {
    decimal_t t = 0;
    for (const auto &seg : segs) {
      if (tau >= seg.tf) { t += seg.dT; continue; }
      if (tau > seg.ti) t += seg.getT(tau) - seg.getT(seg.ti);
      break;
    }
    return t;
  }
  /// virtual time at real time t (bisection inside the segment that holds it)
  decimal_t getTau(decimal_t t) const {
    decimal_t acc = 0;
    for (const auto &seg : segs) {
      if (t > acc + seg.dT) { acc += seg.dT; continue; }
      decimal_t lo = seg.ti, hi = seg.tf;
      for (int it = 0; it < 80; it++) {
        const decimal_t mid = (lo + hi) / 2;
        if (acc + seg.getT(mid) - seg.getT(seg.ti) < t) lo = mid; else hi = mid;
      }
      return (lo + hi) / 2;
    }
    return segs.empty() ? t : segs.back().tf;
  }
  std::vector<LambdaSeg> segs;
};

template <int Dim>
class Trajectory {
 public:
  Trajectory() : total_t_(0) {}
  /// from primitives (obstacle_config.hpp:36)
  Trajectory(const vec_E<Primitive<Dim>> &prs) : segs(prs), total_t_(0) {
    taus.push_back(0);
    for (const auto &pr : prs) taus.push_back(pr.t() + taus.back());
    Ts = taus;
    total_t_ = taus.back();
  }
  vec_E<Primitive<Dim>> getPrimitives() const { return segs; }
  decimal_t getTotalTime() const { return total_t_; }
  std::vector<decimal_t> getSegmentTimes() const {
    std::vector<decimal_t> dts;
    for (size_t i = 0; i + 1 < Ts.size(); i++) dts.push_back(Ts[i + 1] - Ts[i]);
    return dts;
  }
  /// waypoints at the segment joints (map_planner_node.cpp:217)
  vec_E<Waypoint<Dim>> getWaypoints() const {
    vec_E<Waypoint<Dim>> ws;
    if (segs.empty()) return ws;
    decimal_t t = 0;
    for (const auto &seg : segs) {
      ws.push_back(seg.evaluate(0));
      ws.back().t = t;
      t += seg.t();
    }
    ws.push_back(segs.back().evaluate(segs.back().t()));
    ws.back().t = t;
    return ws;
  }
  /// state at time t (robot.hpp:96)
  Waypoint<Dim> evaluate(decimal_t time) const {
    if (segs.empty()) return Waypoint<Dim>();
    decimal_t tau = time < 0 ? 0 : (time > total_t_ ? total_t_ : time);
    for (size_t id = 0; id < segs.size(); id++) {
      if ((tau >= taus[id] && tau < taus[id + 1]) || id + 1 == segs.size()) {
        Waypoint<Dim> p = segs[id].evaluate(tau - taus[id]);
        p.t = time;
        return p;
      }
    }
    return Waypoint<Dim>();
  }
  /// N + 1 equally spaced samples over the whole trajectory (trajectory_extractor.hpp:9-10: N = ceil(total / dt))
  vec_E<Command<Dim>> sample(int N) const {
    vec_E<Command<Dim>> ps;
    if (segs.empty() || N <= 0) return ps;
    const decimal_t dt = total_t_ / N;
    for (int i = 0; i <= N; i++) {
      const Waypoint<Dim> w = evaluate(i * dt);
      Command<Dim> c;
      c.pos = w.pos; c.vel = w.vel; c.acc = w.acc; c.jrk = w.jrk;
      c.yaw = w.yaw;
      c.yaw_dot = yaw_rate(i * dt);
      c.t = i * dt;
      ps.push_back(c);
    }
    return ps;
  }
  /// state at time t as a Command (primitive_ellipsoid_utils.h:67-68,84-85); false outside [0, total time]
  bool evaluate(decimal_t time, Command<Dim> &c) const {
    if (segs.empty() || time < 0 || time > total_t_) return false;
    const Waypoint<Dim> w = evaluate(time);
    c.pos = w.pos; c.vel = w.vel; c.acc = w.acc; c.jrk = w.jrk;
    c.yaw = w.yaw;
    c.yaw_dot = yaw_rate(time);
    c.t = time;
    return true;
  }
  Lambda lambda() const { return lambda_; }
  /// total control effort of the derivative `control` selects / of yaw (map_planner_node.cpp:210-214)
  decimal_t J(const Control::Control &control) const {
    decimal_t j = 0;
    for (const auto &seg : segs) j += seg.J(control);
    return j;
  }
  decimal_t Jyaw() const {
    decimal_t j = 0;
    for (const auto &seg : segs) j += seg.Jyaw();
    return j;
  }
  vec_E<Primitive<Dim>> segs;
  std::vector<decimal_t> taus;
  std::vector<decimal_t> Ts;
  decimal_t total_t_;
  Lambda lambda_;

 private:
  decimal_t yaw_rate(decimal_t time) const {
    const decimal_t tau = time < 0 ? 0 : (time > total_t_ ? total_t_ : time);
    for (size_t id = 0; id < segs.size(); id++)
      if ((tau >= taus[id] && tau < taus[id + 1]) || id + 1 == segs.size()) return segs[id].pr_yaw().v(tau - taus[id]);
    return 0;
  }
};
typedef Trajectory<2> Trajectory2D;
typedef Trajectory<3> Trajectory3D;
#endif
