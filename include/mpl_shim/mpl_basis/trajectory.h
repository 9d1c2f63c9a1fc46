/**
 * @file trajectory.h  (mplx shim of <mpl_basis/trajectory.h>)
 * Trajectory<Dim>: the piecewise primitive sequence returned by getTraj().  Public members as the
 * ROS glue needs them (segs, taus, Ts, total_t_: primitive_ros_utils.h:77,154-179); no time
 * re-scaling (lambda) in this shim.
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
      c.t = i * dt;
      ps.push_back(c);
    }
    return ps;
  }
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
};
typedef Trajectory<2> Trajectory2D;
typedef Trajectory<3> Trajectory3D;
#endif
