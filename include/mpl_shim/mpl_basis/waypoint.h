/**
 * @file waypoint.h  (mplx shim of <mpl_basis/waypoint.h>)
 * Waypoint<Dim>: the search-state record.  Fields used in-tree: pos, vel, acc, jrk, yaw, t, enable_t,
 * use_pos/use_vel/use_acc/use_jrk/use_yaw and control (map_planner_node.cpp:155-171,
 * env_poly_map.h:63-64); the use_* flags and `control` share storage, so `use_pos = use_vel = true`
 * IS Control::ACC and `Waypoint3D goal(start.control)` copies the flag set (map_planner_node.cpp:167).
 */
#ifndef MPLX_SHIM_WAYPOINT_H
#define MPLX_SHIM_WAYPOINT_H
#include <mpl_basis/control.h>
#include <mpl_basis/data_type.h>

#include <functional>

template <int Dim>
struct Waypoint {
  Waypoint() : control(Control::NONE) {}
  Waypoint(Control::Control c) : control(c) {}
  Vecf<Dim> pos, vel, acc, jrk;
  decimal_t yaw{0};
  decimal_t t{0};
  union {
    struct {
      bool use_pos : 1;
      bool use_vel : 1;
      bool use_acc : 1;
      bool use_jrk : 1;
      bool use_yaw : 1;
    };
    Control::Control control : 5;
  };
  bool enable_t{false};
  /// Quantised key of the state: pos / 0.01, vel acc jrk / 0.1 per axis for the enabled fields, yaw / 0.1, t / 0.1 when
  /// enable_t [UNVERIFIED resolutions, the ones the device and the oracle use].  Two waypoints are equal when their
  /// keys are (upstream compares boost::hash_combine of the same integers: deviation D3 of oracle/mpl_oracle.h).
  std::vector<int> key() const {
    std::vector<int> k;
    for (int i = 0; i < Dim; i++) {
      if (use_pos) k.push_back((int)std::round(pos(i) / 0.01));
      if (use_vel) k.push_back((int)std::round(vel(i) / 0.1));
      if (use_acc) k.push_back((int)std::round(acc(i) / 0.1));
      if (use_jrk) k.push_back((int)std::round(jrk(i) / 0.1));
    }
    if (use_yaw) k.push_back((int)std::round(yaw / 0.1));
    if (enable_t) k.push_back((int)std::round(t / 0.1));
    return k;
  }
  bool operator==(const Waypoint<Dim> &n) const { return key() == n.key(); }
  bool operator!=(const Waypoint<Dim> &n) const { return !(*this == n); }
  void print(const char *str = "") const {
    printf("%s pos: ", str);
    for (int i = 0; i < Dim; i++) printf("%f ", pos(i));
    printf(" vel: ");
    for (int i = 0; i < Dim; i++) printf("%f ", vel(i));
    printf(" t: %f\n", t);
  }
};
typedef Waypoint<2> Waypoint2D;
typedef Waypoint<3> Waypoint3D;

/// hash of the key, boost::hash_combine style (hash_value(Waypoint) upstream)
template <int Dim>
std::size_t hash_value(const Waypoint<Dim> &w) {
  std::size_t val = 0;
  for (int id : w.key()) val ^= std::hash<int>()(id) + 0x9e3779b9 + (val << 6) + (val >> 2);
  return val;
}
namespace mplx_shim {
template <int Dim>
struct WaypointHash {
  std::size_t operator()(const Waypoint<Dim> &w) const { return hash_value(w); }
};
}  // namespace mplx_shim
#endif
