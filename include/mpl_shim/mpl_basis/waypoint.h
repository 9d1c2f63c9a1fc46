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
#endif
