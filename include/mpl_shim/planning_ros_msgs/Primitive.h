/** @file Primitive.h  (mplx shim of the generated message, planning_ros_msgs/msg/Primitive.msg:1-8) */
#ifndef MPLX_SHIM_PRIMITIVE_MSG_H
#define MPLX_SHIM_PRIMITIVE_MSG_H
#include <vector>
namespace planning_ros_msgs {
struct Primitive {
  std::vector<double> cx, cy, cz, cyaw;
  double t = 0;
};
}  // namespace planning_ros_msgs
#endif
