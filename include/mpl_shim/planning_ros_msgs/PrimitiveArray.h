/** @file PrimitiveArray.h  (mplx shim of the generated message, planning_ros_msgs/msg/PrimitiveArray.msg) */
#ifndef MPLX_SHIM_PRIMITIVE_ARRAY_MSG_H
#define MPLX_SHIM_PRIMITIVE_ARRAY_MSG_H
#include <planning_ros_msgs/Header.h>
#include <planning_ros_msgs/Primitive.h>
namespace planning_ros_msgs {
struct PrimitiveArray {
  HeaderLite header;
  std::vector<Primitive> primitives;
};
}  // namespace planning_ros_msgs
#endif
