/** @file Trajectory.h  (mplx shim of the generated message, planning_ros_msgs/msg/Trajectory.msg:1-5) */
#ifndef MPLX_SHIM_TRAJECTORY_MSG_H
#define MPLX_SHIM_TRAJECTORY_MSG_H
#include <planning_ros_msgs/Header.h>
#include <planning_ros_msgs/LambdaSeg.h>
#include <planning_ros_msgs/Primitive.h>
namespace planning_ros_msgs {
struct Trajectory {
  HeaderLite header;
  std::vector<Primitive> primitives;
  std::vector<LambdaSeg> lambda;
};
}  // namespace planning_ros_msgs
#endif
