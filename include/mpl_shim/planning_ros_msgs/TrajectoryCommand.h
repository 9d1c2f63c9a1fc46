/** @file TrajectoryCommand.h  (mplx shim of the generated message, planning_ros_msgs/msg/TrajectoryCommand.msg:1-7) */
#ifndef MPLX_SHIM_TRAJECTORY_COMMAND_MSG_H
#define MPLX_SHIM_TRAJECTORY_COMMAND_MSG_H
#include <planning_ros_msgs/Header.h>
namespace planning_ros_msgs {
struct TrajectoryCommand {
  HeaderLite header;
  Vector3 position, velocity, acceleration, jerk;
  double yaw = 0, yaw_dot = 0;
};
}  // namespace planning_ros_msgs
#endif
