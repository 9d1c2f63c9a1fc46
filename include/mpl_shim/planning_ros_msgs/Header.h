/**
 * @file Header.h  (mplx shim): what the in-tree glue touches of std_msgs/Header and ros::Time -- there is no ROS here.
 * `header.frame_id`, `header.stamp = t0 + ros::Duration(t)` and `ros::Time::now()` (trajectory_extractor.hpp:12,15).
 * ros::Time::now() returns 0: a sampled trajectory's stamps are then the sample times themselves.
 */
#ifndef MPLX_SHIM_HEADER_MSG_H
#define MPLX_SHIM_HEADER_MSG_H
#include <string>

namespace ros {
struct Duration {
  double sec;
  explicit Duration(double s = 0) : sec(s) {}
  double toSec() const { return sec; }
};
struct Time {
  double sec;
  explicit Time(double s = 0) : sec(s) {}
  static Time now() { return Time(0); }
  double toSec() const { return sec; }
  Time operator+(const Duration &d) const { return Time(sec + d.sec); }
};
}  // namespace ros

namespace planning_ros_msgs {
struct HeaderLite {
  std::string frame_id;
  ros::Time stamp;
};
struct Vector3 {
  double x = 0, y = 0, z = 0;
};
}  // namespace planning_ros_msgs
#endif
