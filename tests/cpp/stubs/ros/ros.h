// Stand-in for <ros/ros.h>, test infrastructure only: the reference's mpl_test_node/src/robot.hpp includes it for
// ros::Time::now() and the toSec() of a difference of two of them (robot.hpp:109,127 -- a timing printout), nothing else.
#pragma once
#include <chrono>
#include <memory>
#include <string>
#include <vector>
namespace ros {
struct Duration {
  double s = 0;
  double toSec() const { return s; }
};
struct Time {
  double s = 0;
  static Time now() { return Time{std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()}; }
  Duration operator-(const Time &o) const { return Duration{s - o.s}; }
};
}  // namespace ros
