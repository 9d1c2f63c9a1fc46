/** @file LambdaSeg.h  (mplx shim of the generated message, planning_ros_msgs/msg/LambdaSeg.msg: dT, ti, tf, ca) */
#ifndef MPLX_SHIM_LAMBDA_SEG_MSG_H
#define MPLX_SHIM_LAMBDA_SEG_MSG_H
#include <vector>
namespace planning_ros_msgs {
struct LambdaSeg {
  double dT = 0, ti = 0, tf = 0;
  std::vector<double> ca;
};
}  // namespace planning_ros_msgs
#endif
