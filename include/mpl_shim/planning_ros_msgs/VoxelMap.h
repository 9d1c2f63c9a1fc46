/**
 * @file VoxelMap.h  (mplx shim of the generated <planning_ros_msgs/VoxelMap.h>)
 * The fields of planning_ros_msgs/msg/VoxelMap.msg:1-12 the in-tree code touches (header is carried as a
 * frame id only; there is no ROS here): resolution float32, origin / dim as Point, data int8[] x fastest.
 */
#ifndef MPLX_SHIM_VOXELMAP_MSG_H
#define MPLX_SHIM_VOXELMAP_MSG_H
#include <planning_ros_msgs/Header.h>

#include <string>
#include <vector>

namespace planning_ros_msgs {
struct Point3 {
  double x = 0, y = 0, z = 0;
};
struct VoxelMap {
  HeaderLite header;
  float resolution = 0;
  Point3 origin, dim;
  std::vector<signed char> data;
};
}  // namespace planning_ros_msgs
#endif
