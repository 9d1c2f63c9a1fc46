/**
 * @file voxel_grid.h  (mplx shim of <planning_ros_utils/voxel_grid.h>)
 * The reference's mapper class with its grids resident in HBM (libmplx.so, mplx_grid_*).  Same public
 * methods as planning_ros_utils/include/planning_ros_utils/voxel_grid.h:9-49; getLocalCloud is not
 * provided.  `setMapUtil` is an addition: getMap() handed to a MapUtil device to device, i.e.
 * `setMap(map_util, voxel_mapper_->getMap())` of map_replanner_node.cpp:186-188 without the host copy.
 */
#ifndef MPLX_SHIM_VOXEL_GRID_H
#define MPLX_SHIM_VOXEL_GRID_H
#include <mpl_basis/data_type.h>
#include <mpl_collision/map_util.h>
#include <mplx.h>
#include <planning_ros_msgs/VoxelMap.h>

#include <stdexcept>
#include <string>

class VoxelGrid {
 public:
  VoxelGrid(Vec3f origin, Vec3f dim, float res) {
    double o[3] = {origin(0), origin(1), origin(2)}, d[3] = {dim(0), dim(1), dim(2)};
    if (mplx_grid_create(0, o, d, res, &g_) != MPLX_OK) throw std::runtime_error(std::string("mplx: ") + mplx_grid_last_error(nullptr));
  }
  ~VoxelGrid() { mplx_grid_destroy(g_); }
  VoxelGrid(const VoxelGrid &) = delete;
  VoxelGrid &operator=(const VoxelGrid &) = delete;

  void clear() { check(mplx_grid_clear(g_)); }
  void clear(int nx, int ny) { check(mplx_grid_clear_column(g_, nx, ny)); }
  void fill(int nx, int ny) { check(mplx_grid_fill_column(g_, nx, ny)); }
  void fill(int nx, int ny, int nz) { check(mplx_grid_fill_cell(g_, nx, ny, nz)); }
  void decay() { check(mplx_grid_decay(g_)); }
  bool allocate(const Vec3f &new_dim_d, const Vec3f &new_ori_d) {
    double d[3] = {new_dim_d(0), new_dim_d(1), new_dim_d(2)}, o[3] = {new_ori_d(0), new_ori_d(1), new_ori_d(2)};
    int changed = 0;
    check(mplx_grid_allocate(g_, d, o, &changed));
    return changed != 0;
  }
  void addCloud(const vec_Vec3f &pts) {
    std::vector<double> p = flat(pts);
    check(mplx_grid_add_cloud(g_, (int)pts.size(), p.data()));
  }
  vec_Vec3i addCloud(const vec_Vec3f &pts, const vec_Vec3i &ns) {
    std::vector<double> p = flat(pts);
    std::vector<int32_t> n(3 * ns.size());
    for (size_t k = 0; k < ns.size(); k++)
      for (int i = 0; i < 3; i++) n[3 * k + i] = ns[k](i);
    const size_t cap = pts.size() * ns.size();
    std::vector<int32_t> out(3 * (cap ? cap : 1));
    int n_new = 0;
    check(mplx_grid_add_cloud_inflate(g_, (int)pts.size(), p.data(), (int)ns.size(), n.data(), out.data(), (int)cap, &n_new));
    vec_Vec3i new_obs((size_t)n_new);
    for (int k = 0; k < n_new; k++) new_obs[(size_t)k] = Vec3i(out[3 * (size_t)k], out[3 * (size_t)k + 1], out[3 * (size_t)k + 2]);
    return new_obs;
  }
  vec_Vec3f getCloud() {
    uint64_t n = 0;
    check(mplx_grid_get_cloud(g_, nullptr, 0, &n));
    std::vector<double> p(3 * (size_t)(n ? n : 1));
    check(mplx_grid_get_cloud(g_, p.data(), n, &n));
    vec_Vec3f pts((size_t)n);
    for (uint64_t k = 0; k < n; k++) pts[(size_t)k] = Vec3f(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
    return pts;
  }
  planning_ros_msgs::VoxelMap getMap() { return map(0); }
  planning_ros_msgs::VoxelMap getInflatedMap() { return map(1); }
  /// getMap() (or getInflatedMap()) into a MapUtil without leaving the device
  void setMapUtil(MPL::VoxelMapUtil &map_util, bool inflated = false) {
    check(mplx_grid_to_map(g_, inflated ? 1 : 0, map_util.ctx()));
    map_util.syncInfo();
  }

 private:
  static std::vector<double> flat(const vec_Vec3f &pts) {
    std::vector<double> p(3 * pts.size());
    for (size_t k = 0; k < pts.size(); k++)
      for (int i = 0; i < 3; i++) p[3 * k + i] = pts[k](i);
    return p;
  }
  planning_ros_msgs::VoxelMap map(int inflated) {
    planning_ros_msgs::VoxelMap m;
    int32_t dim[3];
    double ori[3];
    check(mplx_grid_info(g_, dim, ori, &m.resolution));
    m.origin.x = ori[0]; m.origin.y = ori[1]; m.origin.z = ori[2];
    m.dim.x = dim[0]; m.dim.y = dim[1]; m.dim.z = dim[2];
    m.data.resize((size_t)dim[0] * dim[1] * dim[2]);
    check(mplx_grid_get_map(g_, inflated, (int8_t *)m.data.data()));
    return m;
  }
  void check(int rc) { if (rc != MPLX_OK) throw std::runtime_error(std::string("mplx: ") + mplx_grid_last_error(g_)); }
  mplx_grid *g_ = nullptr;
};
#endif
