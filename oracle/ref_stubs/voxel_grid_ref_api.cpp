// C entry points around the reference's own VoxelGrid (compiled from
// /root/reference/planning_ros_utils/src/mapping_utils/voxel_grid.cpp where it lies), so the restatement
// in oracle/mpl_oracle.c can be checked against the real thing.  TEST INFRASTRUCTURE ONLY.
#include <planning_ros_utils/voxel_grid.h>

#include <cstdint>
#include <cstring>

extern "C" {
void *ref_grid_create(const double origin[3], const double dim[3], float res) {
  return new VoxelGrid(Vec3f(origin[0], origin[1], origin[2]), Vec3f(dim[0], dim[1], dim[2]), res);
}
void ref_grid_destroy(void *g) { delete (VoxelGrid *)g; }
int ref_grid_allocate(void *g, const double dim[3], const double ori[3]) {
  return ((VoxelGrid *)g)->allocate(Vec3f(dim[0], dim[1], dim[2]), Vec3f(ori[0], ori[1], ori[2])) ? 1 : 0;
}
void ref_grid_clear(void *g) { ((VoxelGrid *)g)->clear(); }
static vec_Vec3f to_vec(int n, const double *pts) {
  vec_Vec3f v((size_t)n);
  for (int i = 0; i < n; i++) v[(size_t)i] = Vec3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  return v;
}
void ref_grid_add_cloud(void *g, int n, const double *pts) { ((VoxelGrid *)g)->addCloud(to_vec(n, pts)); }
int ref_grid_add_cloud_ns(void *g, int n, const double *pts, int n_ns, const int32_t *ns, int32_t *new_obs, int cap) {
  vec_Vec3i nsv((size_t)n_ns);
  for (int i = 0; i < n_ns; i++) nsv[(size_t)i] = Vec3i(ns[3 * i], ns[3 * i + 1], ns[3 * i + 2]);
  vec_Vec3i out = ((VoxelGrid *)g)->addCloud(to_vec(n, pts), nsv);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++)
    for (int k = 0; k < 3; k++) new_obs[3 * i + k] = out[i](k);
  return (int)out.size();
}
void ref_grid_decay(void *g) { ((VoxelGrid *)g)->decay(); }
void ref_grid_fill_column(void *g, int nx, int ny) { ((VoxelGrid *)g)->fill(nx, ny); }
void ref_grid_fill_cell(void *g, int nx, int ny, int nz) { ((VoxelGrid *)g)->fill(nx, ny, nz); }
void ref_grid_clear_column(void *g, int nx, int ny) { ((VoxelGrid *)g)->clear(nx, ny); }
// VoxelMap fields: dims / origin / resolution and data (x fastest); data may be NULL to query the size
uint64_t ref_grid_get_map(void *g, int inflated, int32_t dim[3], double origin[3], float *res, int8_t *data) {
  planning_ros_msgs::VoxelMap m = inflated ? ((VoxelGrid *)g)->getInflatedMap() : ((VoxelGrid *)g)->getMap();
  dim[0] = (int32_t)m.dim.x; dim[1] = (int32_t)m.dim.y; dim[2] = (int32_t)m.dim.z;
  origin[0] = m.origin.x; origin[1] = m.origin.y; origin[2] = m.origin.z;
  *res = m.resolution;
  if (data) memcpy(data, m.data.data(), m.data.size());
  return m.data.size();
}
uint64_t ref_grid_get_cloud(void *g, double *pts, uint64_t cap) {
  vec_Vec3f c = ((VoxelGrid *)g)->getCloud();
  for (size_t i = 0; i < c.size() && i < cap; i++)
    for (int k = 0; k < 3; k++) pts[3 * i + k] = c[i](k);
  return c.size();
}
}
