/**
 * @file map_util.h  (mplx shim of <mpl_collision/map_util.h>)
 * MPL::MapUtil<Dim> with the grid resident in HBM (libmplx.so).  Methods and meanings as used in-tree
 * (SURVEY.md Appendix A.2): setMap, getOrigin/getDim/getRes/getMap, freeUnknown, floatToInt,
 * isFree/isOccupied/isOutside.  setMap copies the grid to the device; callers that keep editing
 * their own vector (map_replanner_node.cpp:188,226) call setMap again before the next plan().
 * Dim == 2 (OccMapUtil) is the 3-D path with one voxel layer whose centre plane is z = 0.
 */
#ifndef MPLX_SHIM_MAP_UTIL_H
#define MPLX_SHIM_MAP_UTIL_H
#include <mpl_basis/data_type.h>
#include <mplx.h>

#include <stdexcept>
#include <string>

namespace MPL {

typedef std::vector<signed char> Tmap;

template <int Dim>
class MapUtil {
 public:
  MapUtil() {}
  ~MapUtil() { if (ctx_) mplx_ctx_destroy(ctx_); }
  MapUtil(const MapUtil &) = delete;
  MapUtil &operator=(const MapUtil &) = delete;

  void setMap(const Vecf<Dim> &ori, const Veci<Dim> &dim, const Tmap &map, decimal_t res) {
    ensure_ctx();
    int32_t d[3] = {dim(0), dim(1), Dim == 3 ? dim(Dim - 1) : 1};
    double o[3] = {ori(0), ori(1), Dim == 3 ? ori(Dim - 1) : -0.5 * res};
    check(mplx_map_set(ctx_, map.data(), d, o, res));
    origin_d_ = ori;
    dim_ = dim;
    res_ = res;
  }
  Vecf<Dim> getOrigin() const { return origin_d_; }
  Veci<Dim> getDim() const { return dim_; }
  decimal_t getRes() const { return res_; }
  Tmap getMap() {
    Tmap m((size_t)dim_.prod());
    check(mplx_map_get(ctx_, m.data()));
    return m;
  }
  void freeUnknown() { check(mplx_map_free_unknown(ctx_)); }
  Veci<Dim> floatToInt(const Vecf<Dim> &pt) {
    int8_t s;
    return query(pt, s);
  }
  bool isFree(const Vecf<Dim> &pt) { int8_t s; query(pt, s); return s == 0; }
  bool isOccupied(const Vecf<Dim> &pt) { int8_t s; query(pt, s); return s == 1; }
  bool isUnknown(const Vecf<Dim> &pt) { int8_t s; query(pt, s); return s == 2; }
  bool isOutside(const Vecf<Dim> &pt) { int8_t s; query(pt, s); return s == 3; }

  /// the device context planners attach to (not part of the reference API)
  mplx_ctx *ctx() { ensure_ctx(); return ctx_; }

 private:
  void ensure_ctx() {  // constructors stay free of HIP work: global planner objects are legal (map_replanner_node.cpp:14-15)
    if (!ctx_ && mplx_ctx_create(0, &ctx_) != MPLX_OK) throw std::runtime_error(std::string("mplx: ") + mplx_last_error(nullptr));
  }
  void check(int rc) { if (rc != MPLX_OK) throw std::runtime_error(std::string("mplx: ") + mplx_last_error(ctx_)); }
  Veci<Dim> query(const Vecf<Dim> &pt, int8_t &state) {
    double p[3] = {pt(0), pt(1), Dim == 3 ? pt(Dim - 1) : 0.0};
    int32_t c[3];
    check(mplx_map_query(ctx_, 1, p, c, &state));
    Veci<Dim> r;
    for (int i = 0; i < Dim; i++) r(i) = c[i];
    return r;
  }
  mplx_ctx *ctx_ = nullptr;
  Vecf<Dim> origin_d_;
  Veci<Dim> dim_;
  decimal_t res_ = 0;
};

typedef MapUtil<2> OccMapUtil;
typedef MapUtil<3> VoxelMapUtil;

}  // namespace MPL
#endif
