/**
 * @file map_util.h  (mplx shim of <mpl_collision/map_util.h>)
 * MPL::MapUtil<Dim> with the grid resident in HBM (libmplx.so).  Methods and meanings as used in-tree
 * (SURVEY.md Appendix A.2): setMap, getOrigin/getDim/getRes/getMap, freeUnknown, floatToInt,
 * isFree/isOccupied/isOutside (points and cells), dilate, rayTrace, getCloud/getFreeCloud/getUnknownCloud.  setMap copies the grid to the device; callers that keep editing
 * their own vector (map_replanner_node.cpp:188,226) call setMap again before the next plan().
 * Dim == 2 (OccMapUtil) is the 3-D path with one voxel layer whose centre plane is z = 0.
 */
#ifndef MPLX_SHIM_MAP_UTIL_H
#define MPLX_SHIM_MAP_UTIL_H
#include <mpl_basis/data_type.h>
#include <mplx.h>

#include <cmath>
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

  // ---- integer-cell queries (map_replanner_node.cpp:180,217)
  bool isFree(const Veci<Dim> &pn) { return cell(pn) == 0; }
  bool isOccupied(const Veci<Dim> &pn) { return cell(pn) == 1; }
  bool isUnknown(const Veci<Dim> &pn) { return cell(pn) == 2; }
  bool isOutside(const Veci<Dim> &pn) { return cell(pn) == 3; }
  /// states of many cells in one launch (0 free, 1 occupied, 2 unknown, 3 outside) -- not in the
  /// reference API: the loops over rayTrace() results at map_replanner_node.cpp:177-184 become one call
  std::vector<int8_t> cellStates(const vec_Veci<Dim> &pns) {
    std::vector<int32_t> c(3 * pns.size(), 0);
    for (size_t k = 0; k < pns.size(); k++)
      for (int i = 0; i < Dim; i++) c[3 * k + i] = pns[k](i);
    std::vector<int8_t> st(pns.size());
    check(mplx_map_cells(ctx_, (int)pns.size(), c.data(), st.data()));
    return st;
  }
  /// map_planner_node.cpp:75-85
  void dilate(const vec_Veci<Dim> &dilate_neighbor) {
    std::vector<int32_t> off(3 * dilate_neighbor.size(), 0);
    for (size_t k = 0; k < dilate_neighbor.size(); k++)
      for (int i = 0; i < Dim; i++) off[3 * k + i] = dilate_neighbor[k](i);
    check(mplx_map_dilate(ctx_, (int)dilate_neighbor.size(), off.data()));
  }
  /// map_replanner_node.cpp:340 `dilate(0.2, 0.1)`: disc of radius r in xy, +-h in z  [UNVERIFIED: the
  /// implementation is in the un-vendored library; this follows the node's own loop at map_planner_node.cpp:75-83]
  void dilate(decimal_t r, decimal_t h) {
    const int rn = (int)std::ceil(r / res_), hn = Dim == 3 ? (int)std::ceil(h / res_) : 0;
    vec_Veci<Dim> ns;
    for (int nx = -rn; nx <= rn; nx++)
      for (int ny = -rn; ny <= rn; ny++) {
        if (std::hypot(nx, ny) > rn) continue;
        for (int nz = -hn; nz <= hn; nz++) {
          if (nx == 0 && ny == 0 && nz == 0) continue;
          Veci<Dim> n;
          n(0) = nx; n(1) = ny;
          if (Dim == 3) n(Dim - 1) = nz;
          ns.push_back(n);
        }
      }
    dilate(ns);
  }
  /// map_replanner_node.cpp:177,208
  vec_Veci<Dim> rayTrace(const Vecf<Dim> &pt1, const Vecf<Dim> &pt2) {
    double a[3] = {pt1(0), pt1(1), Dim == 3 ? pt1(Dim - 1) : 0.0}, b[3] = {pt2(0), pt2(1), Dim == 3 ? pt2(Dim - 1) : 0.0};
    int n = 0;
    check(mplx_map_raytrace(ctx_, a, b, nullptr, 0, &n));
    std::vector<int32_t> c(3 * (size_t)(n > 0 ? n : 1));
    check(mplx_map_raytrace(ctx_, a, b, c.data(), n, &n));
    vec_Veci<Dim> out((size_t)n);
    for (int k = 0; k < n; k++)
      for (int i = 0; i < Dim; i++) out[(size_t)k](i) = c[3 * (size_t)k + i];
    return out;
  }
  /// map_display.cpp:244,256,266
  vec_Vecf<Dim> getCloud() { return cloud(0); }
  vec_Vecf<Dim> getFreeCloud() { return cloud(1); }
  vec_Vecf<Dim> getUnknownCloud() { return cloud(2); }

  /// re-read origin / dim / res from the device after the grid was replaced there (VoxelGrid::setMapUtil)
  void syncInfo() {
    int32_t d[3];
    double o[3], r;
    check(mplx_map_info(ctx(), d, o, &r));
    for (int i = 0; i < Dim; i++) { dim_(i) = d[i]; origin_d_(i) = o[i]; }
    res_ = r;
  }
  /// the device context planners attach to (not part of the reference API)
  mplx_ctx *ctx() { ensure_ctx(); return ctx_; }
  bool has_ctx() const { return ctx_ != nullptr; }  // (a destructor must not create one)

 private:
  void ensure_ctx() {  // constructors stay free of HIP work: global planner objects are legal (map_replanner_node.cpp:14-15)
    if (!ctx_ && mplx_ctx_create(0, &ctx_) != MPLX_OK) throw std::runtime_error(std::string("mplx: ") + mplx_last_error(nullptr));
  }
  void check(int rc) { if (rc != MPLX_OK) throw std::runtime_error(std::string("mplx: ") + mplx_last_error(ctx_)); }
  int cell(const Veci<Dim> &pn) {
    int32_t c[3] = {pn(0), pn(1), Dim == 3 ? pn(Dim - 1) : 0};
    int8_t s;
    check(mplx_map_cells(ctx_, 1, c, &s));
    return s;
  }
  vec_Vecf<Dim> cloud(int which) {
    uint64_t n = 0;
    check(mplx_map_cloud(ctx_, which, nullptr, 0, &n));
    std::vector<double> p(3 * (size_t)(n > 0 ? n : 1));
    check(mplx_map_cloud(ctx_, which, p.data(), n, &n));
    vec_Vecf<Dim> out((size_t)n);
    for (uint64_t k = 0; k < n; k++)
      for (int i = 0; i < Dim; i++) out[(size_t)k](i) = p[3 * (size_t)k + i];
    return out;
  }
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
