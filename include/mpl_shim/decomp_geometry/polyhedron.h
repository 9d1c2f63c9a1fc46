/**
 * @file decomp_geometry/polyhedron.h  (mplx stand-in for DecompUtil's header of that name)
 *
 * DecompUtil (github.com/sikang/DecompROS, .gitmodules:4-6) is an un-vendored submodule of the reference; the
 * in-tree PolyMap planner (mpl_external_planner/poly_map_planner/) uses exactly this much of it:
 *   Hyperplane<Dim>(p, n), members p_, n_             poly_map_util.h:44-47, primitive_geometry_utils.h:11-12
 *   Polyhedron<Dim>: add(), hyperplanes(), vs_, inside(pt)   simple_obstacle.h:21-29, poly_map_util.h:73
 * [UNVERIFIED recollection] inside(pt) is non-exclusive: every hyperplane has n . (pt - p) <= epsilon_ (1e-10).
 */
#ifndef MPLX_SHIM_DECOMP_POLYHEDRON_H
#define MPLX_SHIM_DECOMP_POLYHEDRON_H
#include <decomp_basis/data_type.h>

template <int Dim>
struct Hyperplane {
  Hyperplane() {}
  Hyperplane(const Vecf<Dim> &p, const Vecf<Dim> &n) : p_(p), n_(n) {}
  /// signed distance of a point (positive on the side the normal points to)
  decimal_t signed_dist(const Vecf<Dim> &pt) const { return n_.dot(pt - p_); }
  decimal_t dist(const Vecf<Dim> &pt) const { return std::abs(signed_dist(pt)); }
  Vecf<Dim> p_;  ///< point on the plane
  Vecf<Dim> n_;  ///< outward normal
};
typedef Hyperplane<2> Hyperplane2D;
typedef Hyperplane<3> Hyperplane3D;

template <int Dim>
struct Polyhedron {
  Polyhedron() {}
  Polyhedron(const vec_E<Hyperplane<Dim>> &vs) : vs_(vs) {}
  void add(const Hyperplane<Dim> &v) { vs_.push_back(v); }
  bool inside(const Vecf<Dim> &pt) const {
    for (const auto &v : vs_)
      if (v.signed_dist(pt) > epsilon_) return false;
    return true;
  }
  vec_E<Hyperplane<Dim>> hyperplanes() const { return vs_; }
  vec_E<Hyperplane<Dim>> vs_;
};
typedef Polyhedron<2> Polyhedron2D;
typedef Polyhedron<3> Polyhedron3D;
#endif
