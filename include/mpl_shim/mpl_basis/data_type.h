/**
 * @file data_type.h  (mplx shim of <mpl_basis/data_type.h>)
 *
 * The typedefs the reference's in-tree code uses (decimal_t, Vecf/Veci,
 * Vec2f..Vec6f, VecDf, vec_E, vec_Vecf, vec_Vec3i, ANSI_COLOR_*).  With MPLX_USE_EIGEN they are the
 * upstream Eigen typedefs; without Eigen (this build container has none) a small fixed-size vector
 * with the members the call sites touch: operator()(i), <<-free construction, +, -, scalar *, dot,
 * norm, lpNorm<Infinity>, Zero(), Constant(), cast.
 */
#ifndef MPLX_SHIM_DATA_TYPE_H
#define MPLX_SHIM_DATA_TYPE_H
#include <cmath>
#include <cstdio>
#include <limits>
#include <memory>
#include <vector>

#define ANSI_COLOR_RED "\x1b[1;31m"
#define ANSI_COLOR_GREEN "\x1b[1;32m"
#define ANSI_COLOR_YELLOW "\x1b[1;33m"
#define ANSI_COLOR_BLUE "\x1b[1;34m"
#define ANSI_COLOR_MAGENTA "\x1b[1;35m"
#define ANSI_COLOR_CYAN "\x1b[1;36m"
#define ANSI_COLOR_RESET "\x1b[0m"

typedef double decimal_t;  // provable in-tree: nh.param("origin_x", origin(0), 0.0) (multi_robot_node.cpp:39-42)

#ifdef MPLX_USE_EIGEN
#include <Eigen/Geometry>
#include <Eigen/StdVector>
template <typename T> using vec_E = std::vector<T, Eigen::aligned_allocator<T>>;
template <int N> using Vecf = Eigen::Matrix<decimal_t, N, 1>;
template <int N> using Veci = Eigen::Matrix<int, N, 1>;
typedef Eigen::Matrix<decimal_t, Eigen::Dynamic, 1> VecDf;
#else
namespace mplx_shim {
template <typename T, int N>
struct Vec {
  T v[N];
  Vec() { for (int i = 0; i < N; i++) v[i] = T(0); }
  template <typename... A, typename = typename std::enable_if<sizeof...(A) == N && (N > 1)>::type>
  Vec(A... a) : v{static_cast<T>(a)...} {}
  T &operator()(int i) { return v[i]; }
  const T &operator()(int i) const { return v[i]; }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  T *data() { return v; }
  const T *data() const { return v; }
  static constexpr int size() { return N; }
  static Vec Zero() { return Vec(); }
  static Vec Ones() { return Constant(T(1)); }
  static Vec Constant(T c) { Vec r; for (int i = 0; i < N; i++) r.v[i] = c; return r; }
  static Vec Unit(int k) { Vec r; r.v[k] = T(1); return r; }
  static Vec UnitX() { return Unit(0); }
  static Vec UnitY() { return Unit(1); }
  static Vec UnitZ() { return Unit(2); }
  Vec operator-() const { Vec r; for (int i = 0; i < N; i++) r.v[i] = -v[i]; return r; }
  Vec &operator+=(const Vec &o) { for (int i = 0; i < N; i++) v[i] += o.v[i]; return *this; }
  Vec &operator-=(const Vec &o) { for (int i = 0; i < N; i++) v[i] -= o.v[i]; return *this; }
  Vec &operator*=(T s) { for (int i = 0; i < N; i++) v[i] *= s; return *this; }
  Vec &operator/=(T s) { for (int i = 0; i < N; i++) v[i] /= s; return *this; }
  bool operator!=(const Vec &o) const { return !(*this == o); }
  const Vec &transpose() const { return *this; }
  template <int P> T lpNorm() const { return lpNormInf(); }  // only lpNorm<Eigen::Infinity> is used in-tree
  Vec operator+(const Vec &o) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] + o.v[i]; return r; }
  Vec operator-(const Vec &o) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] - o.v[i]; return r; }
  Vec operator*(T s) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] * s; return r; }
  Vec operator/(T s) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] / s; return r; }
  bool operator==(const Vec &o) const { for (int i = 0; i < N; i++) if (v[i] != o.v[i]) return false; return true; }
  T dot(const Vec &o) const { T s = T(0); for (int i = 0; i < N; i++) s += v[i] * o.v[i]; return s; }
  T norm() const { return std::sqrt(dot(*this)); }
  T prod() const { T s = T(1); for (int i = 0; i < N; i++) s *= v[i]; return s; }
  T lpNormInf() const { T m = T(0); for (int i = 0; i < N; i++) m = std::fabs(v[i]) > m ? std::fabs(v[i]) : m; return m; }
  template <typename U> Vec<U, N> cast() const { Vec<U, N> r; for (int i = 0; i < N; i++) r.v[i] = static_cast<U>(v[i]); return r; }
  /// Eigen's comma initialiser: `vec << dx, dy, 0, dyaw;` (map_planner_node.cpp:125-126,135-136)
  struct CommaInit {
    Vec &m; int k;
    CommaInit &operator,(T x) { if (k < N) m.v[k++] = x; return *this; }
  };
  CommaInit operator<<(T x) { v[0] = x; return CommaInit{*this, 1}; }
};
template <typename T, int N>
Vec<T, N> operator*(T s, const Vec<T, N> &a) { return a * s; }
/// dynamic-size vector (control inputs: vec_E<VecDf> U; U.push_back(Vec3f(dx, dy, dz)), map_planner_node.cpp:108-139)
struct VecD {
  std::vector<decimal_t> v;
  VecD() {}
  explicit VecD(int n) : v(n, 0.0) {}
  template <int N> VecD(const Vec<decimal_t, N> &o) : v(o.v, o.v + N) {}
  decimal_t &operator()(int i) { return v[i]; }
  const decimal_t &operator()(int i) const { return v[i]; }
  int size() const { return (int)v.size(); }
  int rows() const { return (int)v.size(); }
  decimal_t lpNormInf() const { decimal_t m = 0; for (double e : v) m = std::fabs(e) > m ? std::fabs(e) : m; return m; }
  template <int P> decimal_t lpNorm() const { return lpNormInf(); }  // only lpNorm<Eigen::Infinity> is used in-tree (robot_team.hpp:158)
};
}  // namespace mplx_shim
template <typename T> using vec_E = std::vector<T>;
template <int N> using Vecf = mplx_shim::Vec<decimal_t, N>;
template <int N> using Veci = mplx_shim::Vec<int, N>;
typedef mplx_shim::VecD VecDf;
#endif

template <int N> using vec_Vecf = vec_E<Vecf<N>>;
template <int N> using vec_Veci = vec_E<Veci<N>>;
typedef Vecf<2> Vec2f;
typedef Veci<2> Vec2i;
typedef Vecf<3> Vec3f;
typedef Veci<3> Vec3i;
typedef Vecf<4> Vec4f;
typedef Vecf<6> Vec6f;
typedef vec_E<Vec2f> vec_Vec2f;
typedef vec_E<Vec2i> vec_Vec2i;
typedef vec_E<Vec3f> vec_Vec3f;
typedef vec_E<Vec3i> vec_Vec3i;
#ifndef MPLX_USE_EIGEN
namespace Eigen { enum { Infinity = -1 }; }  // so that `lpNorm<Eigen::Infinity>()` reads the same
/// 3x3 matrix placeholder (Mat3f only appears in the ROS display glue, which is out of scope)
struct Mat3f { decimal_t m[3][3]; };
#endif

#endif
