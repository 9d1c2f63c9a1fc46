// Stand-in for <boost/multi_array.hpp> (absent from this image) with exactly the members the reference's
// planning_ros_utils/src/mapping_utils/voxel_grid.cpp uses: multi_array<T,3> built from
// boost::extents[a][b][c], row-major storage (last index fastest, like boost's default c_storage_order),
// data(), num_elements(), resize(extents), copy assignment, operator[][][].
// TEST INFRASTRUCTURE ONLY (oracle/_ref build recipe); not part of the product.
#ifndef MPLX_REF_STUB_MULTI_ARRAY_HPP
#define MPLX_REF_STUB_MULTI_ARRAY_HPP
#include <cstddef>
#include <vector>
namespace boost {
struct extent_gen3 { long e[3]; };
struct extent_gen2 { long e[2]; extent_gen3 operator[](long c) const { return extent_gen3{{e[0], e[1], c}}; } };
struct extent_gen1 { long e[1]; extent_gen2 operator[](long b) const { return extent_gen2{{e[0], b}}; } };
struct extent_gen0 { extent_gen1 operator[](long a) const { return extent_gen1{{a}}; } };
static const extent_gen0 extents = extent_gen0();
template <typename T, int N>
class multi_array;
template <typename T>
class multi_array<T, 3> {
 public:
  multi_array() { d_[0] = d_[1] = d_[2] = 0; }
  explicit multi_array(const extent_gen3 &x) { alloc(x); }
  void resize(const extent_gen3 &x) {  // boost preserves the overlap; voxel_grid.cpp overwrites the array right after
    multi_array old(*this);
    alloc(x);
    for (long a = 0; a < d_[0] && a < old.d_[0]; a++)
      for (long b = 0; b < d_[1] && b < old.d_[1]; b++)
        for (long c = 0; c < d_[2] && c < old.d_[2]; c++) v_[(a * d_[1] + b) * d_[2] + c] = old.v_[(a * old.d_[1] + b) * old.d_[2] + c];
  }
  T *data() { return v_.data(); }
  const T *data() const { return v_.data(); }
  size_t num_elements() const { return v_.size(); }
  struct row2 {
    T *p;
    T &operator[](long c) const { return p[c]; }
  };
  struct row1 {
    T *p; long d1, d2;
    row2 operator[](long b) const { return row2{p + b * d2}; }
  };
  row1 operator[](long a) { return row1{v_.data() + a * d_[1] * d_[2], d_[1], d_[2]}; }

 private:
  void alloc(const extent_gen3 &x) {
    for (int i = 0; i < 3; i++) d_[i] = x.e[i] > 0 ? x.e[i] : 0;
    v_.assign((size_t)d_[0] * d_[1] * d_[2], T());
  }
  long d_[3];
  std::vector<T> v_;
};
}  // namespace boost
#endif
