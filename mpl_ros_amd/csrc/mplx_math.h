// mplx_math.h -- device/host f64 math of the motion-primitive hot path (gfx950 HIP).
//
// Replaces, for the voxel-map A* path, the un-vendored MPL v1.2 basis functions that the reference
// calls through `Primitive<Dim>(curr, U[i], dt)`, `pr.evaluate(dt)`, `validate_primitive(...)`,
// `pr.max_vel(k)`, `pr.J(control)` and the Waypoint hash
//   (call sites: mpl_external_planner/include/mpl_external_planner/poly_map_planner/env_poly_map.h:54-62,
//    .../ellipsoid_planner/env_cloud.h:59-67, .../ellipsoid_planner/ellipsoid_util.h:67-70).
// Polynomial convention (primitive_geometry_utils.h:12-26): p(t) = c0/120 t^5 + c1/24 t^4 + c2/6 t^3
// + c3/2 t^2 + c4 t + c5 per axis.
//
// Everything is f64 and built from + - * / sqrt round ceil fabs only, compiled with
// -ffp-contract=off, so results are bit-identical to a CPU evaluation of the same expressions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MPLX_HD __host__ __device__ __forceinline__

namespace mplx {

// Control bit flags: a state's control kind is the union of its use_pos/vel/acc/jrk bits.
enum : int { CTRL_VEL = 1, CTRL_ACC = 3, CTRL_JRK = 7, CTRL_SNP = 15 };

// Waypoint key quantisation (resolutions of the Waypoint hash).
constexpr double KEY_RES_POS = 0.01;
constexpr double KEY_RES_VEL = 0.1;
constexpr double KEY_RES_ACC = 0.1;
constexpr double KEY_RES_JRK = 0.1;

constexpr int MAX_KEY = 12;   // 3 axes x (pos, vel, acc, jrk)
constexpr int MAX_STATE = 12; // doubles per stored state

struct State {  // pos, vel, acc, jrk per axis
  double p[3], v[3], a[3], j[3];
};

// number of state doubles / key ints for a control kind (VEL 3, ACC 6, JRK 9, SNP 12)
MPLX_HD int state_len(int control) { return control == CTRL_VEL ? 3 : control == CTRL_ACC ? 6 : control == CTRL_JRK ? 9 : 12; }

// ---- per-axis polynomial in "monomial" form: m[k] multiplies t^k (m5 = c0/120 ... m0 = c5).
// The divisions are the ones the convention writes (c0/120, c1/24, ...); they are hoisted per
// primitive, which leaves every product and sum identical.
struct Poly1 {
  double c[6];
};

MPLX_HD void prim_build_axis(int control, double p, double v, double a, double j, double u, double *c) {
  c[0] = 0; c[1] = 0; c[2] = 0; c[3] = 0; c[4] = 0; c[5] = 0;
  switch (control) {
    case CTRL_VEL: c[4] = u; c[5] = p; break;
    case CTRL_ACC: c[3] = u; c[4] = v; c[5] = p; break;
    case CTRL_JRK: c[2] = u; c[3] = a; c[4] = v; c[5] = p; break;
    case CTRL_SNP: c[1] = u; c[2] = j; c[3] = a; c[4] = v; c[5] = p; break;
    default: break;
  }
}

MPLX_HD double pos_at(const double *c, double t) {
  double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  return c[0] / 120 * t5 + c[1] / 24 * t4 + c[2] / 6 * t3 + c[3] / 2 * t * t + c[4] * t + c[5];
}
MPLX_HD double vel_at(const double *c, double t) {
  double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
  return c[0] / 24 * t4 + c[1] / 6 * t3 + c[2] / 2 * t * t + c[3] * t + c[4];
}
MPLX_HD double acc_at(const double *c, double t) {
  double t2 = t * t, t3 = t2 * t;
  return c[0] / 6 * t3 + c[1] / 2 * t * t + c[2] * t + c[3];
}
MPLX_HD double jrk_at(const double *c, double t) { return c[0] / 2 * t * t + c[1] * t + c[2]; }

// position with pre-divided coefficients q = {c0/120, c1/24, c2/6, c3/2, c4, c5}
MPLX_HD double pos_at_q(const double *q, double t) {
  double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  return q[0] * t5 + q[1] * t4 + q[2] * t3 + q[3] * t * t + q[4] * t + q[5];
}

// roots of  c2 x^2 + c1 x + c0 = 0 / c1 x + c0 = 0 in the order the quadratic formula gives them
// ((-c1 - sqrt(D))/(2 c2) first); returns the count.
MPLX_HD int roots_upto_quad(double c2, double c1, double c0, double *r) {
  if (c2 != 0.0) {
    double D = c1 * c1 - 4 * c2 * c0;
    if (D < 0) return 0;
    double s = sqrt(D);
    r[0] = (-c1 - s) / (2 * c2);
    r[1] = (-c1 + s) / (2 * c2);
    return 2;
  }
  if (c1 != 0.0) {
    r[0] = -c0 / c1;
    return 1;
  }
  return 0;
}

// ---- control-specialised evaluators.
// A primitive built from (state, control input) has structurally zero leading coefficients
// (VEL c0..c3, ACC c0..c2, JRK c0..c1, SNP c0).  With t >= 0 finite every such term is +0.0, and
// (+0.0 + x) equals x except that it turns -0.0 into +0.0; "x + 0.0" reproduces exactly that, so
// these return bit-identical results to pos_at/vel_at/acc_at/jrk_at without the dead multiplies
// and divisions (checked bit-for-bit by tests/cpp/test_math_host.cpp).
template <int CONTROL>
MPLX_HD double pos_at_c(const double *c, double t) {
  if constexpr (CONTROL == CTRL_VEL) {
    return (c[4] * t + 0.0) + c[5];
  } else if constexpr (CONTROL == CTRL_ACC) {
    return ((c[3] / 2 * t * t + 0.0) + c[4] * t) + c[5];
  } else if constexpr (CONTROL == CTRL_JRK) {
    double t2 = t * t, t3 = t2 * t;
    return (((c[2] / 6 * t3 + 0.0) + c[3] / 2 * t * t) + c[4] * t) + c[5];
  } else {
    double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
    return ((((c[1] / 24 * t4 + 0.0) + c[2] / 6 * t3) + c[3] / 2 * t * t) + c[4] * t) + c[5];
  }
}
template <int CONTROL>
MPLX_HD double vel_at_c(const double *c, double t) {
  if constexpr (CONTROL == CTRL_VEL) {
    return c[4] + 0.0;
  } else if constexpr (CONTROL == CTRL_ACC) {
    return (c[3] * t + 0.0) + c[4];
  } else if constexpr (CONTROL == CTRL_JRK) {
    return ((c[2] / 2 * t * t + 0.0) + c[3] * t) + c[4];
  } else {
    double t2 = t * t, t3 = t2 * t;
    return (((c[1] / 6 * t3 + 0.0) + c[2] / 2 * t * t) + c[3] * t) + c[4];
  }
}
template <int CONTROL>
MPLX_HD double acc_at_c(const double *c, double t) {
  if constexpr (CONTROL == CTRL_VEL) {
    return 0.0;
  } else if constexpr (CONTROL == CTRL_ACC) {
    return c[3] + 0.0;
  } else if constexpr (CONTROL == CTRL_JRK) {
    return (c[2] * t + 0.0) + c[3];
  } else {
    return ((c[1] / 2 * t * t + 0.0) + c[2] * t) + c[3];
  }
}
template <int CONTROL>
MPLX_HD double jrk_at_c(const double *c, double t) {
  if constexpr (CONTROL == CTRL_VEL || CONTROL == CTRL_ACC) {
    return 0.0;
  } else if constexpr (CONTROL == CTRL_JRK) {
    return c[2] + 0.0;
  } else {
    return (c[1] * t + 0.0) + c[2];
  }
}
// number of non-zero pre-divided coefficients per axis and their packing q' = {c_first/.., ..., c4, c5}
constexpr int nq_c(int control) { return control == CTRL_VEL ? 2 : control == CTRL_ACC ? 3 : control == CTRL_JRK ? 4 : 5; }
// pack the pre-divided non-zero coefficients of one axis (same divisions as pos_at)
template <int CONTROL>
MPLX_HD void pack_q_c(const double *c, double *q) {
  if constexpr (CONTROL == CTRL_VEL) {
    q[0] = c[4]; q[1] = c[5];
  } else if constexpr (CONTROL == CTRL_ACC) {
    q[0] = c[3] / 2; q[1] = c[4]; q[2] = c[5];
  } else if constexpr (CONTROL == CTRL_JRK) {
    q[0] = c[2] / 6; q[1] = c[3] / 2; q[2] = c[4]; q[3] = c[5];
  } else {
    q[0] = c[1] / 24; q[1] = c[2] / 6; q[2] = c[3] / 2; q[3] = c[4]; q[4] = c[5];
  }
}
template <int CONTROL>
MPLX_HD double pos_at_qc(const double *q, double t) {
  if constexpr (CONTROL == CTRL_VEL) {
    return (q[0] * t + 0.0) + q[1];
  } else if constexpr (CONTROL == CTRL_ACC) {
    return ((q[0] * t * t + 0.0) + q[1] * t) + q[2];
  } else if constexpr (CONTROL == CTRL_JRK) {
    double t2 = t * t, t3 = t2 * t;
    return (((q[0] * t3 + 0.0) + q[1] * t * t) + q[2] * t) + q[3];
  } else {
    double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
    return ((((q[0] * t4 + 0.0) + q[1] * t3) + q[2] * t * t) + q[3] * t) + q[4];
  }
}

// Position for the voxel test only: pos_at_qc without the "+ 0.0" terms.  Those only turn a -0.0
// partial sum into +0.0; a zero of either sign gives the same (p - origin) / res - 0.5 and so the
// same cell (p - origin is -origin for both when origin != 0, and +-0 / res - 0.5 = -0.5 otherwise).
template <int CONTROL>
MPLX_HD double pos_at_qc_cell(const double *q, double t) {
  if constexpr (CONTROL == CTRL_VEL) {
    return q[0] * t + q[1];
  } else if constexpr (CONTROL == CTRL_ACC) {
    return (q[0] * t * t + q[1] * t) + q[2];
  } else if constexpr (CONTROL == CTRL_JRK) {
    double t2 = t * t, t3 = t2 * t;
    return ((q[0] * t3 + q[1] * t * t) + q[2] * t) + q[3];
  } else {
    double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
    return (((q[0] * t4 + q[1] * t3) + q[2] * t * t) + q[3] * t) + q[4];
  }
}

// max |d^k p| on [0,T] with the control-specialised evaluators (same scan as max_abs_deriv)
template <int K, int CONTROL>
MPLX_HD double max_abs_deriv_c(const double *c, double T) {
  double r[2];
  int n;
  if (K == 1)
    n = roots_upto_quad(c[1] / 2, c[2], c[3], r);
  else if (K == 2)
    n = roots_upto_quad(c[0] / 2, c[1], c[2], r);
  else
    n = roots_upto_quad(0.0, c[0], c[1], r);
  auto f = [&](double t) { return K == 1 ? vel_at_c<CONTROL>(c, t) : K == 2 ? acc_at_c<CONTROL>(c, t) : jrk_at_c<CONTROL>(c, t); };
  double mx = fmax(fabs(f(0.0)), fabs(f(T)));
  for (int i = 0; i < n; i++) {
    if (r[i] > 0 && r[i] < T) {
      double v = fabs(f(r[i]));
      mx = v > mx ? v : mx;
    } else if (r[i] >= T)
      break;
  }
  return mx;
}
template <int CONTROL>
MPLX_HD bool validate_and_maxv_c(const double c[3][6], double T, double mv, double ma, double mj, double *max_v_out) {
  double vx = max_abs_deriv_c<1, CONTROL>(c[0], T), vy = max_abs_deriv_c<1, CONTROL>(c[1], T), vz = max_abs_deriv_c<1, CONTROL>(c[2], T);
  double max_v = 0;
  if (vx > max_v) max_v = vx;
  if (vy > max_v) max_v = vy;
  if (vz > max_v) max_v = vz;
  *max_v_out = max_v;
  constexpr bool chk_v = CONTROL == CTRL_ACC || CONTROL == CTRL_JRK || CONTROL == CTRL_SNP;
  constexpr bool chk_a = CONTROL == CTRL_JRK || CONTROL == CTRL_SNP;
  constexpr bool chk_j = CONTROL == CTRL_SNP;
  if (chk_v && mv > 0 && (vx > mv || vy > mv || vz > mv)) return false;
  if (chk_a && ma > 0)
    for (int i = 0; i < 3; i++)
      if (max_abs_deriv_c<2, CONTROL>(c[i], T) > ma) return false;
  if (chk_j && mj > 0)
    for (int i = 0; i < 3; i++)
      if (max_abs_deriv_c<3, CONTROL>(c[i], T) > mj) return false;
  return true;
}

// max |d^k p| on [0,T] for k = 1 (vel), 2 (acc), 3 (jrk): end points plus the interior
// stationary points, scanning the roots in formula order and stopping at the first root >= T.
// Control-built primitives have c0 == 0, so the stationary-point polynomial is at most quadratic.
template <int K>
MPLX_HD double max_abs_deriv(const double *c, double T) {
  double r[2];
  int n;
  if (K == 1)
    n = roots_upto_quad(c[1] / 2, c[2], c[3], r);  // a(t) = c1/2 t^2 + c2 t + c3   (c0/6 t^3 == 0)
  else if (K == 2)
    n = roots_upto_quad(c[0] / 2, c[1], c[2], r);  // j(t)
  else
    n = roots_upto_quad(0.0, c[0], c[1], r);       // snap
  auto f = [&](double t) { return K == 1 ? vel_at(c, t) : K == 2 ? acc_at(c, t) : jrk_at(c, t); };
  double mx = fmax(fabs(f(0.0)), fabs(f(T)));
  for (int i = 0; i < n; i++) {
    if (r[i] > 0 && r[i] < T) {
      double v = fabs(f(r[i]));
      mx = v > mx ? v : mx;
    } else if (r[i] >= T)
      break;
  }
  return mx;
}

// validate_primitive: ACC checks vel; JRK vel+acc; SNP vel+acc+jrk; a limit <= 0 disables it.
// Also returns max_v = max over axes of max_vel (needed for the sampling density).
MPLX_HD bool validate_and_maxv(int control, const double c[3][6], double T, double mv, double ma, double mj, double *max_v_out) {
  double vx = max_abs_deriv<1>(c[0], T), vy = max_abs_deriv<1>(c[1], T), vz = max_abs_deriv<1>(c[2], T);
  double max_v = 0;
  if (vx > max_v) max_v = vx;
  if (vy > max_v) max_v = vy;
  if (vz > max_v) max_v = vz;
  *max_v_out = max_v;
  bool chk_v = control == CTRL_ACC || control == CTRL_JRK || control == CTRL_SNP;
  bool chk_a = control == CTRL_JRK || control == CTRL_SNP;
  bool chk_j = control == CTRL_SNP;
  if (chk_v && mv > 0 && (vx > mv || vy > mv || vz > mv)) return false;
  if (chk_a && ma > 0)
    for (int i = 0; i < 3; i++)
      if (max_abs_deriv<2>(c[i], T) > ma) return false;
  if (chk_j && mj > 0)
    for (int i = 0; i < 3; i++)
      if (max_abs_deriv<3>(c[i], T) > mj) return false;
  return true;
}

MPLX_HD double powi(double t, int n) {
  double r = t;
  for (int i = 1; i < n; i++) r = r * t;
  return r;
}

// J(control): integral over [0,T] of the squared (order)-th derivative, summed over the axes.
// Double sum over the derivative's monomial coefficients in ascending (i,j) order.
MPLX_HD double prim_J(int control, const double c[3][6], double T) {
  int k = control == CTRL_VEL ? 1 : control == CTRL_ACC ? 2 : control == CTRL_JRK ? 3 : 4;
  const double fact[6] = {1, 1, 2, 6, 24, 120};
  double total = 0.0;
  for (int ax = 0; ax < 3; ax++) {
    double q[6];
    int nq = 6 - k;
    for (int m = k; m <= 5; m++) q[m - k] = c[ax][5 - m] / fact[m - k];
    double s = 0.0;
    for (int i = 0; i < nq; i++)
      for (int j = 0; j < nq; j++) s += q[i] * q[j] * powi(T, i + j + 1) / (double)(i + j + 1);
    total += s;
  }
  return total;
}

// quantised key of a state; order per axis: pos, vel, acc, jrk (enabled ones)
MPLX_HD int state_key(int control, const State &s, int32_t *key) {
  int n = 0;
  for (int i = 0; i < 3; i++) {
    if (control & 1) key[n++] = (int32_t)round(s.p[i] / KEY_RES_POS);
    if (control & 2) key[n++] = (int32_t)round(s.v[i] / KEY_RES_VEL);
    if (control & 4) key[n++] = (int32_t)round(s.a[i] / KEY_RES_ACC);
    if (control & 8) key[n++] = (int32_t)round(s.j[i] / KEY_RES_JRK);
  }
  return n;
}

constexpr int key_len_c(int control) { return control == CTRL_VEL ? 3 : control == CTRL_ACC ? 6 : control == CTRL_JRK ? 9 : 12; }
// compile-time-control variant: fixed positions, fully unrollable (keeps the key in registers)
template <int CONTROL>
MPLX_HD void state_key_c(const State &s, int32_t *key) {
  constexpr int per = (CONTROL & 1 ? 1 : 0) + (CONTROL & 2 ? 1 : 0) + (CONTROL & 4 ? 1 : 0) + (CONTROL & 8 ? 1 : 0);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int n = i * per;
    if (CONTROL & 1) key[n++] = (int32_t)round(s.p[i] / KEY_RES_POS);
    if (CONTROL & 2) key[n++] = (int32_t)round(s.v[i] / KEY_RES_VEL);
    if (CONTROL & 4) key[n++] = (int32_t)round(s.a[i] / KEY_RES_ACC);
    if (CONTROL & 8) key[n++] = (int32_t)round(s.j[i] / KEY_RES_JRK);
  }
}

// MapUtil::floatToInt: round((pt - origin)/res - 0.5)
MPLX_HD int32_t float_to_cell(double p, double origin, double res) { return (int32_t)round((p - origin) / res - 0.5); }
// d / r from inv = 1.0 / r without the hardware division sequence: q0 = RN(d inv), e = d - q0 r (exact in
// an FMA), q = RN(q0 + e inv).  With inv the correctly rounded reciprocal and q0 a faithful quotient
// this is the correctly rounded quotient (Markstein's theorem), i.e. bit-identical to d / r; the
// host test checks it against `/` on 10^7 lattice and random points per resolution.  3 instructions
// instead of ~12 in the voxel sampling loop, which is bound by f64 issue.
MPLX_HD double div_by_inv(double d, double r, double inv) {
  const double q0 = d * inv;
  const double e = fma(-q0, r, d);
  return fma(e, inv, q0);
}
MPLX_HD int32_t float_to_cell_inv(double p, double origin, double res, double inv_res) {
  return (int32_t)round(div_by_inv(p - origin, res, inv_res) - 0.5);
}

// ------------------------------------------------------------------ yaw (Waypoint::yaw, use_yaw lattices: map_planner_node.cpp:119-139,165,179)
// A yaw-carrying primitive adds a VEL-type channel yaw(t) = yaw0 + u_yaw t (Primitive::pr_yaw); evaluate() normalises the
// angle, the Waypoint key gains round(yaw / 0.1), validate_primitive gains validate_yaw [UNVERIFIED upstream, restated
// in the tests' CPU checker]: at t = 0 and t = T, when the planar velocity is not zero, its direction must lie within
// yaw_max of the yaw direction: v_hat . (cos yaw, sin yaw) >= cos(yaw_max).
// sin / cos are evaluated by ONE fixed sequence of + - * (Cody-Waite reduction to [-pi/4, pi/4], Taylor polynomials in
// Horner form) on host and device alike, so that the comparison above falls the same way on both (libm and the
// device math library may differ in the last bit).  |error| < 2e-16 for |x| <= 8.
constexpr double KEY_RES_YAW = 0.1;
constexpr int CTRL_YAW_BIT = 16;  // Control::*xYAW = base | 0b10000
MPLX_HD void det_sincos(double x, double *sn, double *cs) {
  const double two_over_pi = 0.63661977236758138, pio2_hi = 1.5707963267948966, pio2_lo = 6.123233995736766e-17;
  const double kf = round(x * two_over_pi);
  const double r = (x - kf * pio2_hi) - kf * pio2_lo;
  const double r2 = r * r;
  // sin r = r (1 - r2/6 (1 - r2/20 (1 - r2/42 (1 - r2/72 (1 - r2/110 (1 - r2/156 (1 - r2/210)))))))
  double ps = 1.0 - r2 / 210.0;
  ps = 1.0 - r2 / 156.0 * ps;
  ps = 1.0 - r2 / 110.0 * ps;
  ps = 1.0 - r2 / 72.0 * ps;
  ps = 1.0 - r2 / 42.0 * ps;
  ps = 1.0 - r2 / 20.0 * ps;
  ps = 1.0 - r2 / 6.0 * ps;
  const double s = r * ps;
  // cos r = 1 - r2/2 (1 - r2/12 (1 - r2/30 (1 - r2/56 (1 - r2/90 (1 - r2/132 (1 - r2/182 (1 - r2/240)))))))
  double pc = 1.0 - r2 / 240.0;
  pc = 1.0 - r2 / 182.0 * pc;
  pc = 1.0 - r2 / 132.0 * pc;
  pc = 1.0 - r2 / 90.0 * pc;
  pc = 1.0 - r2 / 56.0 * pc;
  pc = 1.0 - r2 / 30.0 * pc;
  pc = 1.0 - r2 / 12.0 * pc;
  const double c = 1.0 - r2 / 2.0 * pc;
  const int k = (int)kf & 3;
  *sn = k == 0 ? s : k == 1 ? c : k == 2 ? -s : -c;
  *cs = k == 0 ? c : k == 1 ? -s : k == 2 ? -c : s;
}
// normalize_angle [UNVERIFIED upstream]: into [-pi, pi] by steps of 2 pi
MPLX_HD double normalize_yaw(double q) {
  const double pi = 3.141592653589793;
  while (q > pi) q -= 2.0 * pi;
  while (q < -pi) q += 2.0 * pi;
  return q;
}
// one end of validate_yaw: planar velocity (vx, vy) against the yaw direction
MPLX_HD bool yaw_end_ok(double vx, double vy, double yaw, double cos_max) {
  if (vx == 0.0 && vy == 0.0) return true;
  const double n = sqrt(vx * vx + vy * vy);
  double sn, cs;
  det_sincos(yaw, &sn, &cs);
  const double d = vx / n * cs + vy / n * sn;
  return !(d < cos_max);
}

// ------------------------------------------------------------------ polynomial real roots
// Derivative-chain isolation + safeguarded Newton/bisection, basic arithmetic only (deterministic
// on host and device).  Coefficients ascending: a[0] + a[1] x + ... + a[N] x^N.
// Degrees are compile-time so every array is indexed statically and stays in registers; a zero
// leading coefficient drops to the next lower instantiation (the derivative of a polynomial with a
// non-zero leading coefficient keeps a non-zero leading coefficient, so no other trimming exists).
template <int N>
MPLX_HD double poly_eval(const double *a, double x) {
  double r = a[N];
#pragma unroll
  for (int i = N - 1; i >= 0; i--) r = r * x + a[i];
  return r;
}
template <int N>
MPLX_HD void poly_eval2(const double *a, double x, double &f, double &df) {
  double r = a[N], d = 0.0;
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    d = d * x + r;
    r = r * x + a[i];
  }
  f = r;
  df = d;
}
template <int N>
MPLX_HD double poly_refine(const double *a, double x1, double f1, double x2) {
  double xl, xh;
  if (f1 < 0.0) { xl = x1; xh = x2; } else { xl = x2; xh = x1; }
  double x = 0.5 * (x1 + x2);
  double dxold = fabs(x2 - x1), dx = dxold, f, df;
  poly_eval2<N>(a, x, f, df);
  for (int it = 0; it < 200; it++) {
    if ((((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      x = xl + dx;
      if (xl == x) return x;
    } else {
      dxold = dx;
      dx = f / df;
      double tmp = x;
      x = x - dx;
      if (tmp == x) return x;
    }
    if (fabs(dx) <= 4.0e-16 * fabs(x)) return x;
    poly_eval2<N>(a, x, f, df);
    if (f == 0.0) return x;
    if (f < 0.0) xl = x; else xh = x;
  }
  return x;
}
// roots of a degree-N polynomial (a[N] != 0) inside (lo, hi), ascending; roots has room for N
template <int N>
MPLX_HD int poly_roots_in(const double *a, double lo, double hi, double *roots) {
  if constexpr (N == 1) {
    double r = -a[0] / a[1];
    if (r > lo && r < hi) {
      roots[0] = r;
      return 1;
    }
    return 0;
  } else {
    double d[N], crit[N - 1 > 0 ? N - 1 : 1];
#pragma unroll
    for (int i = 1; i <= N; i++) d[i - 1] = a[i] * (double)i;
    const int nc = poly_roots_in<N - 1>(d, lo, hi, crit);
    int nr = 0;
    double x0 = lo, f0 = poly_eval<N>(a, lo);
#pragma unroll
    for (int k = 0; k <= N - 1; k++) {
      if (k <= nc) {
        double x1 = hi;
#pragma unroll
        for (int m = 0; m < N - 1; m++)
          if (m == k && k < nc) x1 = crit[m];
        double f1 = poly_eval<N>(a, x1);
        double r = 0.0;
        bool have = false;
        if (f1 == 0.0) {
          if (k < nc) { r = x1; have = true; }
        } else if (f0 != 0.0 && ((f0 < 0.0) != (f1 < 0.0))) {
          r = poly_refine<N>(a, x0, f0, x1);
          have = true;
        }
        if (have) {
#pragma unroll
          for (int m = 0; m < N; m++)
            if (m == nr) roots[m] = r;
          nr++;
        }
        x0 = x1;
        f0 = f1;
      }
    }
    return nr;
  }
}
// real roots in (lo, +inf), ascending, of a polynomial of degree <= N
template <int N>
MPLX_HD int poly_roots_above(const double *a, double lo, double *roots) {
  if constexpr (N == 0) {
    return 0;
  } else {
    if (a[N] == 0.0) return poly_roots_above<N - 1>(a, lo, roots);
    // Bound on |root|.  Cauchy: 1 + max_i |a_i / a_N| -- division is
    // sign-symmetric and monotone, so the maximum of the rounded quotients is the rounded quotient of the
    // maximum: one division, same bits.  With a zero coefficient below the leading one (both heuristic
    // polynomials): 1 + sqrt(sum_i |a_i| / |a_N|), a far shorter bracket.
    double m;
    bool gap = false;
    if constexpr (N >= 2) gap = a[N - 1] == 0.0;
    if (gap) {
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < N; i++) sum += fabs(a[i]);
      m = sqrt(sum / fabs(a[N]));
    } else {
      double mx = 0.0;
#pragma unroll
      for (int i = 0; i < N; i++) {
        const double q = fabs(a[i]);
        if (q > mx) mx = q;
      }
      m = mx / fabs(a[N]);
    }
    double hi = 1.0 + m;
    if (!(hi > lo)) return 0;
    return poly_roots_in<N>(a, lo, hi, roots);
  }
}

// ------------------------------------------------------------------ heuristic (env_base::cal_heur)
MPLX_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MPLX_HD double linf3(const double *a, const double *b) {
  double m = 0;
  for (int i = 0; i < 3; i++) {
    double d = fabs(a[i] - b[i]);
    if (d > m) m = d;
  }
  return m;
}
MPLX_HD double heur_min6(double a, double c, double d, double e, double f, double g, double t_bar) {
  double co[7] = {g, f, e, d, c, 0.0, a}, ts[6];
  const int n = poly_roots_above<6>(co, t_bar > 0 ? t_bar : 0.0, ts);
  double best = INFINITY;
#pragma unroll
  for (int i = 0; i <= 6; i++) {  // the roots in ascending order, then t_bar itself
    if (i < n || i == 6) {
      double t = i == 6 ? t_bar : ts[i < 6 ? i : 0];
      if (!(t < t_bar)) {
        double cost = a * t - c / t - d / 2 / t / t - e / 3 / t / t / t - f / 4 / t / t / t / t - g / 5 / t / t / t / t / t;
        if (cost < best) best = cost;
      }
    }
  }
  return best;
}
MPLX_HD double heur_min4(double c5, double c3, double c2, double c1, double w, double t_bar) {
  double co[5] = {c1, c2, c3, 0.0, c5}, ts[4];
  const int n = poly_roots_above<4>(co, t_bar > 0 ? t_bar : 0.0, ts);
  double best = INFINITY;
#pragma unroll
  for (int i = 0; i <= 4; i++) {  // the roots in ascending order, then t_bar itself
    if (i < n || i == 4) {
      double t = i == 4 ? t_bar : ts[i < 4 ? i : 0];
      if (!(t < t_bar)) {
        double c = -c1 / 3 / t / t / t - c2 / 2 / t / t - c3 / t + w * t;
        if (c < best) best = c;
      }
    }
  }
  return best;
}

struct HeurParams {
  double w, v_max;
  int heur_ignore_dynamics;
  int goal_control;
  State goal;
  int32_t goal_key[MAX_KEY];
  int goal_nkey;
  int goal_yaw_key;  // (yaw-carrying searches) round(goal.yaw / 0.1) and the goal's yaw
  double goal_yaw;
};

// min over T >= |dp|_inf / v_max of (optimal-control effort to reach the goal in T) + w T
MPLX_HD double cal_heur(const HeurParams &hp, int control, const State &s) {
  const double w = hp.w, v_max = hp.v_max;
  const State &goal = hp.goal;
  double dp[3];
  for (int i = 0; i < 3; i++) dp[i] = goal.p[i] - s.p[i];
  // v_max <= 0 is the "unlimited" sentinel (the setters' default, -1): no arrival-time bound exists, so
  // the kinematic term is dropped instead of dividing by a non-positive number
  if (hp.heur_ignore_dynamics) return v_max > 0 ? w * linf3(s.p, goal.p) / v_max : w * linf3(s.p, goal.p);
  const double *v0 = s.v, *v1 = goal.v, *a0 = s.a, *a1 = goal.a;
  double t_bar = v_max > 0 ? linf3(s.p, goal.p) / v_max : 0.0;
  const int gc = hp.goal_control & 15;  // (the yaw bit does not select another cost-to-go)
  control &= 15;
  if (control == CTRL_JRK && gc == CTRL_JRK) {
    double a0ma1[3] = {a0[0] - a1[0], a0[1] - a1[1], a0[2] - a1[2]};
    double v0pv1[3] = {v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2]};
    double c = -9 * dot3(a0, a0) + 6 * dot3(a0, a1) - 9 * dot3(a1, a1);
    double d = -144 * dot3(a0, v0) - 96 * dot3(a0, v1) + 96 * dot3(a1, v0) + 144 * dot3(a1, v1);
    double e = 360 * dot3(a0ma1, dp) - 576 * dot3(v0, v0) - 1008 * dot3(v0, v1) - 576 * dot3(v1, v1);
    double f = 2880 * dot3(dp, v0pv1);
    double g = -3600 * dot3(dp, dp);
    return heur_min6(w, c, d, e, f, g, t_bar);
  } else if (control == CTRL_JRK && gc == CTRL_ACC) {
    double c = -8 * dot3(a0, a0);
    double d = -112 * dot3(a0, v0) - 48 * dot3(a0, v1);
    double e = 240 * dot3(a0, dp) - 384 * dot3(v0, v0) - 432 * dot3(v0, v1) - 144 * dot3(v1, v1);
    double q[3] = {1600 * v0[0] + 960 * v1[0], 1600 * v0[1] + 960 * v1[1], 1600 * v0[2] + 960 * v1[2]};
    double f = dot3(dp, q);
    double g = -1600 * dot3(dp, dp);
    return heur_min6(w, c, d, e, f, g, t_bar);
  } else if (control == CTRL_JRK && gc == CTRL_VEL) {
    double c = -5 * dot3(a0, a0);
    double d = -40 * dot3(a0, v0);
    double e = 60 * dot3(a0, dp) - 60 * dot3(v0, v0);
    double f = 160 * dot3(dp, v0);
    double g = -100 * dot3(dp, dp);
    return heur_min6(w, c, d, e, f, g, t_bar);
  } else if (control == CTRL_ACC && gc == CTRL_ACC) {
    double v0pv1[3] = {v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2]};
    double c1 = -36 * dot3(dp, dp);
    double c2 = 24 * dot3(v0pv1, dp);
    double c3 = -4 * (dot3(v0, v0) + dot3(v0, v1) + dot3(v1, v1));
    return heur_min4(w, c3, c2, c1, w, t_bar);
  } else if (control == CTRL_ACC && gc == CTRL_VEL) {
    double c1 = -9 * dot3(dp, dp);
    double c2 = 12 * dot3(v0, dp);
    double c3 = -3 * dot3(v0, v0);
    return heur_min4(w, c3, c2, c1, w, t_bar);
  } else if (control == CTRL_VEL && gc == CTRL_VEL) {
    return (w + 1) * sqrt(dot3(dp, dp));
  }
  return v_max > 0 ? w * sqrt(dot3(dp, dp)) / v_max : w * sqrt(dot3(dp, dp));
}

// get_heur: 0 when the state's key equals the goal's key
MPLX_HD double get_heur(const HeurParams &hp, int control, const State &s, const int32_t *key, int nkey) {
  if ((control & 15) == (hp.goal_control & 15) && nkey == hp.goal_nkey) {
    uint32_t diff = 0;  // (no short circuit: a chain of dependent LDS reads on the device otherwise)
    for (int i = 0; i < nkey; i++) diff |= (uint32_t)(key[i] ^ hp.goal_key[i]);
    if (diff == 0u) return 0.0;
  }
  return cal_heur(hp, control, s);
}

// is_goal (t_max handled by the caller)
MPLX_HD bool is_goal_state(const State &s, const State &goal, int goal_control, double tol_pos, double tol_vel, double tol_acc) {
  bool goaled = linf3(s.p, goal.p) <= tol_pos;
  if (goaled && (goal_control & 2) && tol_vel >= 0) goaled = linf3(s.v, goal.v) <= tol_vel;
  if (goaled && (goal_control & 4) && tol_acc >= 0) goaled = linf3(s.a, goal.a) <= tol_acc;
  return goaled;
}

// 64-bit mix of a key tuple (table index + tag); not part of any result.  Two independent 32-bit
// multiply-xorshift lanes (32-bit multiplies are 4x cheaper than 64-bit ones on CDNA).
MPLX_HD uint64_t key_hash64(const int32_t *k, int n) {
  uint32_t a = 0x9E3779B9u, b = 0x85EBCA6Bu;
  for (int i = 0; i < n; i++) {
    const uint32_t x = (uint32_t)k[i];
    a = (a ^ x) * 0x01000193u;
    a ^= a >> 15;
    b = (b + x) * 0xC2B2AE35u;
    b ^= b >> 13;
  }
  a = (a ^ (b >> 3)) * 0x2C1B3C6Du;
  a ^= a >> 12;
  b = (b ^ (a << 7)) * 0x297A2D39u;
  b ^= b >> 15;
  return ((uint64_t)a << 32) | (uint64_t)b;
}

}  // namespace mplx
