// mplx_poly_dev.h -- device side of the moving-obstacle (PolyMap) environment, 2-D.
//
// Restates, operation for operation, the reference's in-tree arithmetic (paths relative to the reference repo):
//   mpl_external_planner/include/mpl_external_planner/poly_map_planner/env_poly_map.h:45-73   get_succ, intrinsic cost
//   .../poly_map_planner/poly_map_util.h:72-109                     isInside, isFree(pt, t), isFree(pr, t)
//   .../poly_map_planner/primitive_geometry_utils.h:5-173           collide() x 3 (static / linear / nonlinear obstacle)
//   .../poly_map_planner/simple_obstacle.h:6-240                    obstacle classes (inside, poly(t))
// and what those call from the un-vendored submodules, as include/mpl_shim states it (the compiled-reference checker
// of the tests is built against the same statements): Primitive1D p/v/a/j, Primitive::J,
// validate_primitive, Trajectory::evaluate, solve(), Polyhedron::inside (DecompUtil, epsilon 1e-10).
// Every sum keeps the reference's order (e.g. `a += n(i) * cs[i](0)` over i, then `a /= 120`), so results are
// bit-identical to the host evaluation; compiled -ffp-contract=off like the rest.
//
// Restricted to VEL / ACC primitives and obstacle trajectories (what the multi-robot node uses): the hyperplane
// equation is then at most quadratic in t.  A higher degree is reported (POLY_UNSUPPORTED), never approximated.
#pragma once
#include "mplx_math.h"

namespace mplx {

constexpr double POLY_EPS = 1e-10;  // Polyhedron::inside tolerance (DecompUtil epsilon_)
constexpr int POLY_MAX_U = 32;

struct PolyHP { double px, py, nx, ny; };            // Hyperplane2D: point p_, outward normal n_
struct PolySeg { double c[2][6]; double T; };        // Primitive2D of an obstacle trajectory
struct PolyObs {
  int32_t kind;                                      // 0 static, 1 linear, 2 nonlinear
  int32_t hp_off, n_hp, seg_off, n_seg, dis_front, dis_back, pad;
  double p[2], v[2], cov_v, start_t, total_t;        // representative point, velocity (linear), trajectory start / length
};
struct PolyWorld {                                   // what one planner sees: bounding box + obstacle set + start time
  int32_t obs_off, n_obs;
  double start_t;
  PolyHP bbox[4];
};
struct PolyDev {
  const PolyHP *hps;
  const PolySeg *segs;
  const PolyObs *obs;
  const PolyWorld *worlds;
  int32_t control, n_u;
  const double *U;                                   // n_u x 2
  double dt, v_max, a_max, j_max, w;
};

// ---- Primitive1D as include/mpl_shim/mpl_basis/primitive.h writes it (left-to-right products)
MPLX_HD double pp_p(const double *c, double t) {
  return c[0] / 120 * t * t * t * t * t + c[1] / 24 * t * t * t * t + c[2] / 6 * t * t * t + c[3] / 2 * t * t + c[4] * t + c[5];
}
MPLX_HD double pp_v(const double *c, double t) { return c[0] / 24 * t * t * t * t + c[1] / 6 * t * t * t + c[2] / 2 * t * t + c[3] * t + c[4]; }
MPLX_HD double pp_a(const double *c, double t) { return c[0] / 6 * t * t * t + c[1] / 2 * t * t + c[2] * t + c[3]; }
MPLX_HD double pp_j(const double *c, double t) { return c[0] / 2 * t * t + c[1] * t + c[2]; }
// Primitive1D::J(t, control): double sum over the derivative's monomial coefficients, ascending (i, j)
MPLX_HD double pp_J(const double *c, double t, int control) {
  const int k = (control & 15) == CTRL_VEL ? 1 : (control & 15) == CTRL_ACC ? 2 : (control & 15) == CTRL_JRK ? 3 : 4;
  const double fact[6] = {1, 1, 2, 6, 24, 120};
  double q[6];
  const int nq = 6 - k;
  for (int m = k; m <= 5; m++) q[m - k] = c[5 - m] / fact[m - k];
  double s = 0.0;
  for (int i = 0; i < nq; i++)
    for (int jj = 0; jj < nq; jj++) {
      double pw = t;
      for (int r = 1; r < i + jj + 1; r++) pw = pw * t;
      s += q[i] * q[jj] * pw / (double)(i + jj + 1);
    }
  return s;
}
// Polyhedron::inside(pt): every hyperplane has n . (pt - p) <= eps
MPLX_HD bool poly_inside(const PolyHP *hp, int n_hp, double x, double y) {
  for (int i = 0; i < n_hp; i++) {
    const double dx = x - hp[i].px, dy = y - hp[i].py;
    double s = 0.0;
    s += hp[i].nx * dx;
    s += hp[i].ny * dy;
    if (s > POLY_EPS) return false;
  }
  return true;
}
// Trajectory::evaluate(time) of an obstacle trajectory (include/mpl_shim/mpl_basis/trajectory.h): clamp to
// [0, total], find the segment by the cumulative times, evaluate at the local time
MPLX_HD void traj_eval(const PolySeg *segs, int n_seg, double total_t, double time, double pos[2], double vel[2], double acc[2], double jrk[2]) {
  pos[0] = pos[1] = vel[0] = vel[1] = acc[0] = acc[1] = jrk[0] = jrk[1] = 0.0;
  if (n_seg <= 0) return;
  const double tau = time < 0 ? 0 : (time > total_t ? total_t : time);
  double t0 = 0.0;
  for (int id = 0; id < n_seg; id++) {
    const double t1 = segs[id].T + t0;
    if ((tau >= t0 && tau < t1) || id + 1 == n_seg) {
      const double lt = tau - t0;
      for (int k = 0; k < 2; k++) {
        pos[k] = pp_p(segs[id].c[k], lt);
        vel[k] = pp_v(segs[id].c[k], lt);
        acc[k] = pp_a(segs[id].c[k], lt);
        jrk[k] = pp_j(segs[id].c[k], lt);
      }
      return;
    }
    t0 = t1;
  }
}
// solve(a, b, c, d, e, f) for degree <= 2 (mpl_shim/mpl_basis/math.h); returns -1 for a higher degree
MPLX_HD int solve_le2(double a, double b, double c, double d, double e, double f, double ts[2]) {
  if (a != 0 || b != 0 || c != 0) return -1;
  if (d != 0) {
    const double p = e * e - 4 * d * f;
    if (p < 0) return 0;
    ts[0] = (-e - sqrt(p)) / (2 * d);
    ts[1] = (-e + sqrt(p)) / (2 * d);
    return 2;
  }
  if (e != 0) {
    ts[0] = -f / e;
    return 1;
  }
  return 0;
}

// obstacle.inside(pt[, t]) of the three classes (simple_obstacle.h:29, :87-90, :131-142)
MPLX_HD bool obs_inside_static(const PolyDev &D, const PolyObs &o, double x, double y) { return poly_inside(D.hps + o.hp_off, o.n_hp, x - o.p[0], y - o.p[1]); }
MPLX_HD bool obs_inside_linear(const PolyDev &D, const PolyObs &o, double x, double y, double t) {
  // poly(t): every hyperplane point moves by v t + p + cov_v n t (simple_obstacle.h:81-85), then Polyhedron::inside
  const PolyHP *hp = D.hps + o.hp_off;
  for (int i = 0; i < o.n_hp; i++) {
    const double qx = hp[i].px + ((o.v[0] * t + o.p[0]) + (hp[i].nx * o.cov_v) * t);
    const double qy = hp[i].py + ((o.v[1] * t + o.p[1]) + (hp[i].ny * o.cov_v) * t);
    double s = 0.0;
    s += hp[i].nx * (x - qx);
    s += hp[i].ny * (y - qy);
    if (s > POLY_EPS) return false;
  }
  return true;
}
MPLX_HD bool obs_inside_nonlinear(const PolyDev &D, const PolyObs &o, double x, double y, double t) {
  t += o.start_t;
  double wp[2], wv[2], wa[2], wj[2];
  traj_eval(D.segs + o.seg_off, o.n_seg, o.total_t, t, wp, wv, wa, wj);
  if (t <= o.total_t && t >= 0) return poly_inside(D.hps + o.hp_off, o.n_hp, x - wp[0], y - wp[1]);
  if (t < 0 && !o.dis_front) return poly_inside(D.hps + o.hp_off, o.n_hp, x - wp[0], y - wp[1]);
  if (t > o.total_t && !o.dis_back) return poly_inside(D.hps + o.hp_off, o.n_hp, x - wp[0], y - wp[1]);
  return false;
}
// PolyMapUtil::isFree(pt, t) restricted to one obstacle (poly_map_util.h:75-88)
MPLX_HD bool obs_point_hits(const PolyDev &D, const PolyObs &o, double x, double y, double t_rel) {
  return o.kind == 0 ? obs_inside_static(D, o, x, y) : o.kind == 1 ? obs_inside_linear(D, o, x, y, t_rel) : obs_inside_nonlinear(D, o, x, y, t_rel);
}

// collide(pr, PolyhedronObstacle) with the obstacle's representative point (px, py) (primitive_geometry_utils.h:5-44)
// returns 1 hit, 0 free, -1 unsupported degree
MPLX_HD int collide_static_at(const PolyDev &D, const double cs[2][6], double T, const PolyObs &o, double px, double py) {
  const PolyHP *hp = D.hps + o.hp_off;
  for (int h = 0; h < o.n_hp; h++) {
    const double n[2] = {hp[h].nx, hp[h].ny};
    double a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
    for (int i = 0; i < 2; i++) {
      a += n[i] * cs[i][0];
      b += n[i] * cs[i][1];
      c += n[i] * cs[i][2];
      d += n[i] * cs[i][3];
      e += n[i] * cs[i][4];
      f += n[i] * cs[i][5];
    }
    a /= 120.0; b /= 24.0; c /= 6.0; d /= 2.0; e /= 1.0;
    {
      double s = 0.0;
      s += n[0] * (hp[h].px + px);
      s += n[1] * (hp[h].py + py);
      f -= s;
    }
    double ts[2];
    const int nr = solve_le2(a, b, c, d, e, f, ts);
    if (nr < 0) return -1;
    for (int r = 0; r < nr; r++) {
      const double it = ts[r];
      if (it >= 0 && it <= T) {
        const double wx = pp_p(cs[0], it), wy = pp_p(cs[1], it);
        if (poly_inside(hp, o.n_hp, wx - px, wy - py)) return 1;
      }
    }
  }
  return 0;
}
// collide(pr, PolyhedronLinearObstacle, t) (primitive_geometry_utils.h:46-94)
MPLX_HD int collide_linear(const PolyDev &D, const double cs[2][6], double T, const PolyObs &o, double t) {
  const PolyHP *hp = D.hps + o.hp_off;
  for (int h = 0; h < o.n_hp; h++) {
    const double n[2] = {hp[h].nx, hp[h].ny};
    const double cov_v[2] = {o.v[0] + n[0] * o.cov_v, o.v[1] + n[1] * o.cov_v};
    double a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
    for (int i = 0; i < 2; i++) {
      a += n[i] * cs[i][0];
      b += n[i] * cs[i][1];
      c += n[i] * cs[i][2];
      d += n[i] * cs[i][3];
      e += n[i] * cs[i][4];
      f += n[i] * cs[i][5];
    }
    a /= 120.0; b /= 24.0; c /= 6.0; d /= 2.0;
    {
      double s = 0.0;
      s += n[0] * cov_v[0];
      s += n[1] * cov_v[1];
      e -= s;
      double s2 = 0.0;
      s2 += n[0] * ((hp[h].px + o.p[0]) + cov_v[0] * t);
      s2 += n[1] * ((hp[h].py + o.p[1]) + cov_v[1] * t);
      f -= s2;
    }
    double ts[2];
    const int nr = solve_le2(a, b, c, d, e, f, ts);
    if (nr < 0) return -1;
    for (int r = 0; r < nr; r++) {
      const double it = ts[r];
      if (it >= 0 && it <= T) {
        const double wx = pp_p(cs[0], it), wy = pp_p(cs[1], it);
        if (obs_inside_linear(D, o, wx, wy, it + t)) return 1;
      }
    }
  }
  return 0;
}
// collide(pr, PolyhedronNonlinearObstacle, t) (primitive_geometry_utils.h:96-173)
MPLX_HD int collide_nonlinear(const PolyDev &D, const double cs[2][6], double prT, const PolyObs &o, double t) {
  const PolySeg *segs = D.segs + o.seg_off;
  const double traj_t = t + o.start_t;
  int start_id = -1;
  double T = 0.0;  // current segment start time
  for (int i = 0; i < o.n_seg; i++) {
    if (traj_t >= T && traj_t < T + segs[i].T) {
      start_id = i;
      break;
    }
    T += segs[i].T;
  }
  if (start_id < 0) {  // outside the trajectory's time span: its clamped end state as a static obstacle, or nothing
    double wp[2], wv[2], wa[2], wj[2];
    traj_eval(segs, o.n_seg, o.total_t, traj_t, wp, wv, wa, wj);
    if (traj_t <= o.total_t && traj_t >= 0) return collide_static_at(D, cs, prT, o, wp[0], wp[1]);
    if (traj_t < 0 && !o.dis_front) return collide_static_at(D, cs, prT, o, wp[0], wp[1]);
    if (traj_t > o.total_t && !o.dis_back) return collide_static_at(D, cs, prT, o, wp[0], wp[1]);
    return 0;
  }
  const PolyHP *hp = D.hps + o.hp_off;
  for (int id = start_id; id < o.n_seg; id++) {
    const double t_residual = T - traj_t < 0 ? 0 : T - traj_t;
    const double start_t = t_residual <= 0 ? traj_t : T;
    if (t_residual > prT) break;
    double wp[2], wv[2], wa[2], wj[2];
    traj_eval(segs, o.n_seg, o.total_t, start_t, wp, wv, wa, wj);
    for (int h = 0; h < o.n_hp; h++) {
      const double n[2] = {hp[h].nx, hp[h].ny};
      const double hpp[2] = {hp[h].px, hp[h].py};
      double a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
      for (int i = 0; i < 2; i++) {
        a += n[i] * cs[i][0];
        b += n[i] * cs[i][1];
        c += n[i] * cs[i][2] - n[i] * wj[i];
        d += n[i] * cs[i][3] - n[i] * wa[i];
        e += n[i] * cs[i][4] - n[i] * wv[i];
        f += n[i] * cs[i][5] - n[i] * (hpp[i] + wp[i]);
      }
      a /= 120; b /= 24; c /= 6; d /= 2;
      double ts[2];
      const int nr = solve_le2(a, b, c, d, e, f, ts);
      if (nr < 0) return -1;
      for (int r = 0; r < nr; r++) {
        const double it = ts[r];
        if (it >= t_residual && it <= prT && T + segs[id].T >= it + start_t && T <= it + start_t) {
          const double cx = pp_p(cs[0], it), cy = pp_p(cs[1], it);
          if (obs_inside_nonlinear(D, o, cx, cy, it + t)) return 1;
        }
      }
    }
    T += segs[id].T;
  }
  return 0;
}
// PolyMapUtil::isFree(pr, t) restricted to one obstacle (poly_map_util.h:92-109; the start-point test is separate)
MPLX_HD int obs_prim_hits(const PolyDev &D, const double cs[2][6], double T, const PolyObs &o, double t_rel) {
  return o.kind == 0 ? collide_static_at(D, cs, T, o, o.p[0], o.p[1]) : o.kind == 1 ? collide_linear(D, cs, T, o, t_rel) : collide_nonlinear(D, cs, T, o, t_rel);
}

// Primitive<2>(curr, u, dt) coefficients (mpl_shim primitive.h) for VEL / ACC
MPLX_HD void poly_prim_build(int control, const double pos[2], const double vel[2], const double u[2], double cs[2][6]) {
  for (int i = 0; i < 2; i++) {
    for (int k = 0; k < 6; k++) cs[i][k] = 0.0;
    if ((control & 15) == CTRL_VEL) { cs[i][4] = u[i]; cs[i][5] = pos[i]; }
    else { cs[i][3] = u[i]; cs[i][4] = vel[i]; cs[i][5] = pos[i]; }
  }
}
// validate_primitive for VEL / ACC (mpl_shim primitive.h): ACC checks max |vel| per axis against v_max > 0; the
// velocity of such a primitive is monotone, so its extrema are the end points
MPLX_HD bool poly_validate(int control, const double cs[2][6], double T, double v_max) {
  if ((control & 15) != CTRL_ACC) return true;
  for (int i = 0; i < 2; i++) {
    const double m = fmax(fabs(pp_v(cs[i], 0.0)), fabs(pp_v(cs[i], T)));
    if (v_max > 0 && m > v_max) return false;
  }
  return true;
}
// env_poly_map::calculate_intrinsic_cost: pr.J(pr.control()) + 0.001 * pr.J(Control::VEL) + w dt (env_poly_map.h:71-73)
MPLX_HD double poly_intrinsic_cost(int control, const double cs[2][6], double T, double w, double dt) {
  double jc = 0;
  for (int k = 0; k < 2; k++) jc += pp_J(cs[k], T, control);
  double jv = 0;
  for (int k = 0; k < 2; k++) jv += pp_J(cs[k], T, CTRL_VEL);
  return jc + 0.001 * jv + w * dt;
}

}  // namespace mplx
