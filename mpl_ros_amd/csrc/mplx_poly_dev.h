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
// Two builds of every routine (template flag GEN).  GEN = false: VEL / ACC primitives and obstacle trajectories (what
// the multi-robot node uses) -- the hyperplane equation is then at most quadratic in t, and a higher degree is reported
// (never approximated).  GEN = true: any degree up to five through solve_any6, the statement-for-statement restatement of
// include/mpl_shim/mpl_basis/math.h's solve() (poly_map_planner_node.cpp:73-85 exposes use_acc / use_jrk; obstacle
// trajectories of JRK robots are cubic).  The host launches GEN = true only when a degree above two can occur.
#pragma once
#include "mplx_math.h"

namespace mplx {

constexpr double POLY_EPS = 1e-10;  // Polyhedron::inside tolerance (DecompUtil epsilon_)
constexpr int POLY_MAX_U = 32;

struct PolyHP { double px, py, nx, ny; };            // Hyperplane2D: point p_, outward normal n_
struct PolySeg { double c[2][6]; double T; };        // Primitive2D of an obstacle trajectory
struct PolyObs {
  int32_t kind;                                      // 0 static, 1 linear, 2 nonlinear
  int32_t hp_off, n_hp, seg_off, n_seg, dis_front, dis_back;
  int32_t cum_off;                                   // (device staging) offset of this trajectory's segment start times in PolyDev::cum
  double radius;                                     // bounding radius of the polyhedron around its reference point (+inf: unknown / unbounded): pruning only
  int32_t fast, pad2;                                // trajectory of VEL / ACC segments with positive durations (set at commit): shortcuts that need it
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
  const double *cum;                                 // null, or per trajectory n_seg + 1 segment start times (cum[0] = 0, cum[k + 1] = segs[k].T + cum[k]:
                                                     // the very sums the loops below accumulate), staged with the world in LDS
  struct PolyPrep *prep_cache;                       // null, or per workgroup POLY_CACHE_LEVELS x POLY_MAX_OBS prepared obstacles (below)
  // look-ahead by helper workgroups (mplx_poly_search.h): the collision outcome of a state -- isFree(start.pos, t) and
  // isFree(pr, t) of its primitives -- is a pure function of the state, so workgroups on otherwise idle compute units
  // compute it for the states a search has just created, before the search pops them
  unsigned long long *help_mask;                     // per node-pool record: POLY_MASK_READY | start-hit / unsupported bits | hit bit per control input
  double *help_ring;                                 // per leader: 2^help_ring_log entries of 8 doubles {id << 32 | record index, pos2, vel2, acc2, t}, indexed by state id
  unsigned long long *help_pub;                      // per leader: states published so far | POLY_PUB_DONE
  int32_t help_ring_log, n_help;                     // helpers per leader (0: no helpers in this launch)
  int32_t prep_slice0, pad3;                         // workgroup b of this launch owns slice prep_slice0 + b of prep_cache
};
constexpr unsigned long long POLY_MASK_READY = 1ull << 63, POLY_MASK_UNSUP = 1ull << 62, POLY_MASK_START = 1ull << 61;  // bits 0..30: hit of control input i
constexpr unsigned long long POLY_PUB_DONE = 1ull << 63;

// ---- Primitive1D as include/mpl_shim/mpl_basis/primitive.h writes it (left-to-right products)
MPLX_HD double pp_p(const double *c, double t) {
  return c[0] / 120 * t * t * t * t * t + c[1] / 24 * t * t * t * t + c[2] / 6 * t * t * t + c[3] / 2 * t * t + c[4] * t + c[5];
}
MPLX_HD double pp_v(const double *c, double t) { return c[0] / 24 * t * t * t * t + c[1] / 6 * t * t * t + c[2] / 2 * t * t + c[3] * t + c[4]; }
MPLX_HD double pp_a(const double *c, double t) { return c[0] / 6 * t * t * t + c[1] / 2 * t * t + c[2] * t + c[3]; }
MPLX_HD double pp_j(const double *c, double t) { return c[0] / 2 * t * t + c[1] * t + c[2]; }
// +0.0 exactly (a -0.0 coefficient would make the dropped terms -0.0, and a sum of those is not absorbed the same way)
MPLX_HD bool is_pzero(double x) { return x == 0.0 && !__builtin_signbit(x); }
MPLX_HD bool lead_pzero(const double *c) { return is_pzero(c[0]) && is_pzero(c[1]) && is_pzero(c[2]); }
// short evaluation of a polynomial whose three leading coefficients are zero (VEL / ACC primitives and trajectories):
// with t >= 0 the dropped terms are +0.0, so (x + 0.0) reproduces the full expression bit for bit (as pos_at_c in mplx_math.h)
MPLX_HD double pp_p_auto(const double *c, double t) {
  if (lead_pzero(c) && t >= 0.0) return ((c[3] / 2 * t * t + 0.0) + c[4] * t) + c[5];
  return pp_p(c, t);
}
// the same for the derivatives: with c0 = c1 = c2 = 0 and t >= 0 every dropped term is +0.0
MPLX_HD double pp_v_auto(const double *c, double t) {
  if (lead_pzero(c) && t >= 0.0) return (c[3] * t + 0.0) + c[4];
  return pp_v(c, t);
}
MPLX_HD double pp_a_auto(const double *c, double t) {
  if (lead_pzero(c) && t >= 0.0) return c[3] + 0.0;
  return pp_a(c, t);
}
MPLX_HD double pp_j_auto(const double *c, double t) {
  if (lead_pzero(c) && t >= 0.0) return 0.0;
  return pp_j(c, t);
}
// x / k for a divisor k > 0: a zero numerator keeps its sign, so the division is only executed for x != 0
MPLX_HD double div_nz(double x, double k) { return x == 0.0 ? x : x / k; }
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
MPLX_HD void traj_eval(const PolySeg *segs, int n_seg, double total_t, double time, double pos[2], double vel[2], double acc[2], double jrk[2], const double *cum = nullptr) {
  pos[0] = pos[1] = vel[0] = vel[1] = acc[0] = acc[1] = jrk[0] = jrk[1] = 0.0;
  if (n_seg <= 0) return;
  const double tau = time < 0 ? 0 : (time > total_t ? total_t : time);
  double t0 = 0.0;
  for (int id = 0; id < n_seg; id++) {
    const double t1 = cum ? cum[id + 1] : segs[id].T + t0;
    if ((tau >= t0 && tau < t1) || id + 1 == n_seg) {
      const double lt = tau - t0;
      for (int k = 0; k < 2; k++) {  // (lt >= 0: the short forms apply to VEL / ACC segments)
        pos[k] = pp_p_auto(segs[id].c[k], lt);
        vel[k] = pp_v_auto(segs[id].c[k], lt);
        acc[k] = pp_a_auto(segs[id].c[k], lt);
        jrk[k] = pp_j_auto(segs[id].c[k], lt);
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

// ---- solve(a, b, c, d, e, f) for any degree (include/mpl_shim/mpl_basis/math.h, statement for statement: the
// compiled-reference checker of the tests links that very header).  Degree <= 2: the formulas of solve_le2.  Degree 3..5:
// real_roots() -- the roots of the derivative (recursively) and the Cauchy bound split the axis into monotone pieces,
// every sign change is bisected (200 steps at most, until hi - lo <= 4e-16 |hi + lo|), roots ascending.
MPLX_HD double poly_eval_n(const double *a, int n, double x) {
  double r = a[n];
  for (int i = n - 1; i >= 0; i--) r = r * x + a[i];
  return r;
}
// roots of sum a[i] x^i (a[n] != 0, 2 <= n <= 5) given the roots of its derivative (crit, ascending)
MPLX_HD int poly_roots_level(const double *a, int n, const double *crit, int n_crit, double *out) {
  double m = 0;
  for (int i = 0; i < n; i++) m = fabs(a[i] / a[n]) > m ? fabs(a[i] / a[n]) : m;
  const double bound = 1.0 + m;  // Cauchy
  double xs[7];
  int nx = 0;
  xs[nx++] = -bound;
  for (int k = 0; k < n_crit; k++)
    if (crit[k] > -bound && crit[k] < bound) xs[nx++] = crit[k];
  xs[nx++] = bound;
  int no = 0;
  for (int k = 0; k + 1 < nx; k++) {
    double lo = xs[k], hi = xs[k + 1], flo = poly_eval_n(a, n, lo);
    const double fhi = poly_eval_n(a, n, hi);
    if (flo == 0.0) {
      if (no == 0 || out[no - 1] != lo) out[no++] = lo;
      continue;
    }
    if (fhi == 0.0 || (flo < 0) == (fhi < 0)) {
      if (fhi == 0.0 && k + 2 == nx) out[no++] = hi;
      continue;
    }
    for (int it = 0; it < 200 && hi - lo > 4e-16 * fabs(hi + lo); it++) {
      const double mid = 0.5 * (lo + hi), fm = poly_eval_n(a, n, mid);
      if (fm == 0.0) { lo = hi = mid; break; }
      if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; } else hi = mid;
    }
    out[no++] = 0.5 * (lo + hi);
  }
  return no;
}
// real_roots(a, n), 1 <= n <= 5, a[n] != 0: the recursion unrolled -- the chain of derivatives downwards, the roots upwards
MPLX_HD int poly_real_roots(const double *a, int n, double *out) {
  double P[5][6];
  int deg[5], L = 0;
  for (int i = 0; i <= n; i++) P[0][i] = a[i];
  deg[0] = n;
  bool no_crit = false;  // the deepest level's derivative is a constant: no critical points
  while (deg[L] >= 2) {
    const int nn = deg[L];
    for (int i = 1; i <= nn; i++) P[L + 1][i - 1] = P[L][i] * i;
    int nd = nn - 1;
    while (nd > 0 && P[L + 1][nd] == 0.0) nd--;
    if (nd < 1) { no_crit = true; break; }
    deg[L + 1] = nd;
    L++;
  }
  double ra[5], rb[5];
  double *cur = ra, *nxt = rb;
  int nc = 0;
  if (deg[L] == 1) {
    cur[0] = -P[L][0] / P[L][1];
    nc = 1;
  } else {  // (no_crit) a level of degree >= 2 whose derivative has no root
    nc = poly_roots_level(P[L], deg[L], cur, 0, nxt);
    double *t = cur; cur = nxt; nxt = t;
  }
  (void)no_crit;
  for (int k = L - 1; k >= 0; k--) {
    nc = poly_roots_level(P[k], deg[k], cur, nc, nxt);
    double *t = cur; cur = nxt; nxt = t;
  }
  for (int i = 0; i < nc; i++) out[i] = cur[i];
  return nc;
}
#ifdef __HIPCC__
#define MPLX_NOINLINE __noinline__
#else
#define MPLX_NOINLINE
#endif
// solve(a, b, c, d, e, f): a t^5 + b t^4 + c t^3 + d t^2 + e t + f = 0; ts has room for 5
MPLX_HD MPLX_NOINLINE int solve_any6(double a, double b, double c, double d, double e, double f, double *ts) {
  const double co[6] = {f, e, d, c, b, a};
  if (a != 0) return poly_real_roots(co, 5, ts);
  if (b != 0) return poly_real_roots(co, 4, ts);
  if (c != 0) return poly_real_roots(co, 3, ts);
  return solve_le2(0, 0, 0, d, e, f, ts);
}
template <bool GEN>
MPLX_HD int solve_poly(double a, double b, double c, double d, double e, double f, double *ts) {
  if constexpr (GEN) {
    if (a != 0 || b != 0 || c != 0) return solve_any6(a, b, c, d, e, f, ts);
  }
  return solve_le2(a, b, c, d, e, f, ts);
}
constexpr int POLY_MAX_ROOTS = 5;
// Primitive1D::max_abs(k, t): max over [0, t] of |d^k p| (k = 1 vel, 2 acc, 3 jrk): the end points and the interior
// stationary points, which are the roots `solve` returns for the next derivative (mpl_shim primitive.h extrema())
MPLX_HD double poly_max_abs(const double *c, int k, double t) {
  const double f0 = k == 1 ? pp_v(c, 0.0) : k == 2 ? pp_a(c, 0.0) : pp_j(c, 0.0);
  const double f1 = k == 1 ? pp_v(c, t) : k == 2 ? pp_a(c, t) : pp_j(c, t);
  double m = fabs(f0) < fabs(f1) ? fabs(f1) : fabs(f0);  // std::max(|f(0)|, |f(t)|)
  double ts[POLY_MAX_ROOTS];
  const int nr = k == 1 ? solve_any6(0, 0, c[0] / 6, c[1] / 2, c[2], c[3], ts) : k == 2 ? solve_any6(0, 0, 0, c[0] / 2, c[1], c[2], ts) : solve_any6(0, 0, 0, 0, c[0], c[1], ts);
  for (int r = 0; r < nr; r++) {
    const double x = ts[r];
    if (x > 0 && x < t) {
      const double fx = fabs(k == 1 ? pp_v(c, x) : k == 2 ? pp_a(c, x) : pp_j(c, x));
      m = m < fx ? fx : m;
    }
  }
  return m;
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
MPLX_HD bool obs_inside_nonlinear_pos(const PolyDev &D, const PolyObs &o, double x, double y, double t, int id0 = 0);  // (below: position only)
// PolyMapUtil::isFree(pt, t) restricted to one obstacle (poly_map_util.h:75-88)
MPLX_HD bool obs_point_hits(const PolyDev &D, const PolyObs &o, double x, double y, double t_rel) {
  return o.kind == 0 ? obs_inside_static(D, o, x, y) : o.kind == 1 ? obs_inside_linear(D, o, x, y, t_rel) : obs_inside_nonlinear_pos(D, o, x, y, t_rel);
}

// collide(pr, PolyhedronObstacle) with the obstacle's representative point (px, py) (primitive_geometry_utils.h:5-44)
// returns 1 hit, 0 free, -1 unsupported degree
template <bool GEN = false>
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
    double ts[GEN ? POLY_MAX_ROOTS : 2];
    const int nr = solve_poly<GEN>(a, b, c, d, e, f, ts);
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
template <bool GEN = false>
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
    double ts[GEN ? POLY_MAX_ROOTS : 2];
    const int nr = solve_poly<GEN>(a, b, c, d, e, f, ts);
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
template <bool GEN = false>
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
    if (traj_t <= o.total_t && traj_t >= 0) return collide_static_at<GEN>(D, cs, prT, o, wp[0], wp[1]);
    if (traj_t < 0 && !o.dis_front) return collide_static_at<GEN>(D, cs, prT, o, wp[0], wp[1]);
    if (traj_t > o.total_t && !o.dis_back) return collide_static_at<GEN>(D, cs, prT, o, wp[0], wp[1]);
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
      double ts[GEN ? POLY_MAX_ROOTS : 2];
      const int nr = solve_poly<GEN>(a, b, c, d, e, f, ts);
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
template <bool GEN = false>
MPLX_HD int obs_prim_hits(const PolyDev &D, const double cs[2][6], double T, const PolyObs &o, double t_rel) {
  return o.kind == 0 ? collide_static_at<GEN>(D, cs, T, o, o.p[0], o.p[1]) : o.kind == 1 ? collide_linear<GEN>(D, cs, T, o, t_rel) : collide_nonlinear<GEN>(D, cs, T, o, t_rel);
}


// ------------------------------------------------------------------ get_succ spread over a workgroup
// isFree(pr, t) of all primitives of one node against all obstacles, decomposed below the (primitive, obstacle) pair:
// the reference's collide() loops -- obstacle-trajectory segment x hyperplane x root -- run one (segment, hyperplane)
// per lane, and what only depends on the obstacle and the node's time (which segments the primitive's time span
// overlaps, the obstacle's state at their start) is prepared once per obstacle instead of once per pair.  Every
// arithmetic expression and every sum order is the one of the functions above (the boolean results are OR-ed, which
// has no order); an "unsupported degree" only counts when the reference's loop would have reached it before a hit.
// [one lane per pair, all loops serial, took 50 k cycles per expansion: 78 % of a search]
constexpr int POLY_SLOTS = 3;     // trajectory segments one primitive may overlap here (more: that pair runs the serial collide())
constexpr int POLY_MAX_OBS = 64;  // obstacles prepared per node (a world with more runs the serial collide() per pair)
struct PolySlot {
  double T, t_residual, start_t, seg_T;  // segment start time, max(T - traj_t, 0), evaluation time, segment duration
  double wp[2], wv[2], wa[2], wj[2];     // obstacle state at start_t
};
struct alignas(16) PolyPrep {  // (a multiple of 16 bytes: a time level of the cache is copied to LDS in 16-byte words)
  int32_t mode;     // 0 cannot collide, 1 static polyhedron at (px, py), 2 linear, 3 trajectory slots, 4 serial collide()
  int32_t n_slots;
  double px, py;
  // pruning (conservative, never changes a result): during the primitive's time span every point of the obstacle stays
  // within `reach` (per axis) of (cx, cy); +inf: no bound known
  double cx, cy, reach;
  int32_t id0;       // first segment the span overlaps (mode 3)
  int32_t start_hit; // isFree(start.pos, t) fails against this obstacle (the node position is the same for every primitive)
  unsigned long long tag_t;  // (cache) bits of the time the entry was prepared for
  long long tag_q;           // (cache) query + 1 of the launch (0: empty; the host clears the cache per launch)
  PolySlot slot[POLY_SLOTS];
};
constexpr double POLY_PRUNE_EPS = 1e-6;
// What poly_prepare computes depends on the obstacle and on the node's TIME only, and a search visits few distinct times
// (start + k dt): the prepared obstacles are kept per time level in HBM (per workgroup) and re-used by every later
// expansion at that level.  An entry is valid for (query, exact time bits); anything else recomputes and overwrites.
constexpr int POLY_CACHE_LEVELS = 64;
// position part of traj_eval (same segment selection)
// id0: a segment known to start at or before `time` (0 when nothing is known).  With strictly increasing segment start
// times the scan from id0 finds the segment the scan from 0 finds; callers pass id0 > 0 only for such trajectories
// (PolyObs::fast) and only together with cum.
MPLX_HD void traj_pos(const PolySeg *segs, int n_seg, double total_t, double time, double pos[2], const double *cum = nullptr, int id0 = 0) {
  pos[0] = pos[1] = 0.0;
  if (n_seg <= 0) return;
  const double tau = time < 0 ? 0 : (time > total_t ? total_t : time);
  double t0 = (cum && id0 > 0) ? cum[id0] : 0.0;
  for (int id = (cum && id0 > 0) ? id0 : 0; id < n_seg; id++) {
    const double t1 = cum ? cum[id + 1] : segs[id].T + t0;
    if ((tau >= t0 && tau < t1) || id + 1 == n_seg) {
      const double lt = tau - t0;
      pos[0] = pp_p_auto(segs[id].c[0], lt);
      pos[1] = pp_p_auto(segs[id].c[1], lt);
      return;
    }
    t0 = t1;
  }
}
MPLX_HD bool obs_inside_nonlinear_pos(const PolyDev &D, const PolyObs &o, double x, double y, double t, int id0) {
  t += o.start_t;
  double wp[2];
  traj_pos(D.segs + o.seg_off, o.n_seg, o.total_t, t, wp, D.cum ? D.cum + o.cum_off : nullptr, o.fast ? id0 : 0);
  if (t <= o.total_t && t >= 0) return poly_inside(D.hps + o.hp_off, o.n_hp, x - wp[0], y - wp[1]);
  if (t < 0 && !o.dis_front) return poly_inside(D.hps + o.hp_off, o.n_hp, x - wp[0], y - wp[1]);
  if (t > o.total_t && !o.dis_back) return poly_inside(D.hps + o.hp_off, o.n_hp, x - wp[0], y - wp[1]);
  return false;
}
// state of a trajectory at `time`, the segment known (fast trajectories: start times strictly increase, so segment id
// with cum[id] <= time < cum[id + 1] is the one traj_eval's scan finds)
MPLX_HD void seg_eval(const PolySeg &sg, double lt, double pos[2], double vel[2], double acc[2], double jrk[2]) {
  for (int k = 0; k < 2; k++) {
    pos[k] = pp_p_auto(sg.c[k], lt);
    vel[k] = pp_v_auto(sg.c[k], lt);
    acc[k] = pp_a_auto(sg.c[k], lt);
    jrk[k] = pp_j_auto(sg.c[k], lt);
  }
}
// per obstacle: the head of collide() for a primitive of duration prT starting at t (relative to the world's start),
// the start-point test isFree((x0, y0), t) and the pruning bound
MPLX_HD void poly_prepare_time(const PolyDev &D, const PolyObs &o, double prT, double t, PolyPrep &R) {
  R.n_slots = 0;
  R.id0 = 0;
  R.px = o.p[0];
  R.py = o.p[1];
  R.cx = o.p[0];
  R.cy = o.p[1];
  R.reach = INFINITY;
  R.start_hit = 0;
  if (o.kind == 0) {
    R.mode = 1;
    R.reach = o.radius + POLY_PRUNE_EPS;
    return;
  }
  if (o.kind == 1) {
    R.mode = 2;
    return;
  }
  const PolySeg *segs = D.segs + o.seg_off;
  const double *cum = D.cum ? D.cum + o.cum_off : nullptr;
  const bool fast = o.fast && cum;
  const double traj_t = t + o.start_t;
  int start_id = -1;
  double T = 0.0;
  for (int i = 0; i < o.n_seg; i++) {
    const double T1 = cum ? cum[i + 1] : T + segs[i].T;
    if (traj_t >= T && traj_t < T1) {
      start_id = i;
      break;
    }
    T = T1;  // (T += segs[i].T: the same sum)
  }
  if (start_id < 0) {
    double wp[2], wv[2], wa[2], wj[2];
    traj_eval(segs, o.n_seg, o.total_t, traj_t, wp, wv, wa, wj, cum);
    const bool there = (traj_t <= o.total_t && traj_t >= 0) || (traj_t < 0 && !o.dis_front) || (traj_t > o.total_t && !o.dis_back);
    R.mode = there ? 1 : 0;
    R.px = wp[0];
    R.py = wp[1];
    R.cx = wp[0];
    R.cy = wp[1];
    R.reach = o.radius + POLY_PRUNE_EPS;
    return;
  }
  R.mode = 3;
  R.id0 = start_id;
  double disp[2] = {0.0, 0.0};
  for (int id = start_id; id < o.n_seg; id++) {
    const double t_residual = T - traj_t < 0 ? 0 : T - traj_t;
    const double start_t = t_residual <= 0 ? traj_t : T;
    if (t_residual > prT) break;
    if (R.n_slots == POLY_SLOTS) { R.mode = 4; break; }
    PolySlot &S = R.slot[R.n_slots++];
    S.T = T; S.t_residual = t_residual; S.start_t = start_t; S.seg_T = segs[id].T;
    if (fast) {  // (start_t lies inside segment id: what traj_eval's scan finds; VEL / ACC segment, lt >= 0: the short forms)
      const double lt = start_t - T;
      for (int k = 0; k < 2; k++) {
        const double c3 = segs[id].c[k][3], c4 = segs[id].c[k][4], c5 = segs[id].c[k][5];
        S.wp[k] = ((c3 / 2 * lt * lt + 0.0) + c4 * lt) + c5;
        S.wv[k] = (c3 * lt + 0.0) + c4;
        S.wa[k] = c3 + 0.0;
        S.wj[k] = 0.0;
      }
    }
    else traj_eval(segs, o.n_seg, o.total_t, start_t, S.wp, S.wv, S.wa, S.wj, cum);
    for (int k = 0; k < 2; k++) disp[k] += fabs(S.wv[k]) * prT + 0.5 * fabs(S.wa[k]) * prT * prT;
    T = cum ? cum[id + 1] : T + segs[id].T;
  }
  if (R.mode == 3 && fast) {  // VEL / ACC segments: |c(t) - c(traj_t)| <= sum over the slots of |v| prT + |a| prT^2 / 2
    R.cx = R.slot[0].wp[0];
    R.cy = R.slot[0].wp[1];
    R.reach = (disp[0] > disp[1] ? disp[0] : disp[1]) + o.radius + POLY_PRUNE_EPS;
  }
}
// isFree((x0, y0), t) against one obstacle, from its prepared state: PolyMapUtil::isFree(pt, t) (poly_map_util.h:75-88)
//   static: inside(p_);  trajectory outside its time span: inside(clamped end) unless it has disappeared (mode 0);
//   inside its span: the obstacle at traj_t = slot 0's evaluation point (the first slot starts at traj_t)
MPLX_HD bool poly_start_test(const PolyDev &D, const PolyObs &o, const PolyPrep &R, double x0, double y0, double t) {
  if (R.mode == 0) return false;
  if (R.mode == 1) return poly_inside(D.hps + o.hp_off, o.n_hp, x0 - R.px, y0 - R.py);
  if (R.mode == 2) return obs_inside_linear(D, o, x0, y0, t);
  if (R.mode == 3) return poly_inside(D.hps + o.hp_off, o.n_hp, x0 - R.slot[0].wp[0], y0 - R.slot[0].wp[1]);
  return obs_inside_nonlinear_pos(D, o, x0, y0, t, 0);
}
// one (slot, hyperplane) of collide(): 1 hit, 0 free, -1 unsupported degree
template <bool GEN = false>
MPLX_HD int poly_item(const PolyDev &D, const double (*cs)[6], double prT, const PolyObs &o, const PolyPrep &R, int s, int h, double t) {
  const PolyHP *hp = D.hps + o.hp_off;
  const double n[2] = {hp[h].nx, hp[h].ny};
  double a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
  if (R.mode == 1) {  // collide_static_at(..., R.px, R.py), hyperplane h
    for (int i = 0; i < 2; i++) {
      a += n[i] * cs[i][0];
      b += n[i] * cs[i][1];
      c += n[i] * cs[i][2];
      d += n[i] * cs[i][3];
      e += n[i] * cs[i][4];
      f += n[i] * cs[i][5];
    }
    a = div_nz(a, 120.0); b = div_nz(b, 24.0); c = div_nz(c, 6.0); d *= 0.5;  // (x / 2 == x * 0.5 exactly; e /= 1.0 is the identity)
    {
      double sm = 0.0;
      sm += n[0] * (hp[h].px + R.px);
      sm += n[1] * (hp[h].py + R.py);
      f -= sm;
    }
    double ts[GEN ? POLY_MAX_ROOTS : 2];
    const int nr = solve_poly<GEN>(a, b, c, d, e, f, ts);
    if (nr < 0) return -1;
    for (int r = 0; r < nr; r++) {
      const double it = ts[r];
      if (it >= 0 && it <= prT) {
        const double wx = pp_p_auto(cs[0], it), wy = pp_p_auto(cs[1], it);
        if (poly_inside(hp, o.n_hp, wx - R.px, wy - R.py)) return 1;
      }
    }
    return 0;
  }
  if (R.mode == 2) {  // collide_linear, hyperplane h
    const double cov_v[2] = {o.v[0] + n[0] * o.cov_v, o.v[1] + n[1] * o.cov_v};
    for (int i = 0; i < 2; i++) {
      a += n[i] * cs[i][0];
      b += n[i] * cs[i][1];
      c += n[i] * cs[i][2];
      d += n[i] * cs[i][3];
      e += n[i] * cs[i][4];
      f += n[i] * cs[i][5];
    }
    a = div_nz(a, 120.0); b = div_nz(b, 24.0); c = div_nz(c, 6.0); d *= 0.5;
    {
      double sm = 0.0;
      sm += n[0] * cov_v[0];
      sm += n[1] * cov_v[1];
      e -= sm;
      double s2 = 0.0;
      s2 += n[0] * ((hp[h].px + o.p[0]) + cov_v[0] * t);
      s2 += n[1] * ((hp[h].py + o.p[1]) + cov_v[1] * t);
      f -= s2;
    }
    double ts[GEN ? POLY_MAX_ROOTS : 2];
    const int nr = solve_poly<GEN>(a, b, c, d, e, f, ts);
    if (nr < 0) return -1;
    for (int r = 0; r < nr; r++) {
      const double it = ts[r];
      if (it >= 0 && it <= prT) {
        const double wx = pp_p_auto(cs[0], it), wy = pp_p_auto(cs[1], it);
        if (obs_inside_linear(D, o, wx, wy, it + t)) return 1;
      }
    }
    return 0;
  }
  // collide_nonlinear: segment slot s, hyperplane h
  const PolySlot &S = R.slot[s];
  const double hpp[2] = {hp[h].px, hp[h].py};
  for (int i = 0; i < 2; i++) {
    a += n[i] * cs[i][0];
    b += n[i] * cs[i][1];
    c += n[i] * cs[i][2] - n[i] * S.wj[i];
    d += n[i] * cs[i][3] - n[i] * S.wa[i];
    e += n[i] * cs[i][4] - n[i] * S.wv[i];
    f += n[i] * cs[i][5] - n[i] * (hpp[i] + S.wp[i]);
  }
  a = div_nz(a, 120.0); b = div_nz(b, 24.0); c = div_nz(c, 6.0); d *= 0.5;
  double ts[GEN ? POLY_MAX_ROOTS : 2];
  const int nr = solve_poly<GEN>(a, b, c, d, e, f, ts);
  if (nr < 0) return -1;
  for (int r = 0; r < nr; r++) {
    const double it = ts[r];
    if (it >= S.t_residual && it <= prT && S.T + S.seg_T >= it + S.start_t && S.T <= it + S.start_t) {
      const double cx = pp_p_auto(cs[0], it), cy = pp_p_auto(cs[1], it);
      if (obs_inside_nonlinear_pos(D, o, cx, cy, it + t, R.id0)) return 1;
    }
  }
  return 0;
}

#ifdef __HIPCC__
// isFree(pr, t_rel) of every valid primitive (cs[i], i < n_u) against every obstacle of world W, the workgroup's lanes
// spread over (primitive, obstacle, segment slot, hyperplane); ORs into hit[i], sets *unsupported.  LDS scratch:
// prep[POLY_MAX_OBS], hit_idx / uns_idx[POLY_MAX_U * POLY_MAX_OBS], hp_max.  Every thread of the workgroup must call.
// cache_q: query + 1 (tags the per-level cache entries); mid_hook(): called by every thread once the obstacles are
// prepared, before the (long) item loop -- the caller can put memory traffic of its own in flight there
struct PolyNoHook { __device__ __forceinline__ void operator()() const {} };
template <int BLOCK, class Hook = PolyNoHook, bool GEN = false>
__device__ __forceinline__ void poly_collide_all(const PolyDev &D, const PolyWorld &W, const double (*cs)[2][6], const int32_t *valid, int n_u, double T, double t_rel,
                                                 PolyPrep *prep, uint32_t *hit_idx, uint32_t *uns_idx, int32_t *hp_max, int32_t *hit, int32_t *unsupported, int32_t *start_hit,
                                                 int tid, long long cache_q, Hook mid_hook, unsigned long long *cyc = nullptr, unsigned long long *lds_level = nullptr) {
  // lds_level (LDS, 2 words, or null): the (query, time) the entries of prep[] were prepared for by an earlier call of
  // this workgroup -- an expansion at the same time level re-uses them where they lie
  const int n_obs = W.n_obs;
  unsigned long long tc0 = __builtin_readcyclecounter();
  const double x0 = pp_p_auto(cs[0][0], 0.0), y0 = pp_p_auto(cs[0][1], 0.0);  // pr.evaluate(0): the node position, whatever the primitive
  if (n_obs > POLY_MAX_OBS) {  // (uniform) a crowded world: one lane per pair, the reference's loops as they are
    for (int j = tid; j < n_obs; j += BLOCK)
      if (obs_point_hits(D, D.obs[W.obs_off + j], x0, y0, t_rel)) *start_hit = 1;
    const int pairs = n_u * n_obs;
    for (int e = tid; e < pairs; e += BLOCK) {
      const int i = e / n_obs, j = e % n_obs;
      if (!valid[i]) continue;
      double c[2][6];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) c[a][b] = cs[i][a][b];
      const int r = obs_prim_hits<GEN>(D, c, T, D.obs[W.obs_off + j], t_rel);
      if (r < 0) *unsupported = 1;
      if (r > 0) hit[i] = 1;
    }
    __syncthreads();
    mid_hook();
    return;
  }
  // hit_idx[n_pairs .. ] doubles as the list of pairs that survive the pruning (u32 each), uns_idx[n_pairs] as its length
  const int n_pairs = n_u * n_obs;
  if (tid == 0) { *hp_max = 1; uns_idx[POLY_MAX_U * POLY_MAX_OBS - 1] = 0u; }
  for (int e = tid; e < n_pairs; e += BLOCK) { hit_idx[e] = 0xFFFFFFFFu; uns_idx[e] = 0xFFFFFFFFu; }
  // the time-dependent part comes from the per-level cache when this (query, time) has been prepared before: the level's
  // entries lie contiguously in HBM / L2 and are copied by the whole workgroup in 16-byte words (one lane copying its own
  // 368-byte entry word by word cost most of this phase); a level that is already in LDS is not copied at all
  const unsigned long long tbits = (unsigned long long)__double_as_longlong(t_rel);
  PolyPrep *level = nullptr;
  if (D.prep_cache && T > 0 && cache_q != 0) {  // (uniform)
    const double lv = t_rel / T;
    const int k = lv >= 0 && lv < (double)POLY_CACHE_LEVELS ? (int)(lv + 0.5) : -1;
    if (k >= 0 && k < POLY_CACHE_LEVELS) level = D.prep_cache + ((size_t)(D.prep_slice0 + (int)blockIdx.x) * POLY_CACHE_LEVELS + (size_t)k) * POLY_MAX_OBS;
  }
  const bool in_lds = lds_level && cache_q != 0 && lds_level[0] == (unsigned long long)cache_q && lds_level[1] == tbits;  // (uniform: written behind a barrier)
  if (level && !in_lds) {
    static_assert(sizeof(PolyPrep) % 16 == 0, "16-byte words");
    const uint4 *src = (const uint4 *)level;
    uint4 *dst = (uint4 *)prep;
    const int nw = n_obs * (int)(sizeof(PolyPrep) / 16);
    for (int i = tid; i < nw; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  if (tid < n_obs) {
    const PolyObs &o = D.obs[W.obs_off + tid];
    PolyPrep &R = prep[tid];
    if (!((level || in_lds) && R.tag_q == cache_q && R.tag_t == tbits)) {
      poly_prepare_time(D, o, T, t_rel, R);
      R.tag_t = tbits;
      R.tag_q = cache_q;
      if (level) level[tid] = R;
    }
    if (poly_start_test(D, o, R, x0, y0, t_rel)) *start_hit = 1;
    atomicMax(hp_max, o.n_hp);
  }
  __syncthreads();
  if (lds_level && tid == 0) { lds_level[0] = (unsigned long long)cache_q; lds_level[1] = tbits; }  // (read by the NEXT call, behind its barriers)
  mid_hook();
  if (cyc && tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); cyc[5] += now - tc0; tc0 = now; }
  // pruning: a primitive stays within |v| T + |u| T^2 / 2 (per axis) of the node, an obstacle within its reach of
  // (cx, cy); further apart than the two together (plus a margin far above rounding) they cannot meet.  Conservative:
  // a pruned pair is one whose collide() returns false.  [16 robots in a 10 m arena: most pairs]
  uint32_t *plist = hit_idx + POLY_MAX_U * POLY_MAX_OBS / 2, *plen = &uns_idx[POLY_MAX_U * POLY_MAX_OBS - 1];
  const bool can_list = n_pairs <= POLY_MAX_U * POLY_MAX_OBS / 2 - 1;
  for (int pair = tid; pair < n_pairs; pair += BLOCK) {
    const int i = pair / n_obs, j = pair - i * n_obs;
    if (!valid[i] || prep[j].mode == 0) continue;
    bool keep = true;
    const PolyPrep &R = prep[j];
    if (R.reach < INFINITY && lead_pzero(cs[i][0]) && lead_pzero(cs[i][1])) {
      const double rx = fabs(cs[i][0][4]) * T + 0.5 * fabs(cs[i][0][3]) * T * T + R.reach + POLY_PRUNE_EPS;
      const double ry = fabs(cs[i][1][4]) * T + 0.5 * fabs(cs[i][1][3]) * T * T + R.reach + POLY_PRUNE_EPS;
      keep = !(fabs(x0 - R.cx) > rx || fabs(y0 - R.cy) > ry);
    }
    if (keep && can_list) plist[atomicAdd(plen, 1u)] = (uint32_t)pair;
  }
  __syncthreads();
  // items: one per (surviving pair, segment slot, hyperplane) that exists.  The surviving pairs first write their items
  // into a dense list (upper half of uns_idx, which only uses its first n_pairs entries and its last one): a pair owns
  // n_slots x n_hp of them (1 x n_hp for a static / linear obstacle, one for the serial fall-back) -- typically 4 to 8
  // instead of the POLY_SLOTS x HP = 12 (padded to 16) of a rectangular index space, i.e. fewer rounds of the lanes.
  // A list that does not fit (uniform) falls back to the rectangular space.
  const int HP = *hp_max;
  const int n_list = can_list ? (int)*plen : n_pairs;
  uint32_t *ilist = uns_idx + POLY_MAX_U * POLY_MAX_OBS / 2, *ilen = &uns_idx[POLY_MAX_U * POLY_MAX_OBS - 2];
  constexpr int ICAP = POLY_MAX_U * POLY_MAX_OBS / 2 - 2;
  const bool dense = can_list && n_pairs <= POLY_MAX_U * POLY_MAX_OBS / 2 && POLY_SLOTS <= 15 && HP <= 15;
  if (dense) {
    if (tid == 0) *ilen = 0u;
    __syncthreads();
    for (int li = tid; li < n_list; li += BLOCK) {
      const int pair = (int)plist[li];
      const int i = pair / n_obs, j = pair - i * n_obs;
      if (!valid[i]) continue;
      const PolyPrep &R = prep[j];
      const int mode = R.mode, nh = D.obs[W.obs_off + j].n_hp;
      const int ns_ = mode == 4 ? 1 : mode == 3 ? R.n_slots : 1, nh_ = mode == 4 ? 1 : nh;
      const int cnt = mode == 0 ? 0 : ns_ * nh_;
      if (cnt == 0) continue;
      const uint32_t base = atomicAdd(ilen, (uint32_t)cnt);
      if (base + (uint32_t)cnt <= (uint32_t)ICAP)
        for (int s_ = 0; s_ < ns_; s_++)
          for (int h_ = 0; h_ < nh_; h_++) ilist[base + (uint32_t)(s_ * nh_ + h_)] = ((uint32_t)li << 8) | ((uint32_t)s_ << 4) | (uint32_t)h_;
    }
    __syncthreads();
  }
  const bool use_list = dense && *ilen <= (uint32_t)ICAP;  // (uniform)
  int sub_log = 0;
  while ((1 << sub_log) < POLY_SLOTS * HP) sub_log++;
  const int total = use_list ? (int)*ilen : (n_list << sub_log);
  for (int e = tid; e < total; e += BLOCK) {
    int li, s, h;
    bool first;
    if (use_list) {
      const uint32_t it = ilist[e];
      li = (int)(it >> 8); s = (int)((it >> 4) & 15u); h = (int)(it & 15u);
      first = s == 0 && h == 0;
    } else {
      li = e >> sub_log;
      const int sub = e & ((1 << sub_log) - 1);
      if (sub >= POLY_SLOTS * HP) continue;
      s = sub / HP; h = sub - s * HP;
      first = sub == 0;
    }
    const int pair = can_list ? (int)plist[li] : li;
    const int i = pair / n_obs, j = pair - i * n_obs;
    if (!valid[i]) continue;
    const PolyPrep &R = prep[j];
    const PolyObs &o = D.obs[W.obs_off + j];
    const int mode = R.mode;
    if (mode == 0) continue;
    if (mode == 4) { if (!first) continue; }
    else if (h >= o.n_hp || (mode == 3 ? s >= R.n_slots : s != 0)) continue;
    const uint32_t idx = (uint32_t)(s * HP + h);
    if (hit_idx[pair] < idx) continue;  // the reference's loop has already returned (a benign race: only saves work)
    int r;
    if (mode == 4) {
      double c[2][6];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) c[a][b] = cs[i][a][b];
      r = obs_prim_hits<GEN>(D, c, T, o, t_rel);
    } else {
      r = poly_item<GEN>(D, cs[i], T, o, R, s, h, t_rel);
    }
    if (r > 0) atomicMin(&hit_idx[pair], idx);
    if (r < 0) atomicMin(&uns_idx[pair], idx);
  }
  if (cyc && tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); cyc[6] += now - tc0; tc0 = now; }
  __syncthreads();
  if (cyc && tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); cyc[7] += now - tc0; tc0 = now; }
  for (int e = tid; e < n_pairs; e += BLOCK) {
    const uint32_t hi = hit_idx[e], ui = uns_idx[e];
    if (hi != 0xFFFFFFFFu && hi < ui) hit[e / n_obs] = 1;   // collide() returned true before meeting an unsupported degree
    else if (ui != 0xFFFFFFFFu) *unsupported = 1;
  }
  __syncthreads();
}
#endif

#ifdef __HIPCC__
// The obstacle set of one world staged in LDS for the duration of a search (every traj_eval / collide() walks the
// trajectory's segments: from global memory that is a chain of dependent L2 round trips per call -- 7-10 k cycles per
// expansion before staging).  Worlds that do not fit keep reading global memory.
constexpr int POLY_LDS_HPS = 256, POLY_LDS_SEGS = 256;
struct PolyWorldLds {
  PolyObs obs[POLY_MAX_OBS];
  PolyHP hps[POLY_LDS_HPS];
  PolySeg segs[POLY_LDS_SEGS];
  double cum[POLY_LDS_SEGS + POLY_MAX_OBS];
  int32_t ok, n_hps, n_segs, pad;
};
// every thread of the workgroup calls; returns the PolyDev / PolyWorld to use (LDS-backed when the world fits)
template <int BLOCK>
__device__ __forceinline__ void poly_stage_world(const PolyDev &D, const PolyWorld &W, PolyWorldLds &L, int tid, PolyDev &DL, PolyWorld &WL) {
  DL = D;
  WL = W;
  if (tid == 0) {
    int nh = 0, ns = 0;
    bool fits = W.n_obs <= POLY_MAX_OBS;
    for (int j = 0; j < W.n_obs && fits; j++) {
      const PolyObs &o = D.obs[W.obs_off + j];
      nh += o.n_hp;
      ns += o.n_seg;
      fits = nh <= POLY_LDS_HPS && ns <= POLY_LDS_SEGS;
    }
    L.ok = fits ? 1 : 0;
    if (fits) {  // per obstacle: its offsets inside the LDS arrays, its segment start times
      int hoff = 0, soff = 0, coff = 0;
      for (int j = 0; j < W.n_obs; j++) {
        PolyObs o = D.obs[W.obs_off + j];
        const int gs = o.seg_off;
        o.hp_off = hoff; o.seg_off = soff; o.cum_off = coff;
        double t0 = 0.0;
        L.cum[coff] = 0.0;
        for (int k = 0; k < o.n_seg; k++) {
          t0 = D.segs[gs + k].T + t0;
          L.cum[coff + k + 1] = t0;
        }
        L.obs[j] = o;
        hoff += o.n_hp; soff += o.n_seg; coff += o.n_seg + 1;
      }
      L.n_hps = hoff;
      L.n_segs = soff;
    }
  }
  __syncthreads();
  if (!L.ok) return;
  // copy hyperplanes and segments (global offsets re-read from the global records: obstacle j's range)
  for (int j = 0; j < W.n_obs; j++) {
    const PolyObs &g = D.obs[W.obs_off + j];
    const PolyObs &l = L.obs[j];
    for (int k = tid; k < g.n_hp; k += BLOCK) L.hps[l.hp_off + k] = D.hps[g.hp_off + k];
    for (int k = tid; k < g.n_seg; k += BLOCK) L.segs[l.seg_off + k] = D.segs[g.seg_off + k];
  }
  __syncthreads();
  DL.obs = L.obs;
  DL.hps = L.hps;
  DL.segs = L.segs;
  DL.cum = L.cum;
  WL.obs_off = 0;
}
#endif

// Primitive<2>(curr, u, dt) coefficients (mpl_shim primitive.h)
MPLX_HD void poly_prim_build(int control, const double pos[2], const double vel[2], const double u[2], double cs[2][6], const double *acc = nullptr, const double *jrk = nullptr) {
  for (int i = 0; i < 2; i++) {
    for (int k = 0; k < 6; k++) cs[i][k] = 0.0;
    const int kind = control & 15;
    if (kind == CTRL_VEL) { cs[i][4] = u[i]; cs[i][5] = pos[i]; }
    else if (kind == CTRL_ACC) { cs[i][3] = u[i]; cs[i][4] = vel[i]; cs[i][5] = pos[i]; }
    else if (kind == CTRL_JRK) { cs[i][2] = u[i]; cs[i][3] = acc ? acc[i] : 0.0; cs[i][4] = vel[i]; cs[i][5] = pos[i]; }
    else { cs[i][1] = u[i]; cs[i][2] = jrk ? jrk[i] : 0.0; cs[i][3] = acc ? acc[i] : 0.0; cs[i][4] = vel[i]; cs[i][5] = pos[i]; }
  }
}
// validate_primitive for VEL / ACC (mpl_shim primitive.h): ACC checks max |vel| per axis against v_max > 0; the
// velocity of such a primitive is monotone, so its extrema are the end points
// JRK / SNP: the general statement (max_vel / max_acc / max_jrk with their interior extrema)
MPLX_HD bool poly_validate(int control, const double cs[2][6], double T, double v_max, double a_max = -1.0, double j_max = -1.0) {
  const int kind = control & 15;
  if (kind == CTRL_VEL) return true;
  if (kind == CTRL_ACC) {
    for (int i = 0; i < 2; i++) {
      const double m = fmax(fabs(pp_v_auto(cs[i], 0.0)), fabs(pp_v_auto(cs[i], T)));
      if (v_max > 0 && m > v_max) return false;
    }
    return true;
  }
  for (int i = 0; i < 2; i++)
    if (v_max > 0 && poly_max_abs(cs[i], 1, T) > v_max) return false;
  for (int i = 0; i < 2; i++)
    if (a_max > 0 && poly_max_abs(cs[i], 2, T) > a_max) return false;
  if (kind == CTRL_SNP)
    for (int i = 0; i < 2; i++)
      if (j_max > 0 && poly_max_abs(cs[i], 3, T) > j_max) return false;
  return true;
}
// env_poly_map::calculate_intrinsic_cost: pr.J(pr.control()) + 0.001 * pr.J(Control::VEL) + w dt (env_poly_map.h:71-73)
MPLX_HD double poly_intrinsic_cost(int control, const double cs[2][6], double T, double w, double dt) {
  if ((control & 15) == CTRL_ACC && lead_pzero(cs[0]) && lead_pzero(cs[1])) {
    // Primitive1D::J's double sum with the structurally zero coefficients of an ACC primitive taken out: the dropped
    // terms are +-0.0 added to a non-negative running sum, i.e. no-ops.  J(ACC): q = {c3, 0, 0, 0}: the (0, 0) term;
    // J(VEL): q = {c4, c3, 0, 0, 0}: terms (0,0), (0,1), (1,0), (1,1) in that order.
    double jc = 0, jv = 0;
    for (int k = 0; k < 2; k++) {
      const double q0 = cs[k][3] / 1.0;
      double s = 0.0;
      s += q0 * q0 * T / 1.0;
      jc += s;
    }
    for (int k = 0; k < 2; k++) {
      const double q0 = cs[k][4] / 1.0, q1 = cs[k][3] / 1.0;
      const double t2 = T * T, t3 = t2 * T;
      double s = 0.0;
      s += q0 * q0 * T / 1.0;
      s += q0 * q1 * t2 / 2.0;
      s += q1 * q0 * t2 / 2.0;
      s += q1 * q1 * t3 / 3.0;
      jv += s;
    }
    return jc + 0.001 * jv + w * dt;
  }
  double jc = 0;
  for (int k = 0; k < 2; k++) jc += pp_J(cs[k], T, control);
  double jv = 0;
  for (int k = 0; k < 2; k++) jv += pp_J(cs[k], T, CTRL_VEL);
  return jc + 0.001 * jv + w * dt;
}

}  // namespace mplx
