/*
 * mpl_oracle.c -- CPU restatement (plain C99) of the MPL v1.2 voxel-map A* path.
 * TEST INFRASTRUCTURE ONLY -- see mpl_oracle.h for scope, citations and the PARITY UNPINNED note.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math  (no FMA contraction: the HIP side is
 * compiled the same way so f64 results are bit-comparable).  Only + - * / sqrt round ceil fabs
 * are used on values that reach results, so host libm and device ocml cannot disagree.
 *
 * Tags:  [IN-TREE file:line]  = follows code visible under /root/reference
 *        [UNVERIFIED]         = recollection of upstream motion_primitive_library
 *        [DEVIATION]          = deliberate, documented difference from the recollected upstream
 */
#include "mpl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ constants [UNVERIFIED] */
#define ORC_HASH_RES_POS 0.01 /* waypoint.h hash_value: round(pos/0.01) */
#define ORC_HASH_RES_VEL 0.1
#define ORC_HASH_RES_ACC 0.1
#define ORC_HASH_RES_JRK 0.1
#define ORC_HASH_RES_T 0.1
#define ORC_HASH_RES_YAW 0.1

/* ------------------------------------------------------------------ small helpers */
/* [UNVERIFIED Q1 of mpl_oracle.h] upstream is recalled to call std::pow(t, n) here; equal at t = 1, possibly one ulp apart at
 * the interior sample times of cubic / quartic terms */
static double pw(double t, int n) { /* t^n by repeated multiplication, left to right */
  double r = t;
  for (int i = 1; i < n; i++) r = r * t;
  return r;
}

/* ---- audit of open question Q1 (mpl_oracle.h): the recollected upstream form of Primitive1D::p raises t to the powers >= 3 with
 * std::pow.  With the audit on, every collision sample of is_free(pr) is evaluated BOTH ways and the cells compared: a search whose
 * count of differing cells is 0 does not depend on the question.  (Global, not per planner: an audit is a single-threaded test run.) */
static int g_q1_on = 0;
static uint64_t g_q1_samples = 0, g_q1_pos_bits = 0, g_q1_cells = 0;
void orc_q1_audit(int on) { g_q1_on = on; g_q1_samples = g_q1_pos_bits = g_q1_cells = 0; }
void orc_q1_counts(uint64_t out[3]) { out[0] = g_q1_samples; out[1] = g_q1_pos_bits; out[2] = g_q1_cells; }
static double p1_p_libm_pow(const double *c, double t) { /* [UNVERIFIED Q1] powers >= 3 through libm's pow, the square as t * t */
  return c[0] / 120 * pow(t, 5) + c[1] / 24 * pow(t, 4) + c[2] / 6 * pow(t, 3) + c[3] / 2 * t * t + c[4] * t + c[5];
}

/* Primitive1D p/v/a/j  [IN-TREE primitive_geometry_utils.h:12-26 convention] */
static double p1_p(const double *c, double t) {
  return c[0] / 120 * pw(t, 5) + c[1] / 24 * pw(t, 4) + c[2] / 6 * pw(t, 3) + c[3] / 2 * t * t + c[4] * t + c[5];
}
static double p1_v(const double *c, double t) {
  return c[0] / 24 * pw(t, 4) + c[1] / 6 * pw(t, 3) + c[2] / 2 * t * t + c[3] * t + c[4];
}
static double p1_a(const double *c, double t) { return c[0] / 6 * pw(t, 3) + c[1] / 2 * t * t + c[2] * t + c[3]; }
static double p1_j(const double *c, double t) { return c[0] / 2 * t * t + c[1] * t + c[2]; }

/* ------------------------------------------------------------------ polynomial root finding
 * [DEVIATION D1] upstream math.h uses Cardano / Ferrari closed forms (acos, cos, cbrt) for degree 3-4
 * and Eigen companion-matrix eigenvalues for degree 5-6.  Neither is bit-reproducible between a
 * host libm and device code, so degree >= 3 uses a deterministic derivative-chain isolation with
 * a safeguarded Newton/bisection iteration built from + - * / only.  Degree <= 2 keeps the
 * upstream formulas. */
static double poly_eval(const double *a, int n, double x) {
  double r = a[n];
  for (int i = n - 1; i >= 0; i--) r = r * x + a[i];
  return r;
}
static void poly_eval2(const double *a, int n, double x, double *f, double *df) {
  double r = a[n], d = 0.0;
  for (int i = n - 1; i >= 0; i--) {
    d = d * x + r;
    r = r * x + a[i];
  }
  *f = r;
  *df = d;
}
/* root of a (degree n) bracketed by [x1,x2] with f(x1)*f(x2) < 0 */
static double poly_refine(const double *a, int n, double x1, double f1, double x2, double f2) {
  double xl, xh;
  (void)f2;
  if (f1 < 0.0) {
    xl = x1;
    xh = x2;
  } else {
    xl = x2;
    xh = x1;
  }
  double x = 0.5 * (x1 + x2);
  double dxold = fabs(x2 - x1), dx = dxold, f, df;
  poly_eval2(a, n, x, &f, &df);
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
    poly_eval2(a, n, x, &f, &df);
    if (f == 0.0) return x;
    if (f < 0.0)
      xl = x;
    else
      xh = x;
  }
  return x;
}
/* real roots of degree-n polynomial (a[n] != 0) inside (lo, hi), ascending */
static int poly_roots_in(const double *a, int n, double lo, double hi, double *roots) {
  if (n <= 0) return 0;
  if (n == 1) {
    double r = -a[0] / a[1];
    if (r > lo && r < hi) {
      roots[0] = r;
      return 1;
    }
    return 0;
  }
  double d[8], crit[8];
  for (int i = 1; i <= n; i++) d[i - 1] = a[i] * (double)i;
  int nd = n - 1;
  while (nd > 0 && d[nd] == 0.0) nd--;
  int nc = poly_roots_in(d, nd, lo, hi, crit);
  int nr = 0;
  double x0 = lo, f0 = poly_eval(a, n, lo);
  for (int k = 0; k <= nc; k++) {
    double x1 = (k < nc) ? crit[k] : hi;
    double f1 = poly_eval(a, n, x1);
    if (f1 == 0.0) {
      if (k < nc) roots[nr++] = x1; /* root exactly on a critical point; hi itself is excluded */
    } else if (f0 != 0.0 && ((f0 < 0.0) != (f1 < 0.0))) {
      roots[nr++] = poly_refine(a, n, x0, f0, x1, f1);
    }
    x0 = x1;
    f0 = f1;
  }
  return nr;
}
int orc_poly_roots_above(const double *a_in, int n_in, double lo, double *roots) {
  double a[8];
  int n = n_in;
  for (int i = 0; i <= n; i++) a[i] = a_in[i];
  while (n > 0 && a[n] == 0.0) n--;
  if (n == 0) return 0;
  /* bound on |root|: Cauchy's 1 + max|a_i/a_n|; when the coefficient below the leading one is zero (true
   * of both heuristic polynomials) |x| >= 1 gives |sum_{i<n} a_i x^i| <= S |x|^(n-2) with S = sum|a_i|,
   * so every root has |x| <= max(1, sqrt(S/|a_n|)) -- a far shorter bracket, fewer iterations. */
  double m = 0.0;
  if (n >= 2 && a[n - 1] == 0.0) {
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum += fabs(a[i]);
    m = sqrt(sum / fabs(a[n]));
  } else {
    for (int i = 0; i < n; i++) {
      double q = fabs(a[i] / a[n]);
      if (q > m) m = q;
    }
  }
  double hi = 1.0 + m;
  if (!(hi > lo)) return 0;
  return poly_roots_in(a, n, lo, hi, roots);
}

/* upstream solve(0, b, c, d, e) restricted to what control-built primitives need: the leading
 * cubic coefficient b is c0/6 == 0 for every primitive built from (state, control), leaving
 * quad / linear.  [UNVERIFIED math.h quad(): roots (-c -+ sqrt(c^2-4bd)) / (2b), in that order] */
static int solve_upto_cubic(double b, double c, double d, double e, double *r) {
  if (b != 0.0) { /* general cubic (never reached by control-built primitives) */
    double a[4] = {e, d, c, b};
    double m = fabs(e / b);
    if (fabs(d / b) > m) m = fabs(d / b);
    if (fabs(c / b) > m) m = fabs(c / b);
    return poly_roots_in(a, 3, -(1.0 + m), 1.0 + m, r);
  }
  if (c != 0.0) {
    double p = d * d - 4 * c * e;
    if (p < 0) return 0;
    r[0] = (-d - sqrt(p)) / (2 * c);
    r[1] = (-d + sqrt(p)) / (2 * c);
    return 2;
  }
  if (d != 0.0) {
    r[0] = -e / d;
    return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ Primitive (a3, a4) */
/* [UNVERIFIED primitive.h ctor] VEL c=[0,0,0,0,u,p]; ACC [0,0,0,u,v,p]; JRK [0,0,u,a,v,p];
 * SNP [0,u,j,a,v,p] -- follows from the in-tree convention (c2<->jrk0, c3<->acc0, c4<->vel0,
 * c5<->pos0: primitive_geometry_utils.h:135-146). */
void orc_primitive_build(const orc_waypoint *p, const double *u, double dt, orc_primitive *out) {
  memset(out, 0, sizeof(*out));
  out->t = dt;
  out->control = p->control;
  out->cyaw[5] = p->yaw; /* (no yaw input: the yaw stays) */
  for (int i = 0; i < 3; i++) {
    double *c = out->c[i];
    switch (p->control & 15) {
      case ORC_VEL: c[4] = u[i]; c[5] = p->pos[i]; break;
      case ORC_ACC: c[3] = u[i]; c[4] = p->vel[i]; c[5] = p->pos[i]; break;
      case ORC_JRK: c[2] = u[i]; c[3] = p->acc[i]; c[4] = p->vel[i]; c[5] = p->pos[i]; break;
      case ORC_SNP: c[1] = u[i]; c[2] = p->jrk[i]; c[3] = p->acc[i]; c[4] = p->vel[i]; c[5] = p->pos[i]; break;
      default: break;
    }
  }
}
/* [UNVERIFIED primitive.h ctor with a Vec4f input] the yaw channel is a VEL-type Primitive1D: yaw(t) = yaw0 + u_yaw t */
void orc_primitive_build_yaw(const orc_waypoint *p, const double *u, double u_yaw, double dt, orc_primitive *out) {
  orc_primitive_build(p, u, dt, out);
  out->cyaw[4] = u_yaw;
}
/* deterministic sin / cos and angle normalisation: the same sequences of + - * as mplx_math.h (the decision of
 * validate_yaw must fall the same way on host and device; libm and the device library may differ in the last bit) */
static void det_sincos(double x, double *sn, double *cs) {
  const double two_over_pi = 0.63661977236758138, pio2_hi = 1.5707963267948966, pio2_lo = 6.123233995736766e-17;
  const double kf = round(x * two_over_pi);
  const double r = (x - kf * pio2_hi) - kf * pio2_lo;
  const double r2 = r * r;
  double ps = 1.0 - r2 / 210.0;
  ps = 1.0 - r2 / 156.0 * ps;
  ps = 1.0 - r2 / 110.0 * ps;
  ps = 1.0 - r2 / 72.0 * ps;
  ps = 1.0 - r2 / 42.0 * ps;
  ps = 1.0 - r2 / 20.0 * ps;
  ps = 1.0 - r2 / 6.0 * ps;
  const double s = r * ps;
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
void orc_det_sincos(double x, double *sn, double *cs) { det_sincos(x, sn, cs); } /* (exported for the known-answer test) */
/* [UNVERIFIED angle normalisation of Primitive::evaluate] into [-pi, pi] by steps of 2 pi */
static double normalize_yaw(double q) {
  const double pi = 3.141592653589793;
  while (q > pi) q -= 2.0 * pi;
  while (q < -pi) q += 2.0 * pi;
  return q;
}
void orc_primitive_evaluate(const orc_primitive *pr, double t, orc_waypoint *out) {
  memset(out, 0, sizeof(*out));
  out->control = pr->control;
  for (int k = 0; k < 3; k++) {
    out->pos[k] = p1_p(pr->c[k], t);
    out->vel[k] = p1_v(pr->c[k], t);
    out->acc[k] = p1_a(pr->c[k], t);
    out->jrk[k] = p1_j(pr->c[k], t);
  }
  if (pr->control & ORC_YAW) out->yaw = normalize_yaw(p1_p(pr->cyaw, t));
}
/* [UNVERIFIED primitive.h validate_yaw(pr, my)] at both ends of the primitive, when the planar velocity is not zero,
 * its direction must lie within my of the yaw direction: v_hat . (cos yaw, sin yaw) >= cos(my).  my <= 0: no check. */
int orc_validate_yaw(const orc_primitive *pr, double my) {
  if (my <= 0) return 1;
  double sm, cmax;
  det_sincos(my, &sm, &cmax);
  const double ts[2] = {0.0, pr->t};
  for (int e = 0; e < 2; e++) {
    orc_waypoint w;
    orc_primitive_evaluate(pr, ts[e], &w);
    const double vx = w.vel[0], vy = w.vel[1];
    if (vx == 0.0 && vy == 0.0) continue;
    const double n = sqrt(vx * vx + vy * vy);
    double sn, cs;
    det_sincos(w.yaw, &sn, &cs);
    const double d = vx / n * cs + vy / n * sn;
    if (d < cmax) return 0;
  }
  return 1;
}

/* [UNVERIFIED primitive.h extrema_* / max_*]  extrema of v on (0,t) = roots of a(t); the loop
 * stops at the first root >= t (upstream `else if (it >= t) break;`). */
static int extrema_filter(const double *roots, int n, double t, double *ts) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (roots[i] > 0 && roots[i] < t)
      ts[m++] = roots[i];
    else if (roots[i] >= t)
      break;
  }
  return m;
}
double orc_primitive_max_vel(const orc_primitive *pr, int k) {
  const double *c = pr->c[k];
  double r[4], ts[4];
  int n = solve_upto_cubic(c[0] / 6, c[1] / 2, c[2], c[3], r);
  int m = extrema_filter(r, n, pr->t, ts);
  double mx = fmax(fabs(p1_v(c, 0)), fabs(p1_v(c, pr->t)));
  for (int i = 0; i < m; i++) {
    double v = fabs(p1_v(c, ts[i]));
    mx = v > mx ? v : mx;
  }
  return mx;
}
double orc_primitive_max_acc(const orc_primitive *pr, int k) {
  const double *c = pr->c[k];
  double r[4], ts[4];
  int n = solve_upto_cubic(0, c[0] / 2, c[1], c[2], r);
  int m = extrema_filter(r, n, pr->t, ts);
  double mx = fmax(fabs(p1_a(c, 0)), fabs(p1_a(c, pr->t)));
  for (int i = 0; i < m; i++) {
    double v = fabs(p1_a(c, ts[i]));
    mx = v > mx ? v : mx;
  }
  return mx;
}
double orc_primitive_max_jrk(const orc_primitive *pr, int k) {
  const double *c = pr->c[k];
  double r[4], ts[4];
  int n = solve_upto_cubic(0, 0, c[0], c[1], r);
  int m = extrema_filter(r, n, pr->t, ts);
  double mx = fmax(fabs(p1_j(c, 0)), fabs(p1_j(c, pr->t)));
  for (int i = 0; i < m; i++) {
    double v = fabs(p1_j(c, ts[i]));
    mx = v > mx ? v : mx;
  }
  return mx;
}

/* [UNVERIFIED primitive.h validate_primitive / validate_xxx]  ACC checks vel; JRK checks vel,acc;
 * SNP checks vel,acc,jrk; VEL checks nothing.  A limit <= 0 disables its check. */
int orc_validate_primitive(const orc_primitive *pr, double mv, double ma, double mj) {
  const int ctl = pr->control & 15;
  int chk_v = ctl == ORC_ACC || ctl == ORC_JRK || ctl == ORC_SNP;
  int chk_a = ctl == ORC_JRK || ctl == ORC_SNP;
  int chk_j = ctl == ORC_SNP;
  if (chk_v)
    for (int i = 0; i < 3; i++)
      if (mv > 0 && orc_primitive_max_vel(pr, i) > mv) return 0;
  if (chk_a)
    for (int i = 0; i < 3; i++)
      if (ma > 0 && orc_primitive_max_acc(pr, i) > ma) return 0;
  if (chk_j)
    for (int i = 0; i < 3; i++)
      if (mj > 0 && orc_primitive_max_jrk(pr, i) > mj) return 0;
  return 1;
}

/* J(control) = integral over [0,t] of the squared k-th derivative, summed over axes (a9).
 * [DEVIATION D2] upstream expands the integral into a fixed closed form; here it is the double sum
 * sum_i sum_j q_i q_j t^(i+j+1)/(i+j+1) over the monomial coefficients q of the derivative, in
 * ascending (i,j) order.  Same value analytically (ACC control: u^2 t). */
double orc_primitive_J(const orc_primitive *pr, int control) {
  control &= 15; /* (Primitive::J(control) sums the position channels; the yaw effort is Jyaw, not part of the edge cost) */
  int k = control == ORC_VEL ? 1 : control == ORC_ACC ? 2 : control == ORC_JRK ? 3 : 4;
  static const double fact[6] = {1, 1, 2, 6, 24, 120};
  double total = 0.0;
  for (int ax = 0; ax < 3; ax++) {
    const double *c = pr->c[ax];
    double q[6];
    int nq = 6 - k; /* derivative has degree 5-k */
    /* p = sum_{m=0..5} c[5-m]/m! t^m ; k-th derivative coefficient of t^(m-k) is c[5-m]/(m-k)! */
    for (int m = k; m <= 5; m++) q[m - k] = c[5 - m] / fact[m - k];
    double s = 0.0;
    for (int i = 0; i < nq; i++)
      for (int j = 0; j < nq; j++) s += q[i] * q[j] * pw(pr->t, i + j + 1) / (double)(i + j + 1);
    total += s;
  }
  return total;
}

/* ------------------------------------------------------------------ Waypoint key (a2) */
/* [UNVERIFIED waypoint.h hash_value] per axis: pos/0.01, vel/0.1, acc/0.1, jrk/0.1 for the
 * enabled fields, then t/0.1 when enable_t.  [DEVIATION D3] nodes are identified by this integer
 * tuple itself, not by boost::hash_combine of it (upstream operator== compares hash values; the
 * tuple is the semantic intent and is free of hash collisions). */
int orc_waypoint_key(const orc_waypoint *w, int32_t *key) {
  int n = 0;
  for (int i = 0; i < 3; i++) {
    if (w->control & 1) key[n++] = (int32_t)round(w->pos[i] / ORC_HASH_RES_POS);
    if (w->control & 2) key[n++] = (int32_t)round(w->vel[i] / ORC_HASH_RES_VEL);
    if (w->control & 4) key[n++] = (int32_t)round(w->acc[i] / ORC_HASH_RES_ACC);
    if (w->control & 8) key[n++] = (int32_t)round(w->jrk[i] / ORC_HASH_RES_JRK);
  }
  if (w->control & ORC_YAW) key[n++] = (int32_t)round(w->yaw / ORC_HASH_RES_YAW); /* [UNVERIFIED] use_yaw: yaw / 0.1 */
  if (w->enable_t) key[n++] = (int32_t)round(w->t / ORC_HASH_RES_T);
  return n;
}
static int key_equal(const int32_t *a, int na, const int32_t *b, int nb) {
  if (na != nb) return 0;
  for (int i = 0; i < na; i++)
    if (a[i] != b[i]) return 0;
  return 1;
}

/* ------------------------------------------------------------------ planner state */
typedef struct {
  int32_t parent, action, next;
  double cost;
  int32_t blocked, child; /* LPA*: the edge's primitive is not free in the current map (cost counts as +inf); node the entry belongs to */
} orc_edge;

typedef struct {
  orc_waypoint coord;
  int32_t key[16];
  int32_t nkey;
  double g, h;
  int32_t opened, closed;
  int32_t heap_pos;  /* -1 when not in heap */
  int32_t pred_head; /* linked list into edges in push_back (arrival) order, -1 empty */
  int32_t pred_tail;
  double rhs;         /* LPA* one-step look-ahead value */
  int32_t succ_built; /* LPA*: get_succ has been called for this node (its successors own pred entries for it) */
  int32_t pad;
} orc_node;

typedef struct { /* successor emitted with +inf cost: hm_ entry + pred entry upstream, never relaxed */
  int32_t parent, action;
  int32_t converted, pad; /* LPA*: cleared later and turned into a real pred entry */
} orc_blocked;
typedef struct { /* LPA* queue entry (lazy deletion) */
  double k, kg;
  int32_t id, pad;
} orc_lentry;

struct orc_planner {
  /* map (a7) */
  int8_t *map;
  int map_owned; /* 0: adopted with orc_set_map_shared (read-only, not freed) */
  int32_t dim[3];
  double origin[3], res;
  /* config */
  orc_config cfg;
  double *U, *U_yaw;
  orc_waypoint goal;
  int has_goal;
  /* state space */
  orc_node *nodes;
  int n_nodes, cap_nodes;
  orc_edge *edges;
  int n_edges, cap_edges;
  int32_t *table;
  int cap_table; /* power of two */
  int32_t *heap;
  int n_heap, cap_heap;
  /* results */
  int32_t *expanded;
  int n_expanded, cap_expanded;
  int n_closed;
  orc_blocked *blocked; /* in arrival order */
  int n_blocked, cap_blocked;
  int32_t *traj_nodes;
  int32_t *traj_actions;
  orc_waypoint *traj_wps; /* coords of the path states at recoverTraj time (getTraj() returns the stored Trajectory) */
  int traj_len;
  double traj_cost;
  orc_counters cnt;
  /* potential field / search region (mpl_oracle_pot.inc): 0..100 potential, -1 outside the search region */
  int8_t *aux;
  int has_pot, has_region;
  double pot_weight, grad_weight;
  /* LPA* (mpl_oracle_lpa.inc) */
  int use_lpa, lpa_valid;
  int lpa_reroot; /* getSubStateSpace: 0 Dijkstra through the expanded states (L5), 1 plan afresh (L5b), 2 auto (default) */
  orc_lentry *lq;
  int n_lq, cap_lq;
  int root_id, goal_id;
  orc_waypoint lpa_goal;
  int lpa_iterations, lpa_get_succ_calls;
};

orc_planner *orc_create(void) {
  orc_planner *p = (orc_planner *)calloc(1, sizeof(orc_planner));
  p->cfg.control = ORC_ACC;
  p->cfg.dt = 1.0;
  p->cfg.v_max = p->cfg.a_max = p->cfg.j_max = -1.0; /* [UNVERIFIED env_base defaults] */
  p->cfg.w = 10.0;
  p->cfg.eps = 1.0;
  p->cfg.tol_pos = 0.5;
  p->cfg.tol_vel = p->cfg.tol_acc = p->cfg.tol_yaw = -1.0;
  p->cfg.yaw_max = -1.0;
  p->cfg.t_max = INFINITY;
  p->cfg.max_expand = -1;
  p->traj_cost = INFINITY;
  p->lpa_reroot = 2;
  return p;
}
static void free_search(orc_planner *p) {
  free(p->nodes); free(p->edges); free(p->table); free(p->heap); free(p->expanded);
  free(p->traj_nodes); free(p->traj_actions); free(p->blocked); free(p->lq); free(p->traj_wps);
  p->traj_wps = NULL;
  p->lq = NULL; p->n_lq = p->cap_lq = 0;
  p->nodes = NULL; p->edges = NULL; p->table = NULL; p->heap = NULL; p->expanded = NULL;
  p->traj_nodes = NULL; p->traj_actions = NULL; p->blocked = NULL;
  p->n_blocked = p->cap_blocked = 0;
  p->n_nodes = p->cap_nodes = p->n_edges = p->cap_edges = p->cap_table = 0;
  p->n_heap = p->cap_heap = p->n_expanded = p->cap_expanded = p->n_closed = p->traj_len = 0;
}
void orc_destroy(orc_planner *p) {
  free(p->aux);
  p->aux = NULL;
  free(p->U_yaw);
  if (!p) return;
  free_search(p);
  if (p->map_owned) free(p->map);
  free(p->U);
  free(p);
}

/* ------------------------------------------------------------------ MapUtil (a7) */
void orc_potential_clear(orc_planner *p);
void orc_set_map(orc_planner *p, const int8_t *data, const int32_t dim[3], const double origin[3], double res) {
  if (p->aux && (dim[0] != p->dim[0] || dim[1] != p->dim[1] || dim[2] != p->dim[2])) orc_potential_clear(p);
  if (p->map_owned) free(p->map);
  size_t n = (size_t)dim[0] * dim[1] * dim[2];
  p->map = (int8_t *)malloc(n);
  p->map_owned = 1;
  memcpy(p->map, data, n);
  for (int i = 0; i < 3; i++) {
    p->dim[i] = dim[i];
    p->origin[i] = origin[i];
  }
  p->res = res;
}
/* Adopt the caller's grid without copying it (many planner objects, one per thread, on one read-only
 * map: the CPU baseline of bench.py).  The caller keeps it alive; free_unknown / dilate must not be
 * called on such a planner (they would write into the shared grid). */
void orc_set_map_shared(orc_planner *p, const int8_t *data, const int32_t dim[3], const double origin[3], double res) {
  if (p->map_owned) free(p->map);
  p->map = (int8_t *)data;
  p->map_owned = 0;
  for (int i = 0; i < 3; i++) {
    p->dim[i] = dim[i];
    p->origin[i] = origin[i];
  }
  p->res = res;
}
/* [IN-TREE map_planner_node.cpp:71 calls it]  unknown (-1) -> free (0) */
void orc_free_unknown(orc_planner *p) {
  size_t n = (size_t)p->dim[0] * p->dim[1] * p->dim[2];
  for (size_t i = 0; i < n; i++)
    if (p->map[i] == -1) p->map[i] = 0;
}
/* [UNVERIFIED map_util.h dilate(const vec_Veci&); IN-TREE call map_planner_node.cpp:75-85]  every
 * occupied voxel marks voxel + offset occupied (val_occ = 100, voxel_grid.h:43-45) when that is inside
 * the map; written to a copy so that dilation does not cascade (scatter form, as upstream loops). */
void orc_map_dilate(orc_planner *p, int n_off, const int32_t *off) {
  size_t n = (size_t)p->dim[0] * p->dim[1] * p->dim[2];
  int8_t *out = (int8_t *)malloc(n);
  memcpy(out, p->map, n);
  for (int z = 0; z < p->dim[2]; z++)
    for (int y = 0; y < p->dim[1]; y++)
      for (int x = 0; x < p->dim[0]; x++) {
        if (!(p->map[(size_t)x + (size_t)p->dim[0] * y + (size_t)p->dim[0] * p->dim[1] * z] > 0)) continue;
        for (int k = 0; k < n_off; k++) {
          int tx = x + off[3 * k], ty = y + off[3 * k + 1], tz = z + off[3 * k + 2];
          if (tx < 0 || tx >= p->dim[0] || ty < 0 || ty >= p->dim[1] || tz < 0 || tz >= p->dim[2]) continue;
          out[(size_t)tx + (size_t)p->dim[0] * ty + (size_t)p->dim[0] * p->dim[1] * tz] = 100;
        }
      }
  free(p->map);
  p->map = out;
}
void orc_map_get(const orc_planner *p, int8_t *out) { memcpy(out, p->map, (size_t)p->dim[0] * p->dim[1] * p->dim[2]); }
/* [UNVERIFIED map_util.h isFree/isOccupied/isUnknown/isOutside(const Veci&); IN-TREE calls
 * map_replanner_node.cpp:180,217]  0 free (== 0), 1 occupied (> 0), 2 unknown, 3 outside */
int orc_map_cell_state(const orc_planner *p, const int32_t pn[3]) {
  if (pn[0] < 0 || pn[0] >= p->dim[0] || pn[1] < 0 || pn[1] >= p->dim[1] || pn[2] < 0 || pn[2] >= p->dim[2]) return 3;
  int8_t v = p->map[(size_t)pn[0] + (size_t)p->dim[0] * pn[1] + (size_t)p->dim[0] * p->dim[1] * pn[2]];
  return v == 0 ? 0 : (v > 0 ? 1 : 2);
}
/* [UNVERIFIED map_util.h rayTrace(pt1, pt2); IN-TREE calls map_replanner_node.cpp:177,208]  walk from
 * pt1 towards pt2 in max_diff = int(|diff/res|_inf / 0.8) equal steps (end points excluded), stop at
 * the first cell outside the map, emit a cell when it differs from the previous one. */
int orc_map_raytrace(const orc_planner *p, const double p1[3], const double p2[3], int32_t *cells, int cap) {
  double diff[3], m = 0.0;
  for (int i = 0; i < 3; i++) {
    diff[i] = p2[i] - p1[i];
    double q = fabs(diff[i] / p->res);
    if (q > m) m = q;
  }
  double k = 0.8;
  int max_diff = (int)(m / k);
  double s = 1.0 / max_diff;
  double step[3] = {diff[0] * s, diff[1] * s, diff[2] * s};
  int32_t prev[3] = {-1, -1, -1};
  int cnt = 0;
  for (int n = 1; n < max_diff; n++) {
    double pt[3] = {p1[0] + step[0] * n, p1[1] + step[1] * n, p1[2] + step[2] * n};
    int32_t pn[3];
    for (int i = 0; i < 3; i++) pn[i] = (int32_t)round((pt[i] - p->origin[i]) / p->res - 0.5);
    if (pn[0] < 0 || pn[0] >= p->dim[0] || pn[1] < 0 || pn[1] >= p->dim[1] || pn[2] < 0 || pn[2] >= p->dim[2]) break;
    if (pn[0] != prev[0] || pn[1] != prev[1] || pn[2] != prev[2]) {
      if (cnt < cap) {
        cells[3 * cnt] = pn[0];
        cells[3 * cnt + 1] = pn[1];
        cells[3 * cnt + 2] = pn[2];
      }
      cnt++;
    }
    prev[0] = pn[0];
    prev[1] = pn[1];
    prev[2] = pn[2];
  }
  return cnt;
}
/* [UNVERIFIED map_util.h getCloud/getFreeCloud/getUnknownCloud; IN-TREE twin voxel_grid.cpp:18-29 (loop
 * order x, y, z) and :205-207 (intToFloat = (n + 0.5) res + origin); calls map_display.cpp:244,256,266]
 * which: 0 occupied, 1 free, 2 unknown.  Returns the number of voxels of the class. */
uint64_t orc_map_cloud(const orc_planner *p, int which, double *pts, uint64_t cap) {
  uint64_t cnt = 0;
  for (int x = 0; x < p->dim[0]; x++)
    for (int y = 0; y < p->dim[1]; y++)
      for (int z = 0; z < p->dim[2]; z++) {
        int8_t v = p->map[(size_t)x + (size_t)p->dim[0] * y + (size_t)p->dim[0] * p->dim[1] * z];
        int match = which == 0 ? v > 0 : which == 1 ? v == 0 : v < 0;
        if (!match) continue;
        if (cnt < cap) {
          pts[3 * cnt] = ((double)x + 0.5) * p->res + p->origin[0];
          pts[3 * cnt + 1] = ((double)y + 0.5) * p->res + p->origin[1];
          pts[3 * cnt + 2] = ((double)z + 0.5) * p->res + p->origin[2];
        }
        cnt++;
      }
  return cnt;
}
/* [UNVERIFIED map_util.h floatToInt] round((pt - origin)/res - 0.5) */
void orc_float_to_int(const orc_planner *p, const double pt[3], int32_t pn[3]) {
  for (int i = 0; i < 3; i++) pn[i] = (int32_t)round((pt[i] - p->origin[i]) / p->res - 0.5);
}
static int is_outside(const orc_planner *p, const int32_t pn[3]) {
  return pn[0] < 0 || pn[0] >= p->dim[0] || pn[1] < 0 || pn[1] >= p->dim[1] || pn[2] < 0 || pn[2] >= p->dim[2];
}
/* [IN-TREE voxel_grid.cpp:88] x-fastest */
static size_t get_index(const orc_planner *p, const int32_t pn[3]) {
  return (size_t)pn[0] + (size_t)p->dim[0] * pn[1] + (size_t)p->dim[0] * p->dim[1] * pn[2];
}
/* [UNVERIFIED env_map::is_free(pt)] inside and map == 0 */
int orc_is_free_point(const orc_planner *p, const double pt[3]) {
  int32_t pn[3];
  orc_float_to_int(p, pt, pn);
  if (is_outside(p, pn)) return 0;
  return p->map[get_index(p, pn)] == 0;
}
/* [UNVERIFIED env_map::is_free(pr); density rule IN-TREE ellipsoid_util.h:67-70]
 * max_v over axes; n = ceil(max_v * t / res); sample(n) = evaluate(i * (t/n)), i=0..n;
 * blocked if a sample is outside or occupied (>0).
 * [DEVIATION D4] n == 0 (stationary primitive) would give dt = t/0 = inf and evaluate(0*inf = NaN)
 * upstream; here it checks the single sample t = 0. */
int orc_is_free_primitive(orc_planner *p, const orc_primitive *pr) {
  double max_v = 0;
  for (int i = 0; i < 3; i++) {
    double v = orc_primitive_max_vel(pr, i);
    if (v > max_v) max_v = v;
  }
  int n = (int)ceil(max_v * pr->t / p->res);
  double dt = n > 0 ? pr->t / n : 0.0;
  for (int i = 0; i <= n; i++) {
    double t = i * dt, pt[3];
    int32_t pn[3];
    for (int k = 0; k < 3; k++) pt[k] = p1_p(pr->c[k], t);
    orc_float_to_int(p, pt, pn);
    if (g_q1_on) { /* Q1 audit: the same sample through the recollected upstream form */
      double pt2[3];
      int32_t pn2[3];
      for (int k = 0; k < 3; k++) pt2[k] = p1_p_libm_pow(pr->c[k], t);
      orc_float_to_int(p, pt2, pn2);
      g_q1_samples++;
      if (memcmp(pt, pt2, sizeof(pt)) != 0) g_q1_pos_bits++;
      if (pn[0] != pn2[0] || pn[1] != pn2[1] || pn[2] != pn2[2]) g_q1_cells++;
    }
    if (is_outside(p, pn)) return 0;
    p->cnt.n_voxel_reads++;
    if (p->map[get_index(p, pn)] > 0) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ config / goal */
void orc_set_config(orc_planner *p, const orc_config *cfg) {
  free(p->U);
  p->cfg = *cfg;
  p->U = (double *)malloc(sizeof(double) * 3 * (size_t)cfg->n_u);
  memcpy(p->U, cfg->U, sizeof(double) * 3 * (size_t)cfg->n_u);
  p->cfg.U = p->U;
  free(p->U_yaw);
  p->U_yaw = NULL;
  if (cfg->U_yaw) {
    p->U_yaw = (double *)malloc(sizeof(double) * (size_t)cfg->n_u);
    memcpy(p->U_yaw, cfg->U_yaw, sizeof(double) * (size_t)cfg->n_u);
  }
  p->cfg.U_yaw = p->U_yaw;
}
void orc_set_goal(orc_planner *p, const orc_waypoint *goal) {
  p->goal = *goal;
  p->has_goal = 1;
}

/* ------------------------------------------------------------------ env_base heuristic / goal (a11) */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double linf3(const double *a, const double *b) {
  double m = 0;
  for (int i = 0; i < 3; i++) {
    double d = fabs(a[i] - b[i]);
    if (d > m) m = d;
  }
  return m;
}
/* [UNVERIFIED env_base::is_goal] */
int orc_is_goal(const orc_planner *p, const orc_waypoint *s) {
  if (s->t >= p->cfg.t_max) return 1;
  int goaled = linf3(s->pos, p->goal.pos) <= p->cfg.tol_pos;
  if (goaled && (p->goal.control & 2) && p->cfg.tol_vel >= 0) goaled = linf3(s->vel, p->goal.vel) <= p->cfg.tol_vel;
  if (goaled && (p->goal.control & 4) && p->cfg.tol_acc >= 0) goaled = linf3(s->acc, p->goal.acc) <= p->cfg.tol_acc;
  if (goaled && (s->control & ORC_YAW) && p->cfg.tol_yaw >= 0) goaled = fabs(s->yaw - p->goal.yaw) <= p->cfg.tol_yaw; /* [UNVERIFIED] */
  return goaled;
}
/* min over candidate times t >= t_bar of  a t - c/t - d/2/t^2 - e/3/t^3 - f/4/t^4 - g/5/t^5,
 * candidates = real roots of a t^6 + c t^4 + d t^3 + e t^2 + f t + g  (b == 0) plus t_bar. */
static double heur_min6(double a, double c, double d, double e, double f, double g, double t_bar) {
  double co[7] = {g, f, e, d, c, 0.0, a}, ts[8];
  int n = orc_poly_roots_above(co, 6, t_bar > 0 ? t_bar : 0.0, ts);
  ts[n++] = t_bar;
  double best = INFINITY;
  for (int i = 0; i < n; i++) {
    double t = ts[i];
    if (t < t_bar) continue;
    double cost = a * t - c / t - d / 2 / t / t - e / 3 / t / t / t - f / 4 / t / t / t / t - g / 5 / t / t / t / t / t;
    if (cost < best) best = cost;
  }
  return best;
}
/* candidates = real roots of c5 t^4 + c3 t^2 + c2 t + c1 plus t_bar; cost -c1/3/t^3 - c2/2/t^2 - c3/t + w t */
static double heur_min4(double c5, double c3, double c2, double c1, double w, double t_bar) {
  double co[5] = {c1, c2, c3, 0.0, c5}, ts[6];
  int n = orc_poly_roots_above(co, 4, t_bar > 0 ? t_bar : 0.0, ts);
  ts[n++] = t_bar;
  double best = INFINITY;
  for (int i = 0; i < n; i++) {
    double t = ts[i];
    if (t < t_bar) continue;
    double c = -c1 / 3 / t / t / t - c2 / 2 / t / t - c3 / t + w * t;
    if (c < best) best = c;
  }
  return best;
}
/* [UNVERIFIED env_base::cal_heur]  minimum over T of (optimal-control effort to reach the goal in
 * time T) + w T, T >= |dp|_inf / v_max.  The polynomial coefficients below are pinned by
 * tests/test_oracle_kat.py against a numerical optimal-control solution. */
static double cal_heur(const orc_planner *p, const orc_waypoint *s, const orc_waypoint *goal) {
  const double w = p->cfg.w, v_max = p->cfg.v_max;
  double dp[3];
  for (int i = 0; i < 3; i++) dp[i] = goal->pos[i] - s->pos[i];
  /* [DEVIATION D8 / UNVERIFIED env_base::cal_heur] v_max <= 0 means "unlimited" (the default, -1): the bound on the
   * arrival time |dp|_inf / v_max does not exist then -- the kinematic term is dropped (upstream guards
   * the division with `v_max_ > 0`), never divided by a non-positive number. */
  if (p->cfg.heur_ignore_dynamics) return v_max > 0 ? w * linf3(s->pos, goal->pos) / v_max : w * linf3(s->pos, goal->pos);
  const double *v0 = s->vel, *v1 = goal->vel, *a0 = s->acc, *a1 = goal->acc;
  double t_bar = v_max > 0 ? linf3(s->pos, goal->pos) / v_max : 0.0;
  const int sc = s->control & 15, gc = goal->control & 15; /* [UNVERIFIED] the yaw bit does not select another cost-to-go */
  if (sc == ORC_JRK && gc == ORC_JRK) {
    double a0ma1[3] = {a0[0] - a1[0], a0[1] - a1[1], a0[2] - a1[2]};
    double v0pv1[3] = {v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2]};
    double c = -9 * dot3(a0, a0) + 6 * dot3(a0, a1) - 9 * dot3(a1, a1);
    double d = -144 * dot3(a0, v0) - 96 * dot3(a0, v1) + 96 * dot3(a1, v0) + 144 * dot3(a1, v1);
    double e = 360 * dot3(a0ma1, dp) - 576 * dot3(v0, v0) - 1008 * dot3(v0, v1) - 576 * dot3(v1, v1);
    double f = 2880 * dot3(dp, v0pv1);
    double g = -3600 * dot3(dp, dp);
    return heur_min6(w, c, d, e, f, g, t_bar);
  } else if (sc == ORC_JRK && gc == ORC_ACC) {
    double c = -8 * dot3(a0, a0);
    double d = -112 * dot3(a0, v0) - 48 * dot3(a0, v1);
    double e = 240 * dot3(a0, dp) - 384 * dot3(v0, v0) - 432 * dot3(v0, v1) - 144 * dot3(v1, v1);
    double q[3] = {1600 * v0[0] + 960 * v1[0], 1600 * v0[1] + 960 * v1[1], 1600 * v0[2] + 960 * v1[2]};
    double f = dot3(dp, q);
    double g = -1600 * dot3(dp, dp);
    return heur_min6(w, c, d, e, f, g, t_bar);
  } else if (sc == ORC_JRK && gc == ORC_VEL) {
    double c = -5 * dot3(a0, a0);
    double d = -40 * dot3(a0, v0);
    double e = 60 * dot3(a0, dp) - 60 * dot3(v0, v0);
    double f = 160 * dot3(dp, v0);
    double g = -100 * dot3(dp, dp);
    return heur_min6(w, c, d, e, f, g, t_bar);
  } else if (sc == ORC_ACC && gc == ORC_ACC) {
    double v0pv1[3] = {v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2]};
    double c1 = -36 * dot3(dp, dp);
    double c2 = 24 * dot3(v0pv1, dp);
    double c3 = -4 * (dot3(v0, v0) + dot3(v0, v1) + dot3(v1, v1));
    return heur_min4(w, c3, c2, c1, w, t_bar);
  } else if (sc == ORC_ACC && gc == ORC_VEL) {
    double c1 = -9 * dot3(dp, dp);
    double c2 = 12 * dot3(v0, dp);
    double c3 = -3 * dot3(v0, v0);
    return heur_min4(w, c3, c2, c1, w, t_bar);
  } else if (sc == ORC_VEL && gc == ORC_VEL) {
    return (w + 1) * sqrt(dot3(dp, dp));
  }
  return v_max > 0 ? w * sqrt(dot3(dp, dp)) / v_max : w * sqrt(dot3(dp, dp));
}
/* [UNVERIFIED env_base::get_heur] 0 when the state hashes equal to the goal */
double orc_heuristic(const orc_planner *p, const orc_waypoint *s) {
  int32_t ka[16], kb[16];
  int na = orc_waypoint_key(s, ka), nb = orc_waypoint_key(&p->goal, kb);
  if (s->control == p->goal.control && key_equal(ka, na, kb, nb)) return 0;
  return cal_heur(p, s, &p->goal);
}

#include "mpl_oracle_pot.inc"

/* ------------------------------------------------------------------ env_map::get_succ (a8) */
/* [IN-TREE env_poly_map.h:45-69, env_cloud.h:50-70 for the control flow;
 *  UNVERIFIED env_map.h for `tn == curr` + validate + is_free(pr) ? J + w dt : inf] */
int orc_get_succ(orc_planner *p, const orc_waypoint *curr, orc_waypoint *succ, double *succ_cost, int32_t *action_idx) {
  int n = 0;
  int32_t kc[16], kt[16];
  int nkc = orc_waypoint_key(curr, kc);
  p->cnt.n_expansions++;
  for (int i = 0; i < p->cfg.n_u; i++) {
    orc_primitive pr;
    orc_waypoint tn;
    if ((curr->control & ORC_YAW) && p->U_yaw) orc_primitive_build_yaw(curr, p->U + 3 * i, p->U_yaw[i], p->cfg.dt, &pr);
    else orc_primitive_build(curr, p->U + 3 * i, p->cfg.dt, &pr);
    p->cnt.n_primitives++;
    orc_primitive_evaluate(&pr, p->cfg.dt, &tn);
    tn.enable_t = 0; /* compared before tn.t is assigned; env_map never sets enable_t */
    int nkt = orc_waypoint_key(&tn, kt);
    if (key_equal(kt, nkt, kc, nkc) || !orc_validate_primitive(&pr, p->cfg.v_max, p->cfg.a_max, p->cfg.j_max)) continue;
    if ((pr.control & ORC_YAW) && !orc_validate_yaw(&pr, p->cfg.yaw_max)) continue; /* validate_primitive(pr, mv, ma, mj, myaw) */
    tn.t = curr->t + p->cfg.dt; /* [IN-TREE env_cloud.h:65] */
    long potsum = 0;
    const int free_ = p->aux ? prim_traverse(p, &pr, &potsum) : orc_is_free_primitive(p, &pr);
    double cost = free_ ? orc_primitive_J(&pr, pr.control) + p->cfg.w * p->cfg.dt : INFINITY;
    if (free_ && p->aux) cost = cost + p->pot_weight * (double)potsum; /* [P3 of mpl_oracle_pot.inc] */
    succ[n] = tn;
    succ_cost[n] = cost;
    action_idx[n] = i;
    n++;
    p->cnt.n_succ++;
    if (!isinf(cost)) p->cnt.n_succ_finite++;
  }
  return n;
}

/* ------------------------------------------------------------------ state space: hash table */
static uint64_t key_hash(const int32_t *k, int n) {
  uint64_t h = 0xcbf29ce484222325ULL;
  for (int i = 0; i < n; i++) {
    h ^= (uint32_t)k[i];
    h *= 0x100000001b3ULL;
    h ^= h >> 29;
  }
  return h;
}
static void table_insert_raw(orc_planner *p, int id) {
  uint64_t m = (uint64_t)p->cap_table - 1, s = key_hash(p->nodes[id].key, p->nodes[id].nkey) & m;
  while (p->table[s] >= 0) s = (s + 1) & m;
  p->table[s] = id;
}
static void table_grow(orc_planner *p) {
  int ncap = p->cap_table ? p->cap_table * 2 : 1 << 16;
  free(p->table);
  p->table = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncap);
  for (int i = 0; i < ncap; i++) p->table[i] = -1;
  p->cap_table = ncap;
  for (int i = 0; i < p->n_nodes; i++) table_insert_raw(p, i);
}
static int table_find(const orc_planner *p, const int32_t *key, int nkey) {
  if (!p->cap_table) return -1;
  uint64_t m = (uint64_t)p->cap_table - 1, s = key_hash(key, nkey) & m;
  while (p->table[s] >= 0) {
    const orc_node *nd = &p->nodes[p->table[s]];
    if (key_equal(nd->key, nd->nkey, key, nkey)) return p->table[s];
    s = (s + 1) & m;
  }
  return -1;
}
static int node_create(orc_planner *p, const orc_waypoint *coord, const int32_t *key, int nkey) {
  if (p->n_nodes == p->cap_nodes) {
    p->cap_nodes = p->cap_nodes ? p->cap_nodes * 2 : 1 << 14;
    p->nodes = (orc_node *)realloc(p->nodes, sizeof(orc_node) * (size_t)p->cap_nodes);
  }
  int id = p->n_nodes++;
  orc_node *nd = &p->nodes[id];
  nd->coord = *coord;
  memcpy(nd->key, key, sizeof(int32_t) * (size_t)nkey);
  nd->nkey = nkey;
  nd->g = INFINITY;
  nd->h = 0;
  nd->opened = nd->closed = 0;
  nd->heap_pos = -1;
  nd->pred_head = nd->pred_tail = -1;
  nd->rhs = INFINITY;
  nd->succ_built = 0;
  nd->pad = 0;
  if ((size_t)p->n_nodes * 2 > (size_t)p->cap_table)
    table_grow(p);
  else
    table_insert_raw(p, id);
  p->cnt.n_new_nodes++;
  return id;
}
static void edge_append(orc_planner *p, int child, int parent, int action, double cost) {
  if (p->n_edges == p->cap_edges) {
    p->cap_edges = p->cap_edges ? p->cap_edges * 2 : 1 << 16;
    p->edges = (orc_edge *)realloc(p->edges, sizeof(orc_edge) * (size_t)p->cap_edges);
  }
  int e = p->n_edges++;
  p->edges[e].parent = parent;
  p->edges[e].action = action;
  p->edges[e].cost = cost;
  p->edges[e].blocked = 0;
  p->edges[e].child = child;
  p->edges[e].next = -1;
  orc_node *nd = &p->nodes[child];
  if (nd->pred_tail < 0)
    nd->pred_head = e;
  else
    p->edges[nd->pred_tail].next = e;
  nd->pred_tail = e; /* list kept in push_back order, like upstream's pred_* vectors */
}

/* ------------------------------------------------------------------ OPEN: indexed binary min-heap
 * [UNVERIFIED graph_search.h] upstream: boost d_ary_heap<arity 2, mutable> of (fval, node) with
 * compare_pair = { f equal ? min(g,rhs) larger loses : f larger loses }.
 * [DEVIATION D5] remaining ties are unspecified upstream (heap-internal); the total order here is
 * (f, g, node id) ascending, node id = creation order.  Any exact min-priority structure yields
 * the same pop sequence under a strict total order. */
static int heap_less(const orc_planner *p, int a, int b) {
  const orc_node *x = &p->nodes[a], *y = &p->nodes[b];
  double fx = x->g + p->cfg.eps * x->h, fy = y->g + p->cfg.eps * y->h;
  if (fx != fy) return fx < fy;
  if (x->g != y->g) return x->g < y->g;
  return a < b;
}
static void heap_swap(orc_planner *p, int i, int j) {
  int32_t t = p->heap[i];
  p->heap[i] = p->heap[j];
  p->heap[j] = t;
  p->nodes[p->heap[i]].heap_pos = i;
  p->nodes[p->heap[j]].heap_pos = j;
}
static void heap_up(orc_planner *p, int i) {
  while (i > 0) {
    int par = (i - 1) / 2;
    if (!heap_less(p, p->heap[i], p->heap[par])) break;
    heap_swap(p, i, par);
    i = par;
  }
}
static void heap_down(orc_planner *p, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < p->n_heap && heap_less(p, p->heap[l], p->heap[m])) m = l;
    if (r < p->n_heap && heap_less(p, p->heap[r], p->heap[m])) m = r;
    if (m == i) break;
    heap_swap(p, i, m);
    i = m;
  }
}
static void heap_push(orc_planner *p, int id) {
  if (p->n_heap == p->cap_heap) {
    p->cap_heap = p->cap_heap ? p->cap_heap * 2 : 1 << 14;
    p->heap = (int32_t *)realloc(p->heap, sizeof(int32_t) * (size_t)p->cap_heap);
  }
  p->heap[p->n_heap] = id;
  p->nodes[id].heap_pos = p->n_heap;
  p->n_heap++;
  heap_up(p, p->n_heap - 1);
  p->cnt.n_heap_push++;
}
static int heap_pop(orc_planner *p) {
  int id = p->heap[0];
  p->n_heap--;
  if (p->n_heap > 0) {
    p->heap[0] = p->heap[p->n_heap];
    p->nodes[p->heap[0]].heap_pos = 0;
    heap_down(p, 0);
  }
  p->nodes[id].heap_pos = -1;
  return id;
}

/* ------------------------------------------------------------------ recoverTraj
 * [UNVERIFIED graph_search.h recoverTraj] from the goal node repeatedly choose the predecessor
 * edge minimising g(pred) + edge cost (tie: larger g(pred)) until the start key is reached. */
static int recover_traj(orc_planner *p, int node, int start_id) {
  int cap = 64, n = 0;
  int32_t *tn = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap), *ta = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
  tn[0] = node;
  while (p->nodes[node].pred_head >= 0) {
    int min_e = -1;
    double min_rhs = INFINITY, min_g = INFINITY;
    for (int e = p->nodes[node].pred_head; e >= 0; e = p->edges[e].next) {
      double gp = p->nodes[p->edges[e].parent].g, c = p->edges[e].blocked ? INFINITY : p->edges[e].cost;
      if (min_rhs > gp + c) {
        min_rhs = gp + c;
        min_g = gp;
        min_e = e;
      } else if (!isinf(c) && min_rhs == gp + c) {
        if (min_g < gp) {
          min_g = gp;
          min_e = e;
        }
      }
    }
    if (min_e < 0) {
      free(tn); free(ta);
      return 0;
    }
    if (n + 2 > cap) {
      cap *= 2;
      tn = (int32_t *)realloc(tn, sizeof(int32_t) * (size_t)cap);
      ta = (int32_t *)realloc(ta, sizeof(int32_t) * (size_t)cap);
    }
    ta[n] = p->edges[min_e].action;
    node = p->edges[min_e].parent;
    tn[++n] = node;
    if (node == start_id) break;
    if (n > p->n_nodes) { /* cycle guard: cannot happen with positive edge costs */
      free(tn); free(ta);
      return 0;
    }
  }
  /* reverse into start -> goal order */
  free(p->traj_nodes); free(p->traj_actions);
  p->traj_nodes = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
  p->traj_actions = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i <= n; i++) p->traj_nodes[i] = tn[n - i];
  for (int i = 0; i < n; i++) p->traj_actions[i] = ta[n - 1 - i];
  free(p->traj_wps);
  p->traj_wps = (orc_waypoint *)malloc(sizeof(orc_waypoint) * (size_t)(n + 1));
  for (int i = 0; i <= n; i++) p->traj_wps[i] = p->nodes[p->traj_nodes[i]].coord;
  p->traj_len = n;
  free(tn); free(ta);
  return 1;
}

/* ------------------------------------------------------------------ PlannerBase::plan + GraphSearch::Astar (a10, a12) */
/* [UNVERIFIED planner_base.h plan(): is_free(start.pos) else false; fresh StateSpace; set_goal;
 *  graph_search.h Astar(): loop order pop -> close -> get_succ -> relax -> goal test after expansion
 *  -> max_expand -> empty-queue] */
static int lpa_plan(orc_planner *p, const orc_waypoint *start, const orc_waypoint *goal);
int orc_plan(orc_planner *p, const orc_waypoint *start, const orc_waypoint *goal) {
  if (p->use_lpa) return lpa_plan(p, start, goal);
  free_search(p);
  p->lpa_valid = 0;
  p->traj_cost = INFINITY;
  if (!orc_is_free_point(p, start->pos)) return ORC_START_OCCUPIED;
  orc_set_goal(p, goal);
  if (orc_is_goal(p, start)) {
    p->traj_cost = 0;
    return ORC_OK;
  }
  orc_waypoint *succ = (orc_waypoint *)malloc(sizeof(orc_waypoint) * (size_t)p->cfg.n_u);
  double *succ_cost = (double *)malloc(sizeof(double) * (size_t)p->cfg.n_u);
  int32_t *succ_act = (int32_t *)malloc(sizeof(int32_t) * (size_t)p->cfg.n_u);
  int32_t key[16];
  int nkey = orc_waypoint_key(start, key);
  int start_id = node_create(p, start, key, nkey);
  p->nodes[start_id].g = 0;
  p->nodes[start_id].h = p->cfg.eps == 0 ? 0 : orc_heuristic(p, start);
  p->nodes[start_id].opened = 1;
  heap_push(p, start_id);
  int status = ORC_OK, expand_iteration = 0, curr;
  for (;;) {
    expand_iteration++;
    curr = heap_pop(p);
    if (!p->nodes[curr].closed) p->n_closed++;
    p->nodes[curr].closed = 1;
    if (p->n_expanded == p->cap_expanded) {
      p->cap_expanded = p->cap_expanded ? p->cap_expanded * 2 : 1 << 14;
      p->expanded = (int32_t *)realloc(p->expanded, sizeof(int32_t) * (size_t)p->cap_expanded);
    }
    p->expanded[p->n_expanded++] = curr;
    orc_waypoint cw = p->nodes[curr].coord; /* copy: node array may move */
    int ns = orc_get_succ(p, &cw, succ, succ_cost, succ_act);
    for (int s = 0; s < ns; s++) {
      if (isinf(succ_cost[s])) {
        /* [UNVERIFIED graph_search.h] upstream creates the hm_ entry of a blocked successor too (with its
         * heuristic) and pushes a pred entry with cost inf; the relaxation is a no-op (g + inf < g' is
         * never true).  [DEVIATION D7] such successors are kept in a side list (parent, action) instead of
         * the node array, so node ids stay "order of first finite arrival"; hm_.size(), getLinkedNodes and
         * getAllPrimitives are rebuilt from the two (orc_num_states_all, orc_get_blocked_edges). */
        if (p->n_blocked == p->cap_blocked) {
          p->cap_blocked = p->cap_blocked ? p->cap_blocked * 2 : 1 << 14;
          p->blocked = (orc_blocked *)realloc(p->blocked, sizeof(orc_blocked) * (size_t)p->cap_blocked);
        }
        p->blocked[p->n_blocked].parent = curr;
        p->blocked[p->n_blocked].action = succ_act[s];
        p->blocked[p->n_blocked].converted = 0;
        p->blocked[p->n_blocked].pad = 0;
        p->n_blocked++;
        continue;
      }
      nkey = orc_waypoint_key(&succ[s], key);
      int id = table_find(p, key, nkey);
      if (id < 0) {
        id = node_create(p, &succ[s], key, nkey);
        p->nodes[id].h = p->cfg.eps == 0 ? 0 : orc_heuristic(p, &succ[s]);
      }
      edge_append(p, id, curr, succ_act[s], succ_cost[s]);
      double tentative = p->nodes[curr].g + succ_cost[s];
      orc_node *nd = &p->nodes[id];
      if (tentative < nd->g) {
        nd->g = tentative;
        if (nd->opened && !nd->closed) {
          heap_up(p, nd->heap_pos);
          p->cnt.n_heap_decrease++;
        } else if (nd->opened && nd->closed) {
          /* [DEVIATION D6 / UNVERIFIED] a closed node improved (possible only with an inconsistent
           * heuristic, i.e. eps > 1): re-opened here, the textbook weighted-A* behaviour.  Upstream's
           * branch for this case prints "ASTAR ERROR!"; whether it also re-pushes the node could not be
           * checked (source absent).  n_reopen counts the events; it is 0 in every BASELINE config (eps = 1). */
          nd->closed = 0;
          p->n_closed--;
          heap_push(p, id);
          p->cnt.n_reopen++;
        } else {
          nd->opened = 1;
          heap_push(p, id);
        }
      }
    }
    if (orc_is_goal(p, &p->nodes[curr].coord)) break;
    if (p->cfg.max_expand > 0 && expand_iteration >= p->cfg.max_expand) {
      status = ORC_MAX_EXPAND;
      break;
    }
    if (p->n_heap == 0) {
      status = ORC_NO_PATH;
      break;
    }
  }
  free(succ); free(succ_cost); free(succ_act);
  if (status != ORC_OK) return status;
  if (!recover_traj(p, curr, start_id)) return ORC_NO_PATH;
  p->traj_cost = p->nodes[curr].g;
  return ORC_OK;
}

#include "mpl_oracle_lpa.inc"

/* ------------------------------------------------------------------ result getters */
double orc_traj_cost(const orc_planner *p) { return p->traj_cost; }
int orc_num_expanded(const orc_planner *p) { return p->n_expanded; }
void orc_get_expanded(const orc_planner *p, int32_t *ids, double *pos) {
  for (int i = 0; i < p->n_expanded; i++) {
    if (ids) ids[i] = p->expanded[i];
    if (pos)
      for (int k = 0; k < 3; k++) pos[3 * i + k] = p->nodes[p->expanded[i]].coord.pos[k];
  }
}
/* order-dependent fold of the expansion sequence (same fold as the product's mplx_result.expand_hash) */
uint64_t orc_expand_hash(const orc_planner *p) {
  uint64_t h = 0;
  for (int i = 0; i < p->n_expanded; i++) h = h * 0x100000001B3ULL + (uint64_t)((uint32_t)p->expanded[i] + 1u);
  return h;
}
int orc_num_nodes(const orc_planner *p) { return p->n_nodes; }
void orc_get_node(const orc_planner *p, int id, orc_waypoint *coord, double *g, double *h, int32_t *closed) {
  if (coord) *coord = p->nodes[id].coord;
  if (g) *g = p->nodes[id].g;
  if (h) *h = p->nodes[id].h;
  if (closed) *closed = p->nodes[id].closed;
}
int orc_num_closed(const orc_planner *p) { return p->n_closed; }
int orc_get_edges(const orc_planner *p, int32_t *child, int32_t *parent, int32_t *action, int cap) {
  int w = 0;
  for (int i = 0; i < p->n_nodes; i++)
    for (int e = p->nodes[i].pred_head; e >= 0; e = p->edges[e].next) {
      if (w < cap) {
        if (child) child[w] = i;
        if (parent) parent[w] = p->edges[e].parent;
        if (action) action[w] = p->edges[e].action;
      }
      w++;
    }
  return w;
}
/* pred entries with cost inf, in arrival order (expansion order of the parent, then action index) */
int orc_num_blocked(const orc_planner *p) { return p->n_blocked; }
void orc_get_blocked_edges(const orc_planner *p, int32_t *parent, int32_t *action) {
  for (int i = 0; i < p->n_blocked; i++) {
    if (parent) parent[i] = p->blocked[i].parent;
    if (action) action[i] = p->blocked[i].action;
  }
}
/* hm_.size() of upstream: states reached with finite cost + states only ever reached by blocked primitives */
static int cmp_key13(const void *a, const void *b) { return memcmp(a, b, sizeof(int32_t) * 14); }
int orc_num_states_all(const orc_planner *p) {
  int32_t *keys = (int32_t *)calloc((size_t)(p->n_blocked > 0 ? p->n_blocked : 1), sizeof(int32_t) * 14);
  int m = 0;
  for (int i = 0; i < p->n_blocked; i++) {
    orc_primitive pr;
    orc_waypoint tn;
    const orc_waypoint *from = &p->nodes[p->blocked[i].parent].coord;
    if ((from->control & ORC_YAW) && p->U_yaw) orc_primitive_build_yaw(from, p->U + 3 * p->blocked[i].action, p->U_yaw[p->blocked[i].action], p->cfg.dt, &pr);
    else orc_primitive_build(from, p->U + 3 * p->blocked[i].action, p->cfg.dt, &pr);
    orc_primitive_evaluate(&pr, p->cfg.dt, &tn);
    tn.enable_t = 0;
    int32_t *k = keys + 14 * (size_t)m;
    k[0] = orc_waypoint_key(&tn, k + 1);
    if (table_find(p, k + 1, k[0]) < 0) m++; else memset(k, 0, sizeof(int32_t) * 14);
  }
  qsort(keys, (size_t)m, sizeof(int32_t) * 14, cmp_key13);
  int distinct = 0;
  for (int i = 0; i < m; i++)
    if (i == 0 || memcmp(keys + 14 * (size_t)i, keys + 14 * (size_t)(i - 1), sizeof(int32_t) * 14) != 0) distinct++;
  free(keys);
  return p->n_nodes + distinct;
}
int orc_traj_len(const orc_planner *p) { return p->traj_len; }
/* primitives are rebuilt from the stored parent coord + action, like upstream forward_action */
void orc_get_traj(const orc_planner *p, orc_primitive *prs, orc_waypoint *wps, int32_t *actions, int32_t *node_ids) {
  for (int i = 0; i < p->traj_len; i++) {
    const orc_waypoint *from = &p->traj_wps[i];
    if (prs) {
      if ((from->control & ORC_YAW) && p->U_yaw) orc_primitive_build_yaw(from, p->U + 3 * p->traj_actions[i], p->U_yaw[p->traj_actions[i]], p->cfg.dt, &prs[i]);
      else orc_primitive_build(from, p->U + 3 * p->traj_actions[i], p->cfg.dt, &prs[i]);
    }
    if (actions) actions[i] = p->traj_actions[i];
  }
  if (p->traj_len > 0 || p->traj_nodes)
    for (int i = 0; i <= p->traj_len && p->traj_nodes; i++) {
      if (wps) wps[i] = p->traj_wps[i];
      if (node_ids) node_ids[i] = p->traj_nodes[i];
    }
}
void orc_get_counters(const orc_planner *p, orc_counters *c) { *c = p->cnt; }
void orc_reset_counters(orc_planner *p) { memset(&p->cnt, 0, sizeof(p->cnt)); }

/* ------------------------------------------------------------------ VoxelGrid (map ingest)
 * Line-by-line restatement of the in-tree planning_ros_utils/src/mapping_utils/voxel_grid.cpp (the one
 * component of this path whose reference source IS vendored).  Arrays are indexed [x][y][z] with z
 * fastest, like the boost::multi_array<char,3> they replace; res is a float like VoxelGrid::res_. */
struct orc_grid {
  int32_t dim[3], origin[3];
  double origin_d[3];
  float res;
  int8_t *map, *inflated;
};
#define GI(g, x, y, z) (((size_t)(x) * (size_t)(g)->dim[1] + (size_t)(y)) * (size_t)(g)->dim[2] + (size_t)(z))
static size_t grid_cells(const orc_grid *g) { return (size_t)g->dim[0] * g->dim[1] * g->dim[2]; }
/* [IN-TREE voxel_grid.cpp:129-181] */
int orc_grid_allocate(orc_grid *g, const double new_dim_d[3], const double new_ori_d[3]) {
  int32_t nd[3], no[3];
  for (int i = 0; i < 3; i++) {
    nd[i] = (int32_t)(new_dim_d[i] / g->res);
    no[i] = (int32_t)(new_ori_d[i] / g->res);
  }
  if (nd[2] == 0 && no[2] == 0) nd[2] = 1;
  if (nd[0] == g->dim[0] && nd[1] == g->dim[1] && nd[2] == g->dim[2] && no[0] == g->origin[0] && no[1] == g->origin[1] &&
      no[2] == g->origin[2])
    return 0;
  size_t n = (size_t)nd[0] * nd[1] * nd[2];
  int8_t *nm = (int8_t *)malloc(n ? n : 1);
  memset(nm, 0, n); /* val_free */
  for (int l = 0; l < nd[0]; l++)
    for (int w = 0; w < nd[1]; w++)
      for (int h = 0; h < nd[2]; h++)
        if (l + no[0] >= g->origin[0] && w + no[1] >= g->origin[1] && h + no[2] >= g->origin[2] && l + no[0] < g->origin[0] + g->dim[0] &&
            w + no[1] < g->origin[1] + g->dim[1] && h + no[2] < g->origin[2] + g->dim[2]) {
          int nl = l + no[0] - g->origin[0], nw = w + no[1] - g->origin[1], nh = h + no[2] - g->origin[2];
          nm[((size_t)l * nd[1] + w) * nd[2] + h] = g->map[GI(g, nl, nw, nh)];
        }
  free(g->map);
  free(g->inflated);
  g->map = nm;
  g->inflated = (int8_t *)malloc(n ? n : 1); /* inflated_map_ = new_map */
  memcpy(g->inflated, nm, n);
  for (int i = 0; i < 3; i++) {
    g->dim[i] = nd[i];
    g->origin[i] = no[i];
    g->origin_d[i] = new_ori_d[i];
  }
  return 1;
}
/* [IN-TREE voxel_grid.cpp:3-10] */
orc_grid *orc_grid_create(const double origin[3], const double dim[3], float res) {
  orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
  g->res = res;
  orc_grid_allocate(g, dim, origin);
  return g;
}
void orc_grid_destroy(orc_grid *g) {
  if (!g) return;
  free(g->map);
  free(g->inflated);
  free(g);
}
void orc_grid_info(const orc_grid *g, int32_t dim[3], double origin_d[3], float *res) {
  for (int i = 0; i < 3; i++) {
    dim[i] = g->dim[i];
    origin_d[i] = g->origin_d[i];
  }
  *res = g->res;
}
/* [IN-TREE voxel_grid.cpp:12-16] */
void orc_grid_clear(orc_grid *g) {
  memset(g->map, 0, grid_cells(g));
  memset(g->inflated, 0, grid_cells(g));
}
/* [IN-TREE voxel_grid.cpp:201-203] ((pt - origin_d_) / res_).cast<int>(): truncation towards zero */
static void grid_float_to_int(const orc_grid *g, const double *pt, int32_t *pn) {
  for (int i = 0; i < 3; i++) pn[i] = (int32_t)((pt[i] - g->origin_d[i]) / g->res);
}
static int grid_outside(const orc_grid *g, const int32_t *pn) {
  return pn[0] < 0 || pn[0] >= g->dim[0] || pn[1] < 0 || pn[1] >= g->dim[1] || pn[2] < 0 || pn[2] >= g->dim[2];
}
/* [IN-TREE voxel_grid.cpp:183-189] */
void orc_grid_add_cloud(orc_grid *g, int n, const double *pts) {
  for (int i = 0; i < n; i++) {
    int32_t pn[3];
    grid_float_to_int(g, pts + 3 * i, pn);
    if (grid_outside(g, pn)) continue;
    g->map[GI(g, pn[0], pn[1], pn[2])] = 100;
  }
}
/* [IN-TREE voxel_grid.cpp:191-207] returns the number of newly inflated cells (new_obs, cap x 3) */
int orc_grid_add_cloud_ns(orc_grid *g, int n, const double *pts, int n_ns, const int32_t *ns, int32_t *new_obs, int cap) {
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    int32_t pn[3];
    grid_float_to_int(g, pts + 3 * i, pn);
    if (grid_outside(g, pn)) continue;
    if (g->map[GI(g, pn[0], pn[1], pn[2])] != 100) {
      for (int k = 0; k < n_ns; k++) {
        int32_t n2[3] = {pn[0] + ns[3 * k], pn[1] + ns[3 * k + 1], pn[2] + ns[3 * k + 2]};
        if (!grid_outside(g, n2) && g->inflated[GI(g, n2[0], n2[1], n2[2])] != 100) {
          g->inflated[GI(g, n2[0], n2[1], n2[2])] = 100;
          if (cnt < cap) {
            new_obs[3 * cnt] = n2[0];
            new_obs[3 * cnt + 1] = n2[1];
            new_obs[3 * cnt + 2] = n2[2];
          }
          cnt++;
        }
      }
    }
    g->map[GI(g, pn[0], pn[1], pn[2])] = 100;
  }
  return cnt;
}
/* [IN-TREE voxel_grid.cpp:213-224] */
void orc_grid_decay(orc_grid *g) {
  size_t n = grid_cells(g);
  for (size_t i = 0; i < n; i++) {
    if (g->map[i] > 0) g->map[i]--;
    if (g->inflated[i] > 0) g->inflated[i]--;
  }
}
/* [IN-TREE voxel_grid.cpp:31-46] clear(nx,ny) has no bounds test upstream; out-of-range columns are ignored here */
void orc_grid_clear_column(orc_grid *g, int nx, int ny) {
  if (nx < 0 || nx >= g->dim[0] || ny < 0 || ny >= g->dim[1]) return;
  for (int nz = 0; nz < g->dim[2]; nz++) g->map[GI(g, nx, ny, nz)] = 0;
}
void orc_grid_fill_column(orc_grid *g, int nx, int ny) {
  if (nx >= 0 && nx < g->dim[0] && ny >= 0 && ny < g->dim[1])
    for (int nz = 0; nz < g->dim[2]; nz++) g->map[GI(g, nx, ny, nz)] = 100;
}
void orc_grid_fill_cell(orc_grid *g, int nx, int ny, int nz) {
  if (nx >= 0 && nx < g->dim[0] && ny >= 0 && ny < g->dim[1] && nz >= 0 && nz < g->dim[2]) g->map[GI(g, nx, ny, nz)] = 100;
}
/* [IN-TREE voxel_grid.cpp:71-127] getMap / getInflatedMap: VoxelMap.data, x fastest; > 0 -> 100, everything else 0 */
void orc_grid_get_map(const orc_grid *g, int inflated, int8_t *data) {
  const int8_t *m = inflated ? g->inflated : g->map;
  for (int x = 0; x < g->dim[0]; x++)
    for (int y = 0; y < g->dim[1]; y++)
      for (int z = 0; z < g->dim[2]; z++)
        data[(size_t)x + (size_t)g->dim[0] * y + (size_t)g->dim[0] * g->dim[1] * z] = m[GI(g, x, y, z)] > 0 ? 100 : 0;
}
/* [IN-TREE voxel_grid.cpp:18-29, 205-207] */
uint64_t orc_grid_get_cloud(const orc_grid *g, double *pts, uint64_t cap) {
  uint64_t cnt = 0;
  for (int x = 0; x < g->dim[0]; x++)
    for (int y = 0; y < g->dim[1]; y++)
      for (int z = 0; z < g->dim[2]; z++)
        if (g->map[GI(g, x, y, z)] > 0) {
          if (cnt < cap) {
            pts[3 * cnt] = ((double)x + 0.5) * g->res + g->origin_d[0];
            pts[3 * cnt + 1] = ((double)y + 0.5) * g->res + g->origin_d[1];
            pts[3 * cnt + 2] = ((double)z + 0.5) * g->res + g->origin_d[2];
          }
          cnt++;
        }
  return cnt;
}
