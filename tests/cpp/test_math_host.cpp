// Host-side (CPU) bit-exactness checks of the product's f64 math (mpl_ros_amd/csrc/mplx_math.h):
//   1. control-specialised evaluators == generic evaluators, bit for bit
//   2. product math == CPU oracle (oracle/mpl_oracle.c), bit for bit: primitive end state, extrema,
//      validate, J, key, heuristic, polynomial roots
// Built and run by tests/test_math_host.py (g++/hipcc host compile, no GPU).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../mpl_ros_amd/csrc/mplx_math.h"
#include "../../oracle/mpl_oracle.h"

using namespace mplx;

static uint64_t rng_state = 0x12345678abcdefULL;
static uint64_t rnd() {
  rng_state += 0x9E3779B97F4A7C15ULL;
  uint64_t z = rng_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static double uni(double a, double b) { return a + (b - a) * ((rnd() >> 11) * (1.0 / 9007199254740992.0)); }
static double special(int k) {
  static const double v[] = {0.0, -0.0, 1.0, -1.0, 0.5, 1e-300, -1e-300, 0.1, -0.1, 2.0, 1e-17, 123.456};
  return v[k % 12];
}
static bool same(double a, double b) { return memcmp(&a, &b, 8) == 0 || (a != a && b != b); }
static int fails = 0;
#define CHECK(cond, ...)            \
  do {                              \
    if (!(cond)) {                  \
      if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } \
      fails++;                      \
    }                               \
  } while (0)

template <int CONTROL>
static void check_control(int iters) {
  for (int it = 0; it < iters; it++) {
    State s;
    double u[3];
    for (int i = 0; i < 12; i++) ((double *)&s)[i] = (it % 7 == 0) ? special((int)(rnd() % 12)) : round(uni(-3, 3) * 100) / 100;
    for (int i = 0; i < 3; i++) u[i] = (it % 5 == 0) ? special((int)(rnd() % 12)) : (double)((int)(rnd() % 5) - 2) * 0.5;
    double T = (it % 3 == 0) ? 1.0 : uni(0.1, 2.0);
    double c[3][6];
    for (int ax = 0; ax < 3; ax++) prim_build_axis(CONTROL, s.p[ax], s.v[ax], s.a[ax], s.j[ax], u[ax], c[ax]);
    // 1. specialised == generic at T, 0 and sample times
    double ts[5] = {0.0, T, T / 3, T * 0.77, 7 * (T / 13)};
    for (double t : ts)
      for (int ax = 0; ax < 3; ax++) {
        CHECK(same(pos_at(c[ax], t), pos_at_c<CONTROL>(c[ax], t)), "pos ctrl %d", CONTROL);
        CHECK(same(vel_at(c[ax], t), vel_at_c<CONTROL>(c[ax], t)), "vel ctrl %d", CONTROL);
        CHECK(same(acc_at(c[ax], t), acc_at_c<CONTROL>(c[ax], t)), "acc ctrl %d", CONTROL);
        CHECK(same(jrk_at(c[ax], t), jrk_at_c<CONTROL>(c[ax], t)), "jrk ctrl %d", CONTROL);
        double q6[6] = {c[ax][0] / 120, c[ax][1] / 24, c[ax][2] / 6, c[ax][3] / 2, c[ax][4], c[ax][5]}, qc[5];
        pack_q_c<CONTROL>(c[ax], qc);
        CHECK(same(pos_at_q(q6, t), pos_at_qc<CONTROL>(qc, t)), "pos_q ctrl %d", CONTROL);
        CHECK(same(pos_at_q(q6, t), pos_at(c[ax], t)), "pos_q vs pos ctrl %d", CONTROL);
      }
    double mv_g, mv_c;
    bool okg = validate_and_maxv(CONTROL, c, T, 2.0, 1.0, 1.0, &mv_g);
    bool okc = validate_and_maxv_c<CONTROL>(c, T, 2.0, 1.0, 1.0, &mv_c);
    CHECK(okg == okc && same(mv_g, mv_c), "validate ctrl %d", CONTROL);
    // 2. product == oracle
    orc_waypoint w;
    memset(&w, 0, sizeof(w));
    for (int i = 0; i < 3; i++) { w.pos[i] = s.p[i]; w.vel[i] = s.v[i]; w.acc[i] = s.a[i]; w.jrk[i] = s.j[i]; }
    w.control = CONTROL;
    orc_primitive pr;
    orc_primitive_build(&w, u, T, &pr);
    for (int ax = 0; ax < 3; ax++)
      for (int k = 0; k < 6; k++) CHECK(same(pr.c[ax][k], c[ax][k]), "coeff");
    orc_waypoint e;
    orc_primitive_evaluate(&pr, T, &e);
    State tn;
    for (int ax = 0; ax < 3; ax++) {
      tn.p[ax] = pos_at_c<CONTROL>(c[ax], T); tn.v[ax] = vel_at_c<CONTROL>(c[ax], T);
      tn.a[ax] = acc_at_c<CONTROL>(c[ax], T); tn.j[ax] = jrk_at_c<CONTROL>(c[ax], T);
      CHECK(same(tn.p[ax], e.pos[ax]) && same(tn.v[ax], e.vel[ax]) && same(tn.a[ax], e.acc[ax]) && same(tn.j[ax], e.jrk[ax]), "end state ctrl %d", CONTROL);
      CHECK(same(max_abs_deriv_c<1, CONTROL>(c[ax], T), orc_primitive_max_vel(&pr, ax)), "max_vel ctrl %d", CONTROL);
      CHECK(same(max_abs_deriv_c<2, CONTROL>(c[ax], T), orc_primitive_max_acc(&pr, ax)), "max_acc ctrl %d", CONTROL);
      CHECK(same(max_abs_deriv_c<3, CONTROL>(c[ax], T), orc_primitive_max_jrk(&pr, ax)), "max_jrk ctrl %d", CONTROL);
    }
    CHECK((int)okc == orc_validate_primitive(&pr, 2.0, 1.0, 1.0), "validate vs oracle ctrl %d", CONTROL);
    CHECK(same(prim_J(CONTROL, c, T), orc_primitive_J(&pr, CONTROL)), "J ctrl %d", CONTROL);
    int32_t k1[13], k2[13];
    e.control = CONTROL;
    e.enable_t = 0;
    int n1 = orc_waypoint_key(&e, k1);
    state_key_c<CONTROL>(tn, k2);
    CHECK(n1 == key_len_c(CONTROL) && memcmp(k1, k2, 4 * n1) == 0, "key ctrl %d", CONTROL);
    int n3 = state_key(CONTROL, tn, k1);
    CHECK(n3 == n1 && memcmp(k1, k2, 4 * n1) == 0, "key generic ctrl %d", CONTROL);
  }
}

static void check_heuristic(int iters) {
  orc_planner *P = orc_create();
  int8_t map[8] = {0};
  int32_t dim[3] = {2, 2, 2};
  double origin[3] = {0, 0, 0};
  orc_set_map(P, map, dim, origin, 0.1);
  const int ctrls[3] = {CTRL_VEL, CTRL_ACC, CTRL_JRK};
  for (int it = 0; it < iters; it++) {
    int sc = ctrls[1 + rnd() % 2], gc = ctrls[rnd() % 3];
    if (gc > sc) gc = sc;
    double U[3] = {0, 0, 0};
    orc_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.control = sc; cfg.n_u = 1; cfg.U = U; cfg.dt = 1; cfg.v_max = (it % 11 == 0) ? 3.0 : 2.0; cfg.a_max = 1; cfg.j_max = 1;
    cfg.w = (it % 13 == 0) ? 1.0 : 10.0; cfg.eps = 1; cfg.tol_pos = 0.5; cfg.tol_vel = (it % 3 == 0) ? 0.3 : -1; cfg.tol_acc = -1;
    cfg.t_max = INFINITY; cfg.max_expand = -1; cfg.heur_ignore_dynamics = (it % 17 == 0);
    orc_set_config(P, &cfg);
    orc_waypoint s, g;
    memset(&s, 0, sizeof(s));
    memset(&g, 0, sizeof(g));
    s.control = sc;
    g.control = gc;
    bool close = it % 4 == 0;
    for (int i = 0; i < 3; i++) {
      g.pos[i] = round(uni(0, 50) * 100) / 100;
      s.pos[i] = close ? g.pos[i] + round(uni(-0.6, 0.6) * 100) / 100 : round(uni(0, 50) * 100) / 100;
      s.vel[i] = round(uni(-2, 2) * 10) / 10;
      s.acc[i] = round(uni(-1, 1) * 10) / 10;
      g.vel[i] = (it % 5 == 0) ? round(uni(-1, 1) * 10) / 10 : 0.0;
      g.acc[i] = 0.0;
    }
    if (it % 29 == 0) s = g, s.control = sc;
    orc_set_goal(P, &g);
    HeurParams hp;
    hp.w = cfg.w; hp.v_max = cfg.v_max; hp.heur_ignore_dynamics = cfg.heur_ignore_dynamics; hp.goal_control = gc;
    State ss, gs;
    memset(&ss, 0, sizeof(ss));
    memset(&gs, 0, sizeof(gs));
    for (int i = 0; i < 3; i++) {
      ss.p[i] = s.pos[i]; ss.v[i] = (sc & 2) ? s.vel[i] : 0; ss.a[i] = (sc & 4) ? s.acc[i] : 0;
      gs.p[i] = g.pos[i]; gs.v[i] = (gc & 2) ? g.vel[i] : 0; gs.a[i] = (gc & 4) ? g.acc[i] : 0;
    }
    // the oracle reads fields the goal's control does not enable; mirror what the API does (zeros)
    for (int i = 0; i < 3; i++) { if (!(gc & 2)) g.vel[i] = 0; if (!(gc & 4)) g.acc[i] = 0; if (!(sc & 4)) s.acc[i] = 0; }
    orc_set_goal(P, &g);
    hp.goal = gs;
    hp.goal_nkey = state_key(gc, gs, hp.goal_key);
    int32_t key[12];
    int nk = state_key(sc, ss, key);
    double h1 = get_heur(hp, sc, ss, key, nk), h2 = orc_heuristic(P, &s);
    CHECK(same(h1, h2), "heuristic sc %d gc %d: %.17g vs %.17g", sc, gc, h1, h2);
    CHECK((int)is_goal_state(ss, gs, gc, cfg.tol_pos, cfg.tol_vel, cfg.tol_acc) == orc_is_goal(P, &s), "is_goal");
  }
  orc_destroy(P);
}

template <int N>
static void check_roots(int iters) {
  for (int it = 0; it < iters; it++) {
    double a[N + 1];
    for (int i = 0; i <= N; i++) a[i] = (rnd() % 6 == 0) ? 0.0 : uni(-3, 3);
    double lo = (it % 3 == 0) ? 0.0 : uni(0, 2);
    double r1[8], r2[8];
    int n1 = poly_roots_above<N>(a, lo, r1);
    int n2 = orc_poly_roots_above(a, N, lo, r2);
    CHECK(n1 == n2, "root count deg %d: %d vs %d", N, n1, n2);
    for (int i = 0; i < n1 && i < n2; i++) CHECK(same(r1[i], r2[i]), "root deg %d", N);
  }
}

// float_to_cell_inv (reciprocal + two FMAs, used in the voxel sampling loop) == float_to_cell (`/`)
static void check_cell_quantisation(int iters) {
  const double rs[] = {0.1, 0.05, 0.2, 0.25, 0.3, 0.15, 0.01, 1.0 / 3, 0.07, 0.123456789, 1.0, 0.5};
  for (double r : rs) {
    const double inv = 1.0 / r;
    for (int it = 0; it < iters; it++) {
      double o = (it % 4 == 0) ? 0.0 : round(uni(-20, 20) * 10) / 10;
      double p;
      switch (it & 3) {
        case 0: p = uni(-100, 700); break;
        case 1: p = round(uni(-100, 700) * 20) / 20; break;      // lattice positions (multiples of 0.05)
        case 2: p = round(uni(-100, 700) * 100) * 0.01; break;
        default: p = o + (double)(int)uni(-100, 6000) * r; break;  // exactly on cell faces
      }
      CHECK(same(div_by_inv(p - o, r, inv), (p - o) / r), "div_by_inv r %.17g d %.17g", r, p - o);
      CHECK(float_to_cell_inv(p, o, r, inv) == float_to_cell(p, o, r), "float_to_cell_inv r %.17g p %.17g o %.17g", r, p, o);
    }
  }
}

// yaw: the product's det_sincos / yaw_end_ok (mplx_math.h) against the oracle's (orc_det_sincos, orc_validate_yaw): the same
// sequence of operations, so the validate_yaw decision falls the same way on host and device
static void check_yaw(int iters) {
  for (int it = 0; it < iters; it++) {
    const double x = (it % 11 == 0) ? special((int)(rnd() % 12)) : uni(-13.0, 13.0);
    double s0, c0, s1, c1;
    det_sincos(x, &s0, &c0);
    orc_det_sincos(x, &s1, &c1);
    CHECK(same(s0, s1) && same(c0, c1), "det_sincos(%.17g)", x);
    // one ACC primitive with a yaw channel: validate_yaw == both ends through yaw_end_ok
    orc_waypoint w;
    memset(&w, 0, sizeof(w));
    w.control = ORC_ACC | ORC_YAW;
    for (int i = 0; i < 3; i++) { w.pos[i] = uni(-2, 2); w.vel[i] = round(uni(-2, 2) * 10) / 10; }
    w.yaw = uni(-3.1, 3.1);
    const double u[3] = {(double)((int)(rnd() % 3) - 1), (double)((int)(rnd() % 3) - 1), 0.0}, uy = ((int)(rnd() % 3) - 1) * 0.5, T = 1.0, ymax = uni(0.1, 1.2);
    orc_primitive pr;
    orc_primitive_build_yaw(&w, u, uy, T, &pr);
    double sm, cm;
    det_sincos(ymax, &sm, &cm);
    const double yaw1 = normalize_yaw((uy * T + 0.0) + w.yaw);
    const bool ok = yaw_end_ok(vel_at_c<CTRL_ACC>(pr.c[0], 0.0), vel_at_c<CTRL_ACC>(pr.c[1], 0.0), normalize_yaw(w.yaw + 0.0), cm) &&
                    yaw_end_ok(vel_at_c<CTRL_ACC>(pr.c[0], T), vel_at_c<CTRL_ACC>(pr.c[1], T), yaw1, cm);
    CHECK(ok == (orc_validate_yaw(&pr, ymax) != 0), "validate_yaw yaw %.17g uy %g ymax %.17g", w.yaw, uy, ymax);
  }
}

int main() {
  check_yaw(200000);
  check_cell_quantisation(1000000);
  check_control<CTRL_VEL>(20000);
  check_control<CTRL_ACC>(20000);
  check_control<CTRL_JRK>(20000);
  check_control<CTRL_SNP>(20000);
  check_heuristic(20000);
  check_roots<1>(2000); check_roots<2>(4000); check_roots<3>(6000); check_roots<4>(8000); check_roots<5>(8000); check_roots<6>(10000);
  printf("%s (%d failures)\n", fails ? "FAILED" : "OK", fails);
  return fails ? 1 : 0;
}
