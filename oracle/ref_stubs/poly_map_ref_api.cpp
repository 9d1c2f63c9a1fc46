// oracle/_ref/libpolymap_ref.so -- the reference's OWN moving-obstacle environment, compiled from where it lies.
//
// TEST INFRASTRUCTURE ONLY (see oracle/mpl_oracle.h).  The PolyMap planner is the one search environment whose
// arithmetic is vendored in the reference tree:
//   mpl_external_planner/include/mpl_external_planner/poly_map_planner/env_poly_map.h      (get_succ, intrinsic cost)
//   .../poly_map_planner/poly_map_util.h                                                   (isInside, isFree(pt,t), isFree(pr,t))
//   .../poly_map_planner/primitive_geometry_utils.h                                        (the three collide())
//   .../poly_map_planner/simple_obstacle.h                                                 (static / linear / nonlinear obstacles)
// This file only wraps them in a C interface; `make -C oracle ref` compiles it against those headers (never copied)
// and against stand-ins for what they include from un-vendored submodules: mpl_basis / mpl_planner (include/mpl_shim)
// and DecompUtil's polyhedron.h (include/mpl_shim/decomp_geometry, labelled UNVERIFIED).  2-D, like the
// multi-robot configuration (multi_robot_node.cpp).  A best-first search over this environment is added below so
// that whole plans can be checked too: GraphSearch itself is NOT vendored, so that loop is the same restatement
// as oracle/mpl_oracle.c's orc_plan (same order, same tie-breaking), only templated over the reference's get_succ.
#include <mpl_external_planner/poly_map_planner/poly_map_planner.h>

#include <cstring>
#include <map>
#include <queue>

namespace {
struct Ref {
  std::shared_ptr<PolyMapUtil<2>> map_util;
  std::shared_ptr<MPL::env_poly_map<2>> env;
  vec_E<PolyhedronObstacle2D> st;
  vec_E<PolyhedronLinearObstacle2D> lin;
  vec_E<PolyhedronNonlinearObstacle2D> nl;
};
Polyhedron2D poly_of(int n_hp, const double *hp) {
  Polyhedron2D p;
  for (int i = 0; i < n_hp; i++) p.add(Hyperplane2D(Vec2f(hp[4 * i], hp[4 * i + 1]), Vec2f(hp[4 * i + 2], hp[4 * i + 3])));
  return p;
}
Waypoint2D wp_of(const double *s /* pos2 vel2 acc2 jrk2 t */, int control) {
  Waypoint2D w((Control::Control)control);
  w.pos = Vec2f(s[0], s[1]); w.vel = Vec2f(s[2], s[3]); w.acc = Vec2f(s[4], s[5]); w.jrk = Vec2f(s[6], s[7]);
  w.t = s[8];
  return w;
}
void wp_to(const Waypoint2D &w, double *s) {
  s[0] = w.pos(0); s[1] = w.pos(1); s[2] = w.vel(0); s[3] = w.vel(1); s[4] = w.acc(0); s[5] = w.acc(1); s[6] = w.jrk(0); s[7] = w.jrk(1);
  s[8] = w.t;
}
void sync(Ref *r) {
  r->map_util->setStaticObstacle(r->st);
  r->map_util->setLinearObstacle(r->lin);
  r->map_util->setNonlinearObstacle(r->nl);
}
}  // namespace

extern "C" {
void *refpoly_create(double ox, double oy, double dx, double dy) {
  Ref *r = new Ref();
  r->map_util.reset(new PolyMapUtil<2>());
  r->map_util->setBoundingBox(Vec2f(ox, oy), Vec2f(dx, dy));
  r->env.reset(new MPL::env_poly_map<2>(r->map_util));
  return r;
}
void refpoly_destroy(void *h) { delete (Ref *)h; }
void refpoly_set_start_time(void *h, double t) { ((Ref *)h)->map_util->setStartTime(t); }
void refpoly_clear_obstacles(void *h) { Ref *r = (Ref *)h; r->st.clear(); r->lin.clear(); r->nl.clear(); sync(r); }
// hp: n_hp x {px, py, nx, ny}
void refpoly_add_static(void *h, int n_hp, const double *hp, double px, double py) {
  Ref *r = (Ref *)h;
  r->st.push_back(PolyhedronObstacle2D(poly_of(n_hp, hp), Vec2f(px, py)));
  sync(r);
}
void refpoly_add_linear(void *h, int n_hp, const double *hp, double px, double py, double vx, double vy, double cov_v) {
  Ref *r = (Ref *)h;
  PolyhedronLinearObstacle2D o(poly_of(n_hp, hp), Vec2f(px, py), Vec2f(vx, vy));
  o.set_cov_v(cov_v);
  r->lin.push_back(o);
  sync(r);
}
// segs: n_seg x {cx[6], cy[6], T}; the obstacle follows that trajectory from its local time start_t on
void refpoly_add_nonlinear(void *h, int n_hp, const double *hp, int n_seg, const double *segs, int control, double start_t, int dis_front, int dis_back) {
  Ref *r = (Ref *)h;
  vec_E<Primitive2D> prs;
  for (int i = 0; i < n_seg; i++) {
    vec_E<Vec6f> cs(2);
    for (int k = 0; k < 6; k++) { cs[0](k) = segs[13 * i + k]; cs[1](k) = segs[13 * i + 6 + k]; }
    prs.push_back(Primitive2D(cs, segs[13 * i + 12], (Control::Control)control));
  }
  PolyhedronNonlinearObstacle2D o(poly_of(n_hp, hp), Trajectory2D(prs), start_t);
  o.disappear_front_ = dis_front != 0;
  o.disappear_back_ = dis_back != 0;
  r->nl.push_back(o);
  sync(r);
}
void refpoly_set_env(void *h, int n_u, const double *U, double dt, double v_max, double a_max, double j_max, double w) {
  Ref *r = (Ref *)h;
  vec_E<VecDf> Us;
  for (int i = 0; i < n_u; i++) Us.push_back(Vec2f(U[2 * i], U[2 * i + 1]));
  r->env->set_u(Us);
  r->env->set_dt(dt); r->env->set_v_max(v_max); r->env->set_a_max(a_max); r->env->set_j_max(j_max); r->env->set_w(w);
}
int refpoly_is_inside(void *h, double x, double y) { return ((Ref *)h)->map_util->isInside(Vec2f(x, y)) ? 1 : 0; }
int refpoly_is_free_point(void *h, double x, double y, double t) { return ((Ref *)h)->map_util->isFree(Vec2f(x, y), t) ? 1 : 0; }
// env_poly_map::get_succ: state = pos2 vel2 acc2 jrk2 t; out arrays sized n_u (succ: n x 9); returns the number emitted
int refpoly_get_succ(void *h, const double *state, int control, double *succ, double *cost, int *action) {
  Ref *r = (Ref *)h;
  vec_E<Waypoint2D> s;
  std::vector<decimal_t> c;
  std::vector<int> a;
  r->env->get_succ(wp_of(state, control), s, c, a);
  for (size_t i = 0; i < s.size(); i++) { wp_to(s[i], succ + 9 * i); cost[i] = c[i]; action[i] = a[i]; }
  return (int)s.size();
}

// ---- best-first search over the reference environment (GraphSearch::Astar restated, see the header comment).
// Total order of OPEN: (f, g, creation id); goal test after the expansion; max_expand then empty-OPEN.
// heur_mode != 0: w * |dp|_inf / v_max (setHeurIgnoreDynamics(true); v_max <= 0: w * |dp|_inf).
// heur_mode == 0: the dynamics-aware env_base::cal_heur -- what the reference's robots plan with (robot.hpp:109-122 sets
// neither setHeurIgnoreDynamics nor setMaxNum).  Its closed forms live in the CPU oracle (oracle/mpl_oracle.c cal_heur,
// pinned against a numerical optimal-control solve in tests/test_oracle_kat.py); this library does not link the oracle:
// the caller hands over the address of orc_heuristic and an orc_planner configured with the same w / v_max / goal
// (refpoly_set_heuristic), and the search calls through it with the state's z components zero.
// is_goal: |dp|_inf <= tol_pos.
struct PNode { Waypoint2D coord; double g, h; int closed, opened; std::vector<int> pred, pact; std::vector<double> pcost; };
static std::vector<int> g_exp, g_traj_nodes, g_traj_act;
static std::vector<PNode> g_nodes;
static double g_cost;
typedef double (*orc_heur_fn)(const void *planner, const void *waypoint);
static orc_heur_fn g_heur_fn = nullptr;
static const void *g_heur_planner = nullptr;
// layout of orc_waypoint (oracle/mpl_oracle.h); restated here so that this file needs nothing of the oracle's but the call
struct OrcWaypoint { double pos[3], vel[3], acc[3], jrk[3]; double yaw, t; int32_t control, enable_t; };
void refpoly_set_heuristic(void *fn, const void *planner) { g_heur_fn = (orc_heur_fn)fn; g_heur_planner = planner; }
int refpoly_plan(void *h, const double *start, const double *goal, int control, double eps, double tol_pos, int max_expand, int heur_mode) {
  Ref *r = (Ref *)h;
  g_nodes.clear(); g_exp.clear(); g_traj_nodes.clear(); g_traj_act.clear();
  g_cost = std::numeric_limits<double>::infinity();
  Waypoint2D s = wp_of(start, control), g = wp_of(goal, control);
  s.enable_t = true;  // env_poly_map keys its successors with time (env_poly_map.h:64)
  if (!r->env->is_free(s.pos)) return 2;  // PlannerBase::plan: ENV_->is_free(start.pos) = inside the bounding box (env_poly_map.h:33)
  const double v_max = r->env->v_max_, w = r->env->w_;
  auto heur = [&](const Waypoint2D &x) {
    if (heur_mode == 0) {  // dynamics-aware: the oracle's cal_heur (states are time-keyed: never equal to the goal key)
      OrcWaypoint o = OrcWaypoint();
      for (int i = 0; i < 2; i++) { o.pos[i] = x.pos(i); o.vel[i] = x.vel(i); o.acc[i] = x.acc(i); o.jrk[i] = x.jrk(i); }
      o.t = x.t;
      o.control = control;
      o.enable_t = 1;
      return g_heur_fn(g_heur_planner, &o);
    }
    const double d = std::max(std::fabs(x.pos(0) - g.pos(0)), std::fabs(x.pos(1) - g.pos(1)));
    return v_max > 0 ? w * d / v_max : w * d;
  };
  if (heur_mode == 0 && !g_heur_fn) return -1;  // refpoly_set_heuristic first
  auto is_goal = [&](const Waypoint2D &x) { return std::max(std::fabs(x.pos(0) - g.pos(0)), std::fabs(x.pos(1) - g.pos(1))) <= tol_pos; };
  if (is_goal(s)) { g_cost = 0; return 0; }
  std::map<std::vector<int>, int> table;
  auto less = [&](int a, int b) {  // a after b in the queue?
    const double fa = g_nodes[a].g + eps * g_nodes[a].h, fb = g_nodes[b].g + eps * g_nodes[b].h;
    if (fa != fb) return fa > fb;
    if (g_nodes[a].g != g_nodes[b].g) return g_nodes[a].g > g_nodes[b].g;
    return a > b;
  };
  // lazy-deletion heap keyed by (f, g, id) snapshots: stale entries are skipped at pop time
  struct E { double f, g; int id; };
  auto ecmp = [](const E &a, const E &b) { if (a.f != b.f) return a.f > b.f; if (a.g != b.g) return a.g > b.g; return a.id > b.id; };
  std::priority_queue<E, std::vector<E>, decltype(ecmp)> open(ecmp);
  (void)less;
  g_nodes.push_back(PNode{s, 0.0, eps == 0 ? 0.0 : heur(s), 0, 1, {}, {}, {}});
  table[s.key()] = 0;
  open.push(E{eps * g_nodes[0].h, 0.0, 0});
  int status = 0, curr = -1, it = 0;
  vec_E<Waypoint2D> succ;
  std::vector<decimal_t> cost;
  std::vector<int> act;
  for (;;) {
    for (;;) {  // pop the smallest live entry
      if (open.empty()) { curr = -1; break; }
      const E e = open.top();
      open.pop();
      if (g_nodes[e.id].closed || g_nodes[e.id].g != e.g) continue;
      curr = e.id;
      break;
    }
    if (curr < 0) { status = 1; break; }
    it++;
    g_nodes[curr].closed = 1;
    g_exp.push_back(curr);
    const Waypoint2D cw = g_nodes[curr].coord;
    r->env->get_succ(cw, succ, cost, act);
    for (size_t k = 0; k < succ.size(); k++) {
      if (std::isinf(cost[k])) continue;  // (deviation D7 of oracle/mpl_oracle.h: blocked successors are not materialised)
      Waypoint2D tn = succ[k];
      tn.control = cw.control;
      const auto key = tn.key();
      int id;
      auto f = table.find(key);
      if (f == table.end()) {
        id = (int)g_nodes.size();
        g_nodes.push_back(PNode{tn, std::numeric_limits<double>::infinity(), eps == 0 ? 0.0 : heur(tn), 0, 0, {}, {}, {}});
        table[key] = id;
      } else {
        id = f->second;
      }
      g_nodes[id].pred.push_back(curr); g_nodes[id].pact.push_back(act[k]); g_nodes[id].pcost.push_back(cost[k]);
      const double tg = g_nodes[curr].g + cost[k];
      if (tg < g_nodes[id].g) {
        g_nodes[id].g = tg;
        g_nodes[id].closed = 0;  // (re-open: deviation D6)
        g_nodes[id].opened = 1;
        open.push(E{tg + eps * g_nodes[id].h, tg, id});
      }
    }
    if (is_goal(g_nodes[curr].coord)) break;
    if (max_expand > 0 && it >= max_expand) { status = 3; break; }
  }
  if (status != 0) return status;
  // recoverTraj: minimise g(pred) + cost, ties -> larger g(pred), then the oldest record
  int node = curr;
  g_traj_nodes.push_back(node);
  while (!g_nodes[node].pred.empty()) {
    int best = -1;
    double min_rhs = std::numeric_limits<double>::infinity(), min_g = std::numeric_limits<double>::infinity();
    for (size_t e = 0; e < g_nodes[node].pred.size(); e++) {
      const double gp = g_nodes[g_nodes[node].pred[e]].g, rhs = gp + g_nodes[node].pcost[e];
      if (min_rhs > rhs) { min_rhs = rhs; min_g = gp; best = (int)e; }
      else if (min_rhs == rhs && min_g < gp) { min_g = gp; best = (int)e; }
    }
    if (best < 0) return 1;
    g_traj_act.push_back(g_nodes[node].pact[best]);
    node = g_nodes[node].pred[best];
    g_traj_nodes.push_back(node);
    if (node == 0) break;
  }
  g_cost = g_nodes[curr].g;
  return 0;
}
int refpoly_num_expanded() { return (int)g_exp.size(); }
int refpoly_num_nodes() { return (int)g_nodes.size(); }
double refpoly_traj_cost() { return g_cost; }
int refpoly_traj_len() { return (int)g_traj_act.size(); }
// expansion order (node ids), path (goal -> start order reversed to start -> goal), node states
void refpoly_get_expanded(int *ids) { for (size_t i = 0; i < g_exp.size(); i++) ids[i] = g_exp[i]; }
void refpoly_get_traj(int *node_ids, int *actions) {
  const size_t n = g_traj_act.size();
  for (size_t i = 0; i <= n && !g_traj_nodes.empty(); i++) node_ids[i] = g_traj_nodes[n - i];
  for (size_t i = 0; i < n; i++) actions[i] = g_traj_act[n - 1 - i];
}
void refpoly_get_node(int id, double *state, double *g, double *hh) { wp_to(g_nodes[id].coord, state); *g = g_nodes[id].g; *hh = g_nodes[id].h; }

// ------------------------------------------------------------------------------------------------------------------------------
// LPA* over the reference environment (round 6): PlannerBase::plan with setLPAstar(true), PolyMapPlanner::updateNodes
// (poly_map_planner.h:61-93) and getSubStateSpace, as poly_map_replanner_node.cpp:123-186,231 drives them.  The search and the state
// space (graph_search.h LPAstar, state_space.h) are un-vendored [UNVERIFIED]; this is the restatement of oracle/mpl_oracle_lpa.inc
// (choices L1, L4, L6, L7) on the COMPILED reference environment, with two differences that follow the in-tree code:
//   * every successor get_succ emits is a state with a predecessor entry -- also the ones whose primitive is blocked (cost +inf), as
//     poly_map_planner.h:70-86 walks them (pred_coord / pred_action_id / pred_action_cost, "if (!isinf(cost))" / "if (isinf(cost))"):
//     there is no blocked log here (deviation D7 of the voxel oracle does not apply);
//   * updateNodes is the in-tree loop: every entry of every state re-tested with forward_action + isFree(pr, pred.t); what became
//     blocked / free is collected (getBlockedPrimitives / getClearedPrimitives), increaseCost / decreaseCost = the entry's cost turns
//     +inf / back to calculate_intrinsic_cost(pr), and the rhs of every state whose entries changed is recomputed.
//   * getSubStateSpace(k): by planning afresh from the k-th state of the last trajectory (L5b) -- the only realisation here.
// Entries are numbered in creation order (expansion order, then the order get_succ emits): the number both sides report.
struct LNode {
  Waypoint2D coord;
  double g, rhs, h;
  int closed, opened, built;
  std::vector<int> pred, pact, pentry, pblk;
  std::vector<double> pcost;  // calculate_intrinsic_cost(pr), whether blocked or not
};
struct LEntry { double k, kg; int id; };
struct LSpace {
  std::vector<LNode> nodes;
  std::map<std::vector<int>, int> table;
  int root = -1, goal_id = -1, n_entries = 0, iterations = 0, valid = 0;
  double goal[9] = {0}, cost = std::numeric_limits<double>::infinity();
  std::vector<int> expanded, traj_nodes, traj_act;
  std::vector<int> changed_entry, changed_blocked;
  std::vector<int> echild, eparent, eaction;  // entry -> (child, parent, action)
};
static LSpace LS;
static const double L_INF = std::numeric_limits<double>::infinity();

static double l_rhs_of(const LNode &nd) {
  double rhs = L_INF;
  for (size_t e = 0; e < nd.pred.size(); e++) {
    if (nd.pblk[e]) continue;
    const double v = LS.nodes[nd.pred[e]].g + nd.pcost[e];
    if (v < rhs) rhs = v;
  }
  return rhs;
}
int refpoly_lpa_initialized() { return LS.valid; }
void refpoly_lpa_reset() { LS = LSpace(); }

int refpoly_lpa_plan(void *h, const double *start, const double *goal, int control, double eps, double tol_pos, int max_expand, int heur_mode) {
  Ref *r = (Ref *)h;
  LS.cost = L_INF;
  LS.expanded.clear(); LS.traj_nodes.clear(); LS.traj_act.clear();
  LS.iterations = 0;
  Waypoint2D s = wp_of(start, control), g = wp_of(goal, control);
  s.enable_t = true;
  if (!r->env->is_free(s.pos)) return 2;
  const double v_max = r->env->v_max_, w = r->env->w_;
  auto heur = [&](const Waypoint2D &x) {
    if (eps == 0) return 0.0;
    if (heur_mode == 0) {
      OrcWaypoint o = OrcWaypoint();
      for (int i = 0; i < 2; i++) { o.pos[i] = x.pos(i); o.vel[i] = x.vel(i); o.acc[i] = x.acc(i); o.jrk[i] = x.jrk(i); }
      o.t = x.t;
      o.control = control;
      o.enable_t = 1;
      return g_heur_fn(g_heur_planner, &o);
    }
    const double d = std::max(std::fabs(x.pos(0) - g.pos(0)), std::fabs(x.pos(1) - g.pos(1)));
    return v_max > 0 ? w * d / v_max : w * d;
  };
  if (heur_mode == 0 && !g_heur_fn) return -1;
  auto is_goal = [&](const Waypoint2D &x) { return std::max(std::fabs(x.pos(0) - g.pos(0)), std::fabs(x.pos(1) - g.pos(1))) <= tol_pos; };
  bool same_goal = true;
  for (int i = 0; i < 8; i++) same_goal = same_goal && LS.goal[i] == goal[i];
  if (LS.valid && !same_goal) LS.valid = 0;  // L6
  for (int i = 0; i < 9; i++) LS.goal[i] = goal[i];
  if (is_goal(s)) { LS.cost = 0; return 0; }
  if (LS.valid) {  // L6: the start must be the current root
    auto f = LS.table.find(s.key());
    if (f == LS.table.end() || f->second != LS.root) LS.valid = 0;
  }
  if (!LS.valid) {
    LS.nodes.clear(); LS.table.clear(); LS.echild.clear(); LS.eparent.clear(); LS.eaction.clear();
    LS.n_entries = 0;
    LS.nodes.push_back(LNode{s, L_INF, 0.0, heur(s), 0, 1, 0, {}, {}, {}, {}, {}});
    LS.table[s.key()] = 0;
    LS.root = 0;
    LS.goal_id = -1;
    LS.valid = 1;
  }
  auto key_of = [&](const LNode &nd) { return std::min(nd.g, nd.rhs) + eps * nd.h; };
  auto ecmp = [](const LEntry &a, const LEntry &b) { if (a.k != b.k) return a.k > b.k; if (a.kg != b.kg) return a.kg > b.kg; return a.id > b.id; };
  std::priority_queue<LEntry, std::vector<LEntry>, decltype(ecmp)> open(ecmp);
  auto push = [&](int id) { const LNode &nd = LS.nodes[id]; open.push(LEntry{key_of(nd), std::min(nd.g, nd.rhs), id}); };
  auto entry_valid = [&](const LEntry &e) {
    const LNode &nd = LS.nodes[e.id];
    if (nd.g == nd.rhs) return false;
    return e.k == key_of(nd) && e.kg == std::min(nd.g, nd.rhs);
  };
  auto update_node = [&](int id, bool g_changed) {  // oracle/mpl_oracle_lpa.inc lpa_update_node
    LNode &nd = LS.nodes[id];
    const double old_rhs = nd.rhs;
    if (id != LS.root) nd.rhs = l_rhs_of(nd);
    if (nd.g != nd.rhs) {
      const bool had_entry = !g_changed && nd.opened && !nd.closed && nd.rhs == old_rhs;
      nd.opened = 1;
      nd.closed = 0;
      if (!had_entry) push(id);
    } else if (nd.opened && !nd.closed) {
      nd.closed = 1;
    }
  };
  for (size_t i = 0; i < LS.nodes.size(); i++)  // L1: OPEN = the inconsistent states, rebuilt
    if (LS.nodes[i].g != LS.nodes[i].rhs) push((int)i);
  int gid = -1;  // L7
  for (size_t i = 0; i < LS.nodes.size(); i++) {
    const LNode &nd = LS.nodes[i];
    if (nd.g != nd.rhs || std::isinf(nd.g) || !is_goal(nd.coord)) continue;
    if (gid < 0) { gid = (int)i; continue; }
    const LNode &b = LS.nodes[gid];
    const double k = key_of(nd), kb = key_of(b);
    if (k < kb || (k == kb && nd.g < b.g)) gid = (int)i;
  }
  if (gid < 0 && LS.goal_id >= 0 && LS.goal_id < (int)LS.nodes.size() && is_goal(LS.nodes[LS.goal_id].coord)) gid = LS.goal_id;
  vec_E<Waypoint2D> succ;
  std::vector<decimal_t> cost;
  std::vector<int> act;
  int status = 0;
  for (;;) {
    while (!open.empty() && !entry_valid(open.top())) open.pop();
    const bool gcons = gid < 0 || LS.nodes[gid].g == LS.nodes[gid].rhs;
    const double kgoal = gid < 0 ? L_INF : key_of(LS.nodes[gid]);
    if (open.empty()) {
      if (!(gid >= 0 && gcons && !std::isinf(LS.nodes[gid].g))) status = 1;
      break;
    }
    if (!(open.top().k < kgoal || !gcons)) break;
    LS.iterations++;
    const int u = open.top().id;
    open.pop();
    LS.nodes[u].opened = 1;
    LS.nodes[u].closed = 1;
    LS.expanded.push_back(u);
    if (LS.nodes[u].g > LS.nodes[u].rhs) {
      LS.nodes[u].g = LS.nodes[u].rhs;
    } else {
      LS.nodes[u].g = L_INF;
      update_node(u, true);
    }
    const Waypoint2D cw = LS.nodes[u].coord;
    const bool first = !LS.nodes[u].built;
    r->env->get_succ(cw, succ, cost, act);
    std::vector<int> kids;
    for (size_t k = 0; k < succ.size(); k++) {
      Waypoint2D tn = succ[k];
      tn.control = cw.control;
      auto f = LS.table.find(tn.key());
      int id = f == LS.table.end() ? -1 : f->second;
      if (first) {
        if (id < 0) {
          id = (int)LS.nodes.size();
          LS.nodes.push_back(LNode{tn, L_INF, L_INF, heur(tn), 0, 0, 0, {}, {}, {}, {}, {}});
          LS.table[tn.key()] = id;
        }
        Primitive2D pr;
        r->env->forward_action(cw, act[k], pr);
        LNode &c = LS.nodes[id];
        c.pred.push_back(u); c.pact.push_back(act[k]); c.pentry.push_back(LS.n_entries); c.pblk.push_back(std::isinf(cost[k]) ? 1 : 0);
        c.pcost.push_back(r->env->calculate_intrinsic_cost(pr));
        LS.echild.push_back(id); LS.eparent.push_back(u); LS.eaction.push_back(act[k]);
        LS.n_entries++;
      } else if (id < 0) {
        continue;
      }
      if (std::isinf(cost[k])) continue;  // (a blocked entry cannot lower the successor's rhs)
      if (std::find(kids.begin(), kids.end(), id) == kids.end()) kids.push_back(id);
    }
    LS.nodes[u].built = 1;
    for (int id : kids) update_node(id, false);
    if (is_goal(LS.nodes[u].coord) && !std::isinf(LS.nodes[u].g)) gid = u;
    if (max_expand > 0 && LS.iterations >= max_expand) { status = 3; break; }
  }
  LS.goal_id = gid;
  if (status != 0) return status;
  // recoverTraj (oracle/mpl_oracle.c recover_traj): minimise g(pred) + cost over the non-blocked entries, ties -> larger g(pred), then the oldest
  int node = gid;
  LS.traj_nodes.push_back(node);
  while (!LS.nodes[node].pred.empty()) {
    int best = -1;
    double min_rhs = L_INF, min_g = L_INF;
    const LNode &nd = LS.nodes[node];
    for (size_t e = 0; e < nd.pred.size(); e++) {
      const double gp = LS.nodes[nd.pred[e]].g, c = nd.pblk[e] ? L_INF : nd.pcost[e];
      if (min_rhs > gp + c) { min_rhs = gp + c; min_g = gp; best = (int)e; }
      else if (!std::isinf(c) && min_rhs == gp + c && min_g < gp) { min_g = gp; best = (int)e; }
    }
    if (best < 0) return 1;
    LS.traj_act.push_back(nd.pact[best]);
    node = nd.pred[best];
    LS.traj_nodes.push_back(node);
    if (node == LS.root) break;
    if (LS.traj_nodes.size() > LS.nodes.size() + 1) return 1;
  }
  if (node != LS.root) return 1;
  LS.cost = LS.nodes[gid].g;
  return 0;
}

// PolyMapPlanner::updateNodes (poly_map_planner.h:61-93) after the obstacles / the start time of the environment were changed
int refpoly_lpa_update_nodes(void *h, int *n_blocked, int *n_cleared) {
  Ref *r = (Ref *)h;
  LS.changed_entry.clear(); LS.changed_blocked.clear();
  int nb = 0, nc = 0;
  if (!LS.valid) { *n_blocked = 0; *n_cleared = 0; return 0; }
  std::vector<int> dirty(LS.nodes.size(), 0);
  for (size_t i = 0; i < LS.nodes.size(); i++) {
    LNode &nd = LS.nodes[i];
    for (size_t e = 0; e < nd.pred.size(); e++) {
      Primitive2D pr;
      const Waypoint2D &pc = LS.nodes[nd.pred[e]].coord;
      r->env->forward_action(pc, nd.pact[e], pr);
      const bool free_ = r->map_util->isFree(pr, pc.t);
      if (!free_ && !nd.pblk[e]) { nd.pblk[e] = 1; dirty[i] = 1; nb++; LS.changed_entry.push_back(nd.pentry[e]); LS.changed_blocked.push_back(1); }
      else if (free_ && nd.pblk[e]) { nd.pblk[e] = 0; dirty[i] = 1; nc++; LS.changed_entry.push_back(nd.pentry[e]); LS.changed_blocked.push_back(0); }
    }
  }
  for (size_t i = 0; i < LS.nodes.size(); i++)  // increaseCost / decreaseCost: the look-ahead value of every state whose entries changed
    if (dirty[i] && (int)i != LS.root) LS.nodes[i].rhs = l_rhs_of(LS.nodes[i]);
  *n_blocked = nb; *n_cleared = nc;
  return 0;
}
// getSubStateSpace(k) (poly_map_replanner_node.cpp:231) by planning afresh from the k-th state of the last trajectory (L5b)
int refpoly_lpa_sub_state_space(void *h, int k, int control, double eps, double tol_pos, int max_expand, int heur_mode) {
  if (!LS.valid || LS.traj_act.empty() || k < 0 || k > (int)LS.traj_act.size()) return 0;
  const int n = (int)LS.traj_act.size();
  double s[9], goal[9];
  wp_to(LS.nodes[LS.traj_nodes[n - k]].coord, s);  // (traj_nodes is goal -> start)
  for (int i = 0; i < 9; i++) goal[i] = LS.goal[i];
  LS.valid = 0;
  refpoly_lpa_plan(h, s, goal, control, eps, tol_pos, max_expand, heur_mode);
  LS.expanded.clear(); LS.traj_nodes.clear(); LS.traj_act.clear();  // (not a plan: no expansion record, no trajectory of its own)
  LS.cost = L_INF;
  return 0;
}
int refpoly_lpa_num_nodes() { return (int)LS.nodes.size(); }
int refpoly_lpa_num_entries() { return LS.n_entries; }
int refpoly_lpa_iterations() { return LS.iterations; }
double refpoly_lpa_cost() { return LS.cost; }
int refpoly_lpa_traj_len() { return (int)LS.traj_act.size(); }
void refpoly_lpa_get_expanded(int *ids) { for (size_t i = 0; i < LS.expanded.size(); i++) ids[i] = LS.expanded[i]; }
void refpoly_lpa_get_traj(int *node_ids, int *actions) {
  const size_t n = LS.traj_act.size();
  for (size_t i = 0; i <= n && !LS.traj_nodes.empty(); i++) node_ids[i] = LS.traj_nodes[n - i];
  for (size_t i = 0; i < n; i++) actions[i] = LS.traj_act[n - 1 - i];
}
// per state: pos2 vel2 acc2 jrk2 t | g rhs h | closed opened built
void refpoly_lpa_get_node(int id, double *state, double *vals, int *flags) {
  const LNode &nd = LS.nodes[id];
  wp_to(nd.coord, state);
  vals[0] = nd.g; vals[1] = nd.rhs; vals[2] = nd.h;
  flags[0] = nd.closed; flags[1] = nd.opened; flags[2] = nd.built;
}
// per entry, in creation order: child, parent, action, blocked
void refpoly_lpa_get_entries(int *child, int *parent, int *action, int *blocked) {
  for (int e = 0; e < LS.n_entries; e++) { child[e] = LS.echild[e]; parent[e] = LS.eparent[e]; action[e] = LS.eaction[e]; blocked[e] = 0; }
  for (const LNode &nd : LS.nodes)
    for (size_t e = 0; e < nd.pred.size(); e++) blocked[nd.pentry[e]] = nd.pblk[e];
}
int refpoly_lpa_num_changed() { return (int)LS.changed_entry.size(); }
void refpoly_lpa_get_changed(int *entry, int *now_blocked) {
  for (size_t i = 0; i < LS.changed_entry.size(); i++) { entry[i] = LS.changed_entry[i]; now_blocked[i] = LS.changed_blocked[i]; }
}
}
