// mplx_host.cpp -- the host-side steps after the search, behind the C-ABI (include/mplx.h): TrajSolver refinement and
// trajectory sampling.  One implementation: the classes of the drop-in headers (include/mpl_shim/mpl_traj_solver,
// mpl_basis), which is what a C++ caller of the reference API uses directly.  No device code here.
#include "../../include/mplx.h"

#include <mpl_traj_solver/traj_solver.h>

static Primitive3D to_primitive(const mplx_primitive &p) {
  vec_E<Vec6f> cs(4);
  for (int ax = 0; ax < 3; ax++)
    for (int k = 0; k < 6; k++) cs[ax](k) = p.c[ax][k];
  for (int k = 0; k < 6; k++) cs[3](k) = p.cyaw[k];
  return Primitive3D(cs, p.t, (Control::Control)(p.control & 31));
}

extern "C" int mplx_traj_solve(int32_t control, int32_t n_wp, const mplx_waypoint *wps, const double *dts, mplx_primitive *prs) {
  if (n_wp < 2 || !wps || !dts || !prs) return MPLX_ERR_ARG;
  const int kind = control & 15;
  if (kind != MPLX_VEL && kind != MPLX_ACC && kind != MPLX_JRK) return MPLX_ERR_ARG;
  vec_E<Waypoint3D> ws;
  for (int i = 0; i < n_wp; i++) {
    Waypoint3D w((Control::Control)(wps[i].control & 31));
    for (int k = 0; k < 3; k++) { w.pos(k) = wps[i].pos[k]; w.vel(k) = wps[i].vel[k]; w.acc(k) = wps[i].acc[k]; w.jrk(k) = wps[i].jrk[k]; }
    ws.push_back(w);
  }
  PolySolver<3> solver(kind == MPLX_VEL ? 0 : kind == MPLX_ACC ? 1 : 2, kind == MPLX_VEL ? 1 : kind == MPLX_ACC ? 2 : 3);
  vec_E<Primitive3D> out;
  if (!solver.solve(ws, std::vector<decimal_t>(dts, dts + n_wp - 1)) || !solver.toPrimitives(out)) return MPLX_ERR_ARG;
  for (int i = 0; i + 1 < n_wp; i++) {
    mplx_primitive &p = prs[i];
    p = mplx_primitive();
    for (int ax = 0; ax < 3; ax++)
      for (int k = 0; k < 6; k++) p.c[ax][k] = out[i].pr(ax).coeff()(k);
    p.t = out[i].t();
    p.control = (int32_t)out[i].control();
  }
  return MPLX_OK;
}

extern "C" int mplx_traj_sample(int32_t n_prs, const mplx_primitive *prs, int32_t N, mplx_waypoint *out, double *yaw_dot) {
  if (n_prs <= 0 || !prs || N <= 0 || !out) return MPLX_ERR_ARG;
  vec_E<Primitive3D> segs;
  for (int i = 0; i < n_prs; i++) segs.push_back(to_primitive(prs[i]));
  const Trajectory3D traj(segs);
  const auto cmds = traj.sample(N);
  for (int i = 0; i <= N; i++) {
    mplx_waypoint &w = out[i];
    w = mplx_waypoint();
    for (int k = 0; k < 3; k++) { w.pos[k] = cmds[i].pos(k); w.vel[k] = cmds[i].vel(k); w.acc[k] = cmds[i].acc(k); w.jrk[k] = cmds[i].jrk(k); }
    w.yaw = cmds[i].yaw;
    w.t = cmds[i].t;
    w.control = prs[0].control;
    if (yaw_dot) yaw_dot[i] = cmds[i].yaw_dot;
  }
  return MPLX_OK;
}

extern "C" double mplx_traj_effort(int32_t n_prs, const mplx_primitive *prs, int32_t control) {
  double j = 0;
  for (int i = 0; i < n_prs; i++) {
    const Primitive3D p = to_primitive(prs[i]);
    j += (control & MPLX_YAW) && !(control & 15) ? p.Jyaw() : p.J((Control::Control)(control & 15));
  }
  return j;
}
