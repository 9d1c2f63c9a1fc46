"""CPU checker of the trajectory refinement (TEST INFRASTRUCTURE: only tests/ may import this).

Minimum-derivative piecewise polynomial through waypoints, solved as ONE equality-constrained QP over all segment
coefficients (KKT system, numpy) -- a formulation independent of the product's (include/mpl_shim/mpl_traj_solver/
poly_solver.h eliminates the coefficients and solves for the free waypoint derivatives).  PARITY UNPINNED: upstream's
mpl_traj_solver is in the absent motion_primitive_library submodule and the reference holds no golden output of
TrajSolver (map_planner_node.cpp:224-227 only prints J); what is checked is the mathematical statement -- the unique
minimiser of sum_seg int |d^r p|^2 under the waypoint and continuity constraints -- plus closed-form cases."""
import math

import numpy as np


def _falling(n, k):
    v = 1.0
    for m in range(k):
        v *= n - m
    return v


def solve(pos, fixed, dts, s, r):
    """pos/fixed: per waypoint, dict derivative order -> value (np array of the axes) for the FIXED derivatives;
    dts: segment times; s: smoothness order (derivatives 0..s continuous); r: minimised derivative.
    Returns coefficients [segment][n] -> axis vector (ascending monomials, N = 2 (s + 1))."""
    S, N = len(dts), 2 * (s + 1)
    dim = len(pos[0])
    Q = np.zeros((S * N, S * N))
    rows, rhs = [], []

    def drow(seg, k, t):
        row = np.zeros(S * N)
        for n in range(k, N):
            row[seg * N + n] = _falling(n, k) * t ** (n - k)
        return row

    for i, T in enumerate(dts):
        for a in range(r, N):
            for b in range(r, N):
                Q[i * N + a, i * N + b] = _falling(a, r) * _falling(b, r) * T ** (a + b - 2 * r + 1) / (a + b - 2 * r + 1)
    for w in range(S + 1):
        for k in range(s + 1):
            left = drow(w - 1, k, dts[w - 1]) if w > 0 else None
            right = drow(w, k, 0.0) if w < S else None
            if k in fixed[w]:
                for row in (left, right):
                    if row is not None:
                        rows.append(row)
                        rhs.append(np.asarray(fixed[w][k], dtype=float))
            elif left is not None and right is not None:
                rows.append(left - right)
                rhs.append(np.zeros(dim))
    Cm, b = np.array(rows), np.array(rhs)
    m = Cm.shape[0]
    K = np.block([[2 * Q, Cm.T], [Cm, np.zeros((m, m))]])
    sol = np.linalg.lstsq(K, np.vstack([np.zeros((S * N, dim)), b]), rcond=None)[0]
    return sol[:S * N].reshape(S, N, dim)


def cost(coeff, dts, r):
    tot = 0.0
    S, N, dim = coeff.shape
    for i, T in enumerate(dts):
        for a in range(r, N):
            for b in range(r, N):
                tot += _falling(a, r) * _falling(b, r) * T ** (a + b - 2 * r + 1) / (a + b - 2 * r + 1) * float(coeff[i, a] @ coeff[i, b])
    return tot


def primitive_to_monomials(c6):
    """Primitive coefficients (c0/120 t^5 + ... + c5) -> ascending monomials a_n = c(5 - n) / n!"""
    return np.array([c6[5 - n] / math.factorial(n) for n in range(6)])
