"""Independent NumPy checks for the oracle (test helper, not shipped, not the oracle itself).

A third formulation of the rigid-body dynamics -- world-frame classical Newton-Euler about each
link's centre of mass, plus the mechanical energy for a finite-difference power balance -- so the
oracle (body-frame spatial algebra, iDynTree-shaped) and the HIP kernels (base-frame composite
regressor) are both checked against something that shares no code or convention with them.
Vectorised over samples.
"""
from __future__ import annotations

import numpy as np


def rpy_R(rpy):
    """(S,3) -> (S,3,3) Rz(y)Ry(p)Rx(r)."""
    r, p, y = rpy[:, 0], rpy[:, 1], rpy[:, 2]
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    R = np.empty((rpy.shape[0], 3, 3))
    R[:, 0, 0] = cy * cp
    R[:, 0, 1] = cy * sp * sr - sy * cr
    R[:, 0, 2] = cy * sp * cr + sy * sr
    R[:, 1, 0] = sy * cp
    R[:, 1, 1] = sy * sp * sr + cy * cr
    R[:, 1, 2] = sy * sp * cr - cy * sr
    R[:, 2, 0] = -sp
    R[:, 2, 1] = cp * sr
    R[:, 2, 2] = cp * cr
    return R


def axis_angle_R(s, q):
    """unit axis s (3,), angles q (S,) -> (S,3,3)."""
    K = np.array([[0, -s[2], s[1]], [s[2], 0, -s[0]], [-s[1], s[0], 0]])
    return np.eye(3)[None] + np.sin(q)[:, None, None] * K[None] + (1 - np.cos(q))[:, None, None] * (K @ K)[None]


def link_inertials(topo):
    """mass (L,), com (L,3), inertia about the COM in link axes (L,3,3) from the 10 parameters."""
    P = topo.params
    m = P[:, 0].copy()
    com = np.zeros((topo.num_links, 3))
    Ic = np.zeros((topo.num_links, 3, 3))
    for l in range(topo.num_links):
        Io = np.array([[P[l, 4], P[l, 5], P[l, 6]], [P[l, 5], P[l, 7], P[l, 8]], [P[l, 6], P[l, 8], P[l, 9]]])
        if m[l] > 0:
            c = P[l, 1:4] / m[l]
            com[l] = c
            Ic[l] = Io - m[l] * (c @ c * np.eye(3) - np.outer(c, c))
        else:
            Ic[l] = Io
    return m, com, Ic


def world_kinematics(topo, q, dq, ddq, R_wb, v_b, w_b, a_b, dw_b, p_b=None):
    """World-frame pose/velocity/acceleration of every link origin.

    R_wb (S,3,3) base orientation, v_b/a_b linear velocity/acceleration of the base origin (world),
    w_b/dw_b angular velocity/acceleration (world).  Returns dict of (L,S,...) arrays."""
    S = q.shape[0]
    L = topo.num_links
    R = np.zeros((L, S, 3, 3))
    p = np.zeros((L, S, 3))
    v = np.zeros((L, S, 3))
    w = np.zeros((L, S, 3))
    a = np.zeros((L, S, 3))
    dw = np.zeros((L, S, 3))
    for l in topo.traversal():
        par = topo.parent[l]
        if par < 0:
            R[l] = R_wb
            p[l] = 0.0 if p_b is None else p_b
            v[l], w[l], a[l], dw[l] = v_b, w_b, a_b, dw_b
            continue
        d = topo.dof_index[l]
        prismatic = d >= 0 and topo.joint_type[l] == 2
        Rj = topo.rest_R[l][None]
        if d >= 0 and not prismatic:
            Rj = Rj @ axis_angle_R(topo.axis[l], q[:, d])
        R[l] = R[par] @ Rj
        r = np.einsum("sij,j->si", R[par], topo.rest_p[l])
        if prismatic:  # the origin slides along the (world) axis: relative velocity / acceleration terms of a moving point
            sw = np.einsum("sij,j->si", R[l], topo.axis[l])
            r = r + sw * q[:, d : d + 1]
        p[l] = p[par] + r
        v[l] = v[par] + np.cross(w[par], r)
        a[l] = a[par] + np.cross(dw[par], r) + np.cross(w[par], np.cross(w[par], r))
        w[l] = w[par]
        dw[l] = dw[par]
        if prismatic:
            v[l] = v[l] + sw * dq[:, d : d + 1]
            a[l] = a[l] + 2 * np.cross(w[par], sw) * dq[:, d : d + 1] + sw * ddq[:, d : d + 1]
        elif d >= 0:
            sw = np.einsum("sij,j->si", R[l], topo.axis[l])
            w[l] = w[par] + sw * dq[:, d : d + 1]
            dw[l] = dw[par] + sw * ddq[:, d : d + 1] + np.cross(w[par], sw) * dq[:, d : d + 1]
    return {"R": R, "p": p, "v": v, "w": w, "a": a, "dw": dw}


def inverse_dynamics_world(topo, q, dq, ddq, floating, base_vel=None, base_acc=None, rpy=None,
                           gravity=(0.0, 0.0, -9.81), x_inertial=None):
    """Generalized torques (S, n[+6]) by world-frame Newton-Euler about the COMs."""
    S = q.shape[0]
    if x_inertial is not None:
        import copy

        topo = copy.copy(topo)
        topo.params = np.asarray(x_inertial).reshape(-1, 10)
    g = np.asarray(gravity, dtype=float)
    if floating:
        R_wb = np.transpose(rpy_R(rpy), (0, 2, 1))
        v_b, w_b = base_vel[:, :3], base_vel[:, 3:]
        a_b, dw_b = base_acc[:, :3], base_acc[:, 3:]
    else:
        R_wb = np.tile(np.eye(3), (S, 1, 1))
        v_b = w_b = a_b = dw_b = np.zeros((S, 3))
    k = world_kinematics(topo, q, dq, ddq, R_wb, v_b, w_b, a_b, dw_b)
    m, com, Ic = link_inertials(topo)
    L, n = topo.num_links, topo.num_dofs
    F = np.zeros((L, S, 3))  # net force on link
    N = np.zeros((L, S, 3))  # net moment about the link COM
    PC = np.zeros((L, S, 3))  # COM position
    for l in range(L):
        rc = np.einsum("sij,j->si", k["R"][l], com[l])
        PC[l] = k["p"][l] + rc
        ac = k["a"][l] + np.cross(k["dw"][l], rc) + np.cross(k["w"][l], np.cross(k["w"][l], rc))
        F[l] = m[l] * (ac - g[None])
        Iw = k["R"][l] @ Ic[l][None] @ np.transpose(k["R"][l], (0, 2, 1))
        Iw_w = np.einsum("sij,sj->si", Iw, k["w"][l])
        N[l] = np.einsum("sij,sj->si", Iw, k["dw"][l]) + np.cross(k["w"][l], Iw_w)
    fb = 6 if floating else 0
    tau = np.zeros((S, n + fb))
    anc = topo.ancestors_dofs()
    joint_link = {topo.dof_index[l]: l for l in range(L) if topo.dof_index[l] >= 0}
    for l in range(L):
        if floating:
            tau[:, 0:3] += F[l]
            tau[:, 3:6] += N[l] + np.cross(PC[l], F[l])
        for d in anc[l]:
            jl = joint_link[d]
            sw = np.einsum("sij,j->si", k["R"][jl], topo.axis[jl])
            if topo.joint_type[jl] == 2:  # prismatic: the force along the axis
                tau[:, fb + d] += np.einsum("si,si->s", sw, F[l])
                continue
            mom = N[l] + np.cross(PC[l] - k["p"][jl], F[l])
            tau[:, fb + d] += np.einsum("si,si->s", sw, mom)
    return tau


def energy(topo, q, dq, R_wb, p_b, v_b, w_b, gravity=(0.0, 0.0, -9.81)):
    """Kinetic + potential energy (S,)."""
    S = q.shape[0]
    z = np.zeros((S, 3))
    k = world_kinematics(topo, q, dq, np.zeros_like(q), R_wb, v_b, w_b, z, z, p_b=p_b)
    m, com, Ic = link_inertials(topo)
    g = np.asarray(gravity, dtype=float)
    E = np.zeros(S)
    for l in range(topo.num_links):
        rc = np.einsum("sij,j->si", k["R"][l], com[l])
        vc = k["v"][l] + np.cross(k["w"][l], rc)
        wl = np.einsum("sji,sj->si", k["R"][l], k["w"][l])  # angular velocity in link axes
        E += 0.5 * m[l] * np.einsum("si,si->s", vc, vc) + 0.5 * np.einsum("si,ij,sj->s", wl, Ic[l], wl)
        E -= m[l] * ((k["p"][l] + rc) @ g)
    return E
