"""The oracle's regressor against FIRST PRINCIPLES: the Euler-Lagrange equations, derived symbolically (SymPy) from the kinetic and potential
energy of every link written in its standard parameters (mass, first moment, inertia about the link origin) -- no spatial algebra, no
recursion, no code or convention shared with oracle/fbr_oracle.c, tests/np_dynamics.py or the kernels.

    T_l = 1/2 m |v|^2 + v . (w x R h) + 1/2 wb^T I wb,      U_l = - g . (m p + R h),      tau_j = d/dt dL/d(dq_j) - dL/dq_j,   L = T - U

with p, R the pose of the link frame in the world, v = dp/dt, w from dR/dt R^T, wb = R^T w.  The Lagrangian is linear in the parameters, so
column (l, k) of the regressor is the Euler-Lagrange operator applied to the energy of a unit parameter k of link l.  Fixed base, small
random trees with revolute, prismatic and fixed joints (test helper; CPU only)."""
import numpy as np
import pytest

from common import random_states, random_topology

sp = pytest.importorskip("sympy")


def _rodrigues(axis, q):
    s = sp.Matrix(axis)
    K = sp.Matrix([[0, -s[2], s[1]], [s[2], 0, -s[0]], [-s[1], s[0], 0]])
    return sp.eye(3) + sp.sin(q) * K + (1 - sp.cos(q)) * (K * K)


def _lagrange_regressor(t, gravity=(0.0, 0.0, -9.81)):
    """callable (q, dq, ddq) -> (n, 10 L) regressor of the fixed-base tree ``t`` from the symbolic Euler-Lagrange equations"""
    L, n = t.num_links, t.num_dofs
    q = sp.symbols("q0:%d" % n, real=True)
    dq = sp.symbols("dq0:%d" % n, real=True)
    ddq = sp.symbols("ddq0:%d" % n, real=True)
    g = sp.Matrix(gravity)
    R, p = [None] * L, [None] * L
    # links are numbered parents first in tests/common.random_topology
    for l in range(L):
        par = t.parent[l]
        if par < 0:
            R[l], p[l] = sp.eye(3), sp.zeros(3, 1)
            continue
        rest_R = sp.Matrix(np.asarray(t.rest_R[l], dtype=float))
        rest_p = sp.Matrix(np.asarray(t.rest_p[l], dtype=float))
        ax = [float(a) for a in t.axis[l]]
        d = t.dof_index[l]
        jt = t.joint_type[l]
        if jt == 1:
            R[l] = R[par] * rest_R * _rodrigues(ax, q[d])
            p[l] = p[par] + R[par] * rest_p
        elif jt == 2:
            R[l] = R[par] * rest_R
            p[l] = p[par] + R[par] * (rest_p + rest_R * sp.Matrix(ax) * q[d])
        else:
            R[l] = R[par] * rest_R
            p[l] = p[par] + R[par] * rest_p

    def ddt(expr):  # total time derivative of an expression in (q, dq)
        return sum(sp.diff(expr, q[k]) * dq[k] + sp.diff(expr, dq[k]) * ddq[k] for k in range(n))

    def ddt_q(expr):  # ... of an expression in q only
        return sum(sp.diff(expr, q[k]) * dq[k] for k in range(n))

    cols = []
    for l in range(L):
        v = p[l].applyfunc(ddt_q)
        Rd = R[l].applyfunc(ddt_q)
        W = Rd * R[l].T  # [w]x
        w = sp.Matrix([W[2, 1], W[0, 2], W[1, 0]])
        wb = R[l].T * w
        unit = []
        # mass
        unit.append(sp.Rational(1, 2) * (v.T * v)[0] + (g.T * p[l])[0])
        # first moment h = e_k
        for k in range(3):
            e = sp.zeros(3, 1)
            e[k] = 1
            Rh = R[l] * e
            unit.append((v.T * w.cross(Rh))[0] + (g.T * Rh)[0])
        # inertia about the link origin, order xx xy xz yy yz zz
        for (a, b) in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)):
            unit.append(sp.Rational(1, 2) * wb[a] * wb[a] if a == b else wb[a] * wb[b])
        for Lk in unit:  # L = T - U with U = -g.(m p + R h): the +g terms above are -U
            cols.append([ddt(sp.diff(Lk, dq[j])) - sp.diff(Lk, q[j]) for j in range(n)])
    Y = sp.Matrix(n, 10 * L, lambda j, c: cols[c][j])
    f = sp.lambdify([q, dq, ddq], Y, modules="numpy", cse=True)
    return lambda qv, dqv, ddqv: np.asarray(f(list(qv), list(dqv), list(ddqv)), dtype=float)


_SLOW = pytest.mark.skipif(not __import__("os").environ.get("FBR_SLOW_TESTS"), reason="minutes of symbolic differentiation: FBR_SLOW_TESTS=1 (run and green "
                           "when the test was written: 5 fixed-base cases in 46 s, 3 floating-base cases in 311 s, the fourth contact case in 69 s)")


@pytest.mark.parametrize("seed,L,p_fixed,branch,p_prism", [(1, 3, 0.0, 0.0, 0.0), pytest.param(2, 4, 0.0, 0.5, 0.0, marks=_SLOW), (3, 4, 0.25, 0.5, 0.4),
                                                          pytest.param(4, 5, 0.2, 1.0, 0.3, marks=_SLOW), (5, 4, 0.0, 0.0, 1.0)])
def test_oracle_joint_rows_are_the_euler_lagrange_equations(seed, L, p_fixed, branch, p_prism):
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(1000 + seed)
    t = random_topology(rng, L, p_fixed=p_fixed, branchiness=branch, p_prismatic=p_prism)
    if t.num_dofs == 0:
        pytest.skip("no joint")
    Yl = _lagrange_regressor(t)
    om = OracleModel(t, floating=False)
    S = 6
    st = random_states(t, S, rng, 0)
    Yo = om.regressor(st).reshape(S, t.num_dofs, 10 * L)
    for s in range(S):
        Ys = Yl(st["q"][s], st["dq"][s], st["ddq"][s])
        scale = max(1.0, np.abs(Ys).max())
        assert np.abs(Yo[s] - Ys).max() <= 1e-10 * scale, (seed, s, np.abs(Yo[s] - Ys).max())
    # structural zeros of the oracle are zeros of the equations too (a joint does not feel the links it does not carry)
    assert np.abs(Yl(st["q"][0], st["dq"][0], st["ddq"][0])[Yo[0] == 0.0]).max(initial=0.0) <= 1e-10


def _lagrange_regressor_floating(t, gravity=(0.0, 0.0, -9.81)):
    """Floating base: generalised coordinates x = (base position in the world, roll-pitch-yaw with world_R_base = RPY(rpy)^T as the
    reference builds it, joints).  Euler-Lagrange in these TRUE coordinates gives the generalised forces (f, Q_rpy, tau); the moment
    about world axes that the MIXED representation reports follows from the virtual work n . (E d_rpy) = Q_rpy . d_rpy with
    w = E(rpy) d(rpy)/dt:  n = E^-T Q_rpy.  Returns a callable (rpy, v, w, a, dw, q, dq, ddq) -> (6 + n, 10 L)."""
    L, n = t.num_links, t.num_dofs
    pb = sp.symbols("pb0:3", real=True)
    ph = sp.symbols("ph0:3", real=True)
    q = sp.symbols("q0:%d" % n, real=True)
    x = list(pb) + list(ph) + list(q)
    dx = list(sp.symbols("dx0:%d" % len(x), real=True))
    ddx = list(sp.symbols("ddx0:%d" % len(x), real=True))
    g = sp.Matrix(gravity)
    r_, p_, y_ = ph
    Rz = sp.Matrix([[sp.cos(y_), -sp.sin(y_), 0], [sp.sin(y_), sp.cos(y_), 0], [0, 0, 1]])
    Ry = sp.Matrix([[sp.cos(p_), 0, sp.sin(p_)], [0, 1, 0], [-sp.sin(p_), 0, sp.cos(p_)]])
    Rx = sp.Matrix([[1, 0, 0], [0, sp.cos(r_), -sp.sin(r_)], [0, sp.sin(r_), sp.cos(r_)]])
    R, p = [None] * L, [None] * L
    for l in range(L):
        par = t.parent[l]
        if par < 0:
            R[l], p[l] = (Rz * Ry * Rx).T, sp.Matrix(pb)
            continue
        rest_R = sp.Matrix(np.asarray(t.rest_R[l], dtype=float))
        rest_p = sp.Matrix(np.asarray(t.rest_p[l], dtype=float))
        ax = [float(a) for a in t.axis[l]]
        d = t.dof_index[l]
        jt = t.joint_type[l]
        if jt == 1:
            R[l] = R[par] * rest_R * _rodrigues(ax, q[d])
            p[l] = p[par] + R[par] * rest_p
        elif jt == 2:
            R[l] = R[par] * rest_R
            p[l] = p[par] + R[par] * (rest_p + rest_R * sp.Matrix(ax) * q[d])
        else:
            R[l] = R[par] * rest_R
            p[l] = p[par] + R[par] * rest_p
    N = len(x)

    def ddt(expr):
        return sum(sp.diff(expr, x[k]) * dx[k] + sp.diff(expr, dx[k]) * ddx[k] for k in range(N))

    def ddt_x(expr):
        return sum(sp.diff(expr, x[k]) * dx[k] for k in range(N))

    cols = []
    for l in range(L):
        v = p[l].applyfunc(ddt_x)
        W = R[l].applyfunc(ddt_x) * R[l].T
        w = sp.Matrix([W[2, 1], W[0, 2], W[1, 0]])
        wb = R[l].T * w
        unit = [sp.Rational(1, 2) * (v.T * v)[0] + (g.T * p[l])[0]]
        for k in range(3):
            e = sp.zeros(3, 1)
            e[k] = 1
            Rh = R[l] * e
            unit.append((v.T * w.cross(Rh))[0] + (g.T * Rh)[0])
        for (a, b) in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)):
            unit.append(sp.Rational(1, 2) * wb[a] * wb[a] if a == b else wb[a] * wb[b])
        for Lk in unit:
            cols.append([ddt(sp.diff(Lk, dx[j])) - sp.diff(Lk, x[j]) for j in range(N)])
    Q = sp.Matrix(N, 10 * L, lambda j, c: cols[c][j])
    # w = E(rpy) d(rpy)/dt of the base
    Wb = R[0].applyfunc(lambda e: sum(sp.diff(e, ph[k]) * dx[3 + k] for k in range(3))) * R[0].T
    wbase = sp.Matrix([Wb[2, 1], Wb[0, 2], Wb[1, 0]])
    E = wbase.jacobian(sp.Matrix(dx[3:6]))
    dE = [E.applyfunc(lambda e, k=k: sp.diff(e, ph[k])) for k in range(3)]
    fQ = sp.lambdify([x, dx, ddx], Q, modules="numpy", cse=True)
    fE = sp.lambdify([ph], E, modules="numpy")
    fdE = [sp.lambdify([ph], m_, modules="numpy") for m_ in dE]

    def evaluate(rpy, v, w, a, dw, qv, dqv, ddqv):
        Em = np.asarray(fE(list(rpy)), dtype=float)
        dph = np.linalg.solve(Em, w)
        Ed = sum(np.asarray(fdE[k](list(rpy)), dtype=float) * dph[k] for k in range(3))
        ddph = np.linalg.solve(Em, dw - Ed @ dph)
        xv = [0.0, 0.0, 0.0] + list(rpy) + list(qv)
        dxv = list(v) + list(dph) + list(dqv)
        ddxv = list(a) + list(ddph) + list(ddqv)
        Qm = np.asarray(fQ(xv, dxv, ddxv), dtype=float)
        Y = Qm.copy()
        Y[3:6] = np.linalg.solve(Em.T, Qm[3:6])  # moment about world axes (at the base origin): n = E^-T Q_rpy
        return Y

    return evaluate


@pytest.mark.parametrize("seed,L,p_fixed,branch,p_prism", [(11, 2, 0.0, 0.0, 0.0), pytest.param(12, 3, 0.0, 0.5, 0.0, marks=_SLOW),
                                                          pytest.param(13, 3, 0.3, 1.0, 0.5, marks=_SLOW)])
def test_oracle_floating_base_rows_are_the_mixed_generalised_forces(seed, L, p_fixed, branch, p_prism):
    """The six base rows: force in world axes and moment in world axes about the base origin, for a base twist given as (velocity of the
    base origin in world axes, angular velocity in world axes) and its plain time derivative -- the MIXED representation the reference
    hands to iDynTree (identification/model.py:405-446) -- derived from the Lagrangian in true coordinates."""
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(2000 + seed)
    t = random_topology(rng, L, p_fixed=p_fixed, branchiness=branch, p_prismatic=p_prism)
    if t.num_dofs == 0:
        pytest.skip("no joint")
    Yl = _lagrange_regressor_floating(t)
    om = OracleModel(t, floating=True)
    S = 5
    st = random_states(t, S, rng, 1)
    st["rpy"] = rng.uniform(-1.2, 1.2, (S, 3))  # (away from the gimbal lock of the chart; the reference's own tests use 0.1 * U(0, 1))
    n = t.num_dofs
    Yo = om.regressor(st).reshape(S, 6 + n, 10 * L)
    for s in range(S):
        bv, ba = st["base_vel"][s], st["base_acc"][s]
        Ys = Yl(st["rpy"][s], bv[:3], bv[3:], ba[:3], ba[3:], st["q"][s], st["dq"][s], st["ddq"][s])
        scale = max(1.0, np.abs(Ys).max())
        err = np.abs(Yo[s] - Ys)
        assert err.max() <= 1e-9 * scale, (seed, s, "base rows" if err[:6].max() == err.max() else "joint rows", err.max())


@pytest.mark.parametrize("seed,L,floating,p_prism", [(21, 4, 0, 0.0), (22, 4, 0, 0.5), (23, 3, 1, 0.0), pytest.param(24, 4, 1, 0.4, marks=_SLOW)])
def test_oracle_contact_torques_are_the_virtual_work_of_the_wrench(seed, L, floating, p_prism):
    """A4 (J^T w, model.py:535-549) from the principle of virtual work: a wrench (f, n) in world axes at the origin of a frame fixed to a
    link does the work f . dp + n . dtheta on a virtual displacement; with p(x), w = J_w(x) dx/dt derived symbolically from the
    kinematics, the generalised forces are (dp/dx)^T f + J_w^T n -- for a floating base in the true coordinates (position,
    roll-pitch-yaw), the moment rows mapped to world axes by E^-T as in the regressor test above."""
    from scipy.spatial.transform import Rotation

    from oracle.oracle import OracleModel

    rng = np.random.default_rng(3000 + seed)
    t = random_topology(rng, L, p_fixed=0.2, branchiness=0.5, p_prismatic=p_prism)
    if t.num_dofs == 0:
        pytest.skip("no joint")
    n = t.num_dofs
    fl = L - 1
    fR, fp = Rotation.random(random_state=seed).as_matrix(), 0.3 * rng.standard_normal(3)
    t.frames = {"tool": {"link": fl, "R": fR, "p": fp}}
    pb = sp.symbols("pb0:3", real=True)
    ph = sp.symbols("ph0:3", real=True)
    q = sp.symbols("q0:%d" % n, real=True)
    x = (list(pb) + list(ph) if floating else []) + list(q)
    dx = list(sp.symbols("dx0:%d" % len(x), real=True))
    r_, p_, y_ = ph
    Rz = sp.Matrix([[sp.cos(y_), -sp.sin(y_), 0], [sp.sin(y_), sp.cos(y_), 0], [0, 0, 1]])
    Ry = sp.Matrix([[sp.cos(p_), 0, sp.sin(p_)], [0, 1, 0], [-sp.sin(p_), 0, sp.cos(p_)]])
    Rx = sp.Matrix([[1, 0, 0], [0, sp.cos(r_), -sp.sin(r_)], [0, sp.sin(r_), sp.cos(r_)]])
    R, p = [None] * L, [None] * L
    for l in range(L):
        par = t.parent[l]
        if par < 0:
            R[l], p[l] = ((Rz * Ry * Rx).T, sp.Matrix(pb)) if floating else (sp.eye(3), sp.zeros(3, 1))
            continue
        rest_R = sp.Matrix(np.asarray(t.rest_R[l], dtype=float))
        rest_p = sp.Matrix(np.asarray(t.rest_p[l], dtype=float))
        ax = [float(a) for a in t.axis[l]]
        d = t.dof_index[l]
        if t.joint_type[l] == 1:
            R[l], p[l] = R[par] * rest_R * _rodrigues(ax, q[d]), p[par] + R[par] * rest_p
        elif t.joint_type[l] == 2:
            R[l], p[l] = R[par] * rest_R, p[par] + R[par] * (rest_p + rest_R * sp.Matrix(ax) * q[d])
        else:
            R[l], p[l] = R[par] * rest_R, p[par] + R[par] * rest_p
    pf = p[fl] + R[fl] * sp.Matrix(fp)
    N = len(x)
    W = R[fl].applyfunc(lambda e: sum(sp.diff(e, x[k]) * dx[k] for k in range(N))) * R[fl].T
    w = sp.Matrix([W[2, 1], W[0, 2], W[1, 0]])
    Jp, Jw = pf.jacobian(sp.Matrix(x)), w.jacobian(sp.Matrix(dx))
    fJ = sp.lambdify([x], [Jp, Jw], modules="numpy")
    if floating:
        Wb = R[0].applyfunc(lambda e: sum(sp.diff(e, ph[k]) * dx[3 + k] for k in range(3))) * R[0].T
        E = sp.Matrix([Wb[2, 1], Wb[0, 2], Wb[1, 0]]).jacobian(sp.Matrix(dx[3:6]))
        fE = sp.lambdify([ph], E, modules="numpy")
    om = OracleModel(t, floating=bool(floating))
    S = 6
    st = random_states(t, S, rng, floating)
    if floating:
        st["rpy"] = rng.uniform(-1.2, 1.2, (S, 3))
    wr = rng.standard_normal((S, 6))
    got = om.contact_torques(st, "tool", wr)
    for s in range(S):
        xv = ([0.0, 0.0, 0.0] + list(st["rpy"][s]) if floating else []) + list(st["q"][s])
        Jpv, Jwv = (np.asarray(m_, dtype=float) for m_ in fJ(xv))
        Q = Jpv.T @ wr[s, :3] + Jwv.T @ wr[s, 3:]
        if floating:
            Q[3:6] = np.linalg.solve(np.asarray(fE(list(st["rpy"][s])), dtype=float).T, Q[3:6])
        assert np.abs(got[s] - Q).max() <= 1e-11 * max(1.0, np.abs(Q).max()), (seed, s, np.abs(got[s] - Q).max())
