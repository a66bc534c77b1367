"""Pins of the CPU oracle: independent formulations, analytic cases, documented known answers (no GPU)."""
import json
import os

import numpy as np
import pytest

import np_dynamics as nd
from common import CONFIGS, GOLDEN, cfg_id, load_topo, random_states
from oracle.oracle import OracleModel, gram, lin_deps_qr


@pytest.mark.parametrize("name,floating", [("threeLinks", 1), ("kuka_lwr4", 0), ("kuka_lwr4", 1), ("walkman_left_arm", 1),
                                           ("walkman_apriori", 1), ("walkman_apriori", 0)])
def test_regressor_equals_inverse_dynamics(name, floating):
    """The reference's own pin (tests/test_regressors.py:118-126, ||Y x - ID||_F <= 0.01 over 100 states),
    held to 1e-12 relative, plus agreement with the world-frame Newton-Euler formulation."""
    t = load_topo(name)
    rng = np.random.default_rng(10)
    st = random_states(t, 100, rng, floating)
    om = OracleModel(t, floating=floating)
    Y = om.regressor(st)
    x = t.x_std()
    tau = om.inverse_dynamics(st, x)
    assert np.linalg.norm(Y @ x - tau.reshape(-1)) <= 0.01
    assert np.abs(Y @ x - tau.reshape(-1)).max() <= 1e-12 * np.abs(tau).max()
    tau_np = nd.inverse_dynamics_world(t, st["q"], st["dq"], st["ddq"], floating, st.get("base_vel"), st.get("base_acc"),
                                       st.get("rpy"))
    assert np.abs(tau_np - tau).max() <= 1e-12 * np.abs(tau).max()
    # linearity in the parameters
    x2 = x + rng.standard_normal(x.shape)
    assert np.abs(Y @ x2 - om.inverse_dynamics(st, x2).reshape(-1)).max() <= 1e-11 * np.abs(Y @ x2).max()


@pytest.mark.parametrize("name,floating", [("threeLinks", 1), ("kuka_lwr4", 0), ("walkman_left_arm", 1)])
def test_power_balance(name, floating):
    """d/dt (kinetic + potential energy) = base wrench . base twist + tau . dq, by central differences."""
    _power_balance(load_topo(name), floating)


def _power_balance(t, floating):
    rng = np.random.default_rng(11)
    S, n = 8, t.num_dofs
    st = random_states(t, S, rng, floating)
    om = OracleModel(t, floating=floating)
    tau = om.inverse_dynamics(st, t.x_std())
    h = 1e-5

    def E(dt):
        q = st["q"] + dt * st["dq"] + 0.5 * dt * dt * st["ddq"]
        dq = st["dq"] + dt * st["ddq"]
        if floating:
            R0 = np.transpose(nd.rpy_R(st["rpy"]), (0, 2, 1))
            w = st["base_vel"][:, 3:] + dt * st["base_acc"][:, 3:]
            wm = st["base_vel"][:, 3:] + 0.5 * dt * st["base_acc"][:, 3:]
            # first-order rotation update is enough for a central difference
            th = wm * dt
            K = np.zeros((S, 3, 3))
            K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -th[:, 2], th[:, 1], th[:, 2], -th[:, 0], -th[:, 1], th[:, 0]
            Rw = (np.eye(3)[None] + K + 0.5 * K @ K) @ R0
            v = st["base_vel"][:, :3] + dt * st["base_acc"][:, :3]
            p = dt * st["base_vel"][:, :3] + 0.5 * dt * dt * st["base_acc"][:, :3]
        else:
            Rw = np.tile(np.eye(3), (S, 1, 1))
            w = v = p = np.zeros((S, 3))
        return nd.energy(t, q, dq, Rw, p, v, w)

    dE = (E(h) - E(-h)) / (2 * h)
    power = np.einsum("si,si->s", tau[:, (6 if floating else 0):], st["dq"])
    if floating:
        power += np.einsum("si,si->s", tau[:, :6], st["base_vel"])
    assert np.abs(dE - power).max() <= 1e-6 * max(np.abs(power).max(), 1.0)


def test_analytic_cases():
    """Single revolute link: tau = m g c sin(theta)-type gravity torque; static base wrench = total weight."""
    t = load_topo("threeLinks")
    om = OracleModel(t, floating=1)
    S = 5
    rng = np.random.default_rng(0)
    st = dict(q=rng.uniform(-1, 1, (S, 2)), dq=np.zeros((S, 2)), ddq=np.zeros((S, 2)), base_vel=np.zeros((S, 6)),
              base_acc=np.zeros((S, 6)), rpy=np.zeros((S, 3)))
    tau = om.inverse_dynamics(st, t.x_std())
    assert np.allclose(tau[:, 0:2], 0, atol=1e-13) and np.allclose(tau[:, 2], 3.0 * 9.81)
    # joint 1 (axis z = gravity direction, links extend along x): no gravity torque
    assert np.allclose(tau[:, 6], 0, atol=1e-12)
    # fixed-base KUKA at rest: only gravity, joint 0 (vertical axis) carries none
    k = load_topo("kuka_lwr4")
    ok = OracleModel(k)
    stk = dict(q=rng.uniform(-1, 1, (S, 7)), dq=np.zeros((S, 7)), ddq=np.zeros((S, 7)))
    tk = ok.inverse_dynamics(stk, k.x_std())
    assert np.allclose(tk[:, 0], 0, atol=1e-12)


def test_structural_base_ranks_and_non_identifiable_set():
    """P4/P8: base ranks 24 / 43 (+21 friction = 64) / 59 / 213 with the reference's random-state distributions
    (model.py:696-725), raw Gram, scipy pivoted QR, minTol of the shipped configs; KUKA non_id = {0..18,20,22}."""
    s = json.load(open(os.path.join(GOLDEN, "structure.json")))
    g = json.load(open(os.path.join(GOLDEN, "kuka_tutorial_apriori.json")))
    rng = np.random.default_rng(1)
    cases = [("threeLinks", 1, 0, 2000, 1e-4, s["threeLinks"]["base_rank_floating"]),
             ("kuka_lwr4", 0, 0, 5000, 1e-4, s["kuka_lwr4"]["base_rank_fixed"]),
             ("kuka_lwr4", 0, 1, 5000, 1e-4, s["kuka_lwr4"]["base_rank_fixed_friction"]),
             ("walkman_left_arm", 1, 0, 3000, 1e-4, s["walkman_left_arm"]["base_rank_floating"]),
             ("walkman_apriori", 1, 0, 3000, 0.005, s["walkman_apriori"]["base_rank_floating"])]
    for name, fl, fric, ns, tol, rank in cases:
        t = load_topo(name)
        om = OracleModel(t, floating=fl, fric=fric, fric_sym=True)
        st = random_states(t, ns, rng, fl, use_limits=True)
        Y = om.regressor(st, np.tanh(st["dq"] / 0.02))
        d = lin_deps_qr(Y.T @ Y, tol)
        assert d["r"] == rank, (name, d["r"], rank)
        if name == "kuka_lwr4":
            non_id = [c for c in range(80) if not np.any(d["K"][:, c])]
            assert non_id == [i for i in g["non_id"] if i < 80]
            assert np.all(np.abs(Y[:, non_id]).max(axis=0) < 1e-12)
        if name == "walkman_apriori":
            assert not np.any(np.abs(Y).max(axis=0) == 0)  # no structurally zero column on the floating humanoid


def test_kuka_trajectory_fixture_observable_base_count():
    """F5: model/kuka_lwr4.urdf.trajectory_opt_1.npz recorded n_observable_base_params = 64 for its 2409 samples
    (trajectory.py:225-264: YBase of the friction layout, SVD-based observability)."""
    z = np.load(os.path.join(GOLDEN, "kuka_trajectory_opt_1.npz"))
    t = load_topo("kuka_lwr4")
    st = dict(q=z["positions"], dq=z["velocities"], ddq=z["accelerations"])
    assert st["q"].shape == (2409, 7)
    om = OracleModel(t, fric=1, fric_sym=True)
    Y = om.regressor(st, np.tanh(st["dq"] / 0.02))
    rng = np.random.default_rng(2)
    rs = random_states(t, 5000, rng, 0, use_limits=True)
    Yr = om.regressor(rs, np.tanh(rs["dq"] / 0.02))
    d = lin_deps_qr(Yr.T @ Yr, 1e-4)
    assert d["r"] == int(z["n_observable_base_params"]) == 64
    YB = Y[:, d["independent_cols"]]
    sv = np.linalg.svd(YB, compute_uv=False)
    assert int(np.sum(sv > 1e-8 * sv[0])) == 64


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
def test_friction_and_layout_columns(cfg):
    name, fl, fr, sym, grav, strb = cfg
    if not fr:
        pytest.skip("no friction block")
    t = load_topo(name)
    om = OracleModel(t, floating=fl, fric=fr, fric_sym=sym, grav_only=grav, stribeck=strb)
    rng = np.random.default_rng(3)
    S, n = 6, t.num_dofs
    st = random_states(t, S, rng, fl)
    sign = np.tanh(st["dq"] / 0.02)
    Y = om.regressor(st, sign).reshape(S, om.rows, om.P)
    fb = 6 if fl else 0
    c0 = (4 if grav else 10) * t.num_links
    assert np.all(Y[:, :fb, c0:] == 0)
    for s in range(S):
        assert np.array_equal(Y[s, fb:, c0:c0 + n], np.diag(sign[s]))
        if grav:
            continue
        c = c0 + n
        if sym:
            assert np.array_equal(Y[s, fb:, c:c + n], np.diag(st["dq"][s]))
            c += n
        else:
            assert np.array_equal(Y[s, fb:, c:c + n], np.diag(np.maximum(st["dq"][s], 0)))
            assert np.array_equal(Y[s, fb:, c + n:c + 2 * n], np.diag(np.minimum(st["dq"][s], 0)))
            c += 2 * n
        assert np.array_equal(Y[s, fb:, c:c + n], np.eye(n))
        if strb > 0:
            assert np.allclose(Y[s, fb:, c + n:c + 2 * n], np.diag(np.exp(-np.abs(st["dq"][s]) / strb) * np.sign(st["dq"][s])))
    G = gram(Y.reshape(-1, om.P)[:50])
    assert np.allclose(G, Y.reshape(-1, om.P)[:50].T @ Y.reshape(-1, om.P)[:50])


# ------------------------------------------------------------------------------------------------------------------------------------
# iDynTree's own outputs (tools/pin_idyntree.py writes tests/golden/idyntree_<robot>.npz on a machine that has the reference's
# environment).  Absent in this tree: iDynTree 15.0.0 cannot be installed in the build container -- the tests then skip and
# the oracle stays "parity unpinned" at that boundary.
# ------------------------------------------------------------------------------------------------------------------------------------
IDYNTREE_ROBOTS = ["threeLinks", "kuka_lwr4", "walkman_left_arm", "walkman_apriori"]


def idyntree_fixture(name):
    import os

    path = os.path.join(GOLDEN, f"idyntree_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not committed: run tools/pin_idyntree.py where `import idyntree` works")
    return dict(np.load(path, allow_pickle=False))


@pytest.mark.parametrize("name", IDYNTREE_ROBOTS)
def test_oracle_against_idyntree_outputs(name):
    """Regressor, inverse dynamics, a-priori parameter vector, link / DOF serialisation and J^T w of the oracle against what iDynTree
    returned for the same seeded states through the reference's own call sequence (model.py:425-446, 268-296, 535-549)."""
    import sys, os

    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import pin_idyntree

    assert pin_idyntree.compare_with_oracle(name, idyntree_fixture(name), verbose=False) <= 1e-9


def test_pin_tool_comparison_plumbing(tmp_path):
    """The comparison the pin rests on, exercised without iDynTree: a fixture in the tool's format whose "iDynTree outputs" are the oracle's
    own (regressor with the 6 base rows a fixed-base call also returns, torques, the Jacobian assembled from unit wrenches) compares
    to ~0, a perturbed entry and a permuted serialisation are caught, and the tool itself says what is missing (exit status 3)."""
    import os, subprocess, sys

    tools = os.path.join(os.path.dirname(GOLDEN), "..", "tools")
    sys.path.insert(0, tools)
    import pin_idyntree as pin
    from oracle.oracle import OracleModel

    name = "kuka_lwr4"
    t = load_topo(name)
    fx = {"x_std": t.x_std()[: 10 * t.num_links], "link_names": np.array(t.link_names), "dof_names": np.array(t.dof_names)}
    for fl in (0, 1):
        st = pin.make_states(t, bool(fl))
        full = dict(st) if fl else dict(st, base_vel=np.zeros((pin.SAMPLES, 6)), base_acc=np.zeros((pin.SAMPLES, 6)), rpy=np.zeros((pin.SAMPLES, 3)))
        om = OracleModel(t, floating=True)  # iDynTree always returns the 6 base rows
        fx.update({f"fb{fl}_{k}": v for k, v in st.items()})
        fx[f"fb{fl}_Y"] = om.regressor(full).reshape(pin.SAMPLES, om.rows, om.P)
        fx[f"fb{fl}_tau"] = om.inverse_dynamics(full, t.x_std())
        J = np.zeros((pin.SAMPLES, 6, om.rows))
        for i in range(6):
            J[:, i, :] = om.contact_torques(full, pin.ROBOTS[name][1], np.tile(np.eye(6)[i], (pin.SAMPLES, 1)))
        fx[f"fb{fl}_J"] = J
        fx[f"fb{fl}_have_frame"] = np.array(True)
    assert pin.compare_with_oracle(name, fx, verbose=False) <= 1e-13
    bad = dict(fx, fb1_Y=fx["fb1_Y"].copy())
    bad["fb1_Y"][3, 7, 42] += 1e-6 * np.abs(fx["fb1_Y"]).max()
    assert pin.compare_with_oracle(name, bad, verbose=False) > 1e-7
    with pytest.raises(AssertionError, match="link order"):
        pin.compare_with_oracle(name, dict(fx, link_names=fx["link_names"][::-1]), verbose=False)
    if pin._import_idyntree() is None:
        r = subprocess.run([sys.executable, os.path.join(tools, "pin_idyntree.py")], capture_output=True, text=True)
        assert r.returncode == 3 and "idyntree" in r.stderr


# ------------------------------------------------------------------------------------------------------------------------------------
# Prismatic joints (round 5): iDynTree's loader takes any URDF (identification/model.py:60-67); no bundled robot has one, so the pins are
# the three formulations against each other on random trees with mixed joint types, and the physics itself (power balance).
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("floating", [0, 1])
def test_prismatic_joints_three_formulations_and_power_balance(seed, floating):
    from common import random_topology

    rng = np.random.default_rng(500 + seed)
    t = random_topology(rng, 6 + 3 * seed, p_fixed=0.25, branchiness=0.4, p_prismatic=0.5)
    if not any(j == 2 for j in t.joint_type):
        pytest.skip("no prismatic joint drawn")
    st = random_states(t, 25, rng, floating)
    om = OracleModel(t, floating=bool(floating))
    Y = om.regressor(st)
    tau = om.inverse_dynamics(st, t.x_std())
    assert np.abs(Y @ t.x_std()[: om.P] - tau.reshape(-1)).max() <= 1e-12 * np.abs(tau).max()   # the reference's own pin, test_regressors.py:118-126
    tw = nd.inverse_dynamics_world(t, st["q"], st["dq"], st["ddq"], floating, st.get("base_vel"), st.get("base_acc"), st.get("rpy"))
    assert np.abs(tw - tau).max() <= 1e-11 * np.abs(tau).max()                                    # world-frame Newton-Euler about the COMs
    _power_balance(t, floating)                                                                   # energy by finite differences
    # a prismatic joint's row of a DISTAL link's mass column is the (proper) acceleration of that link's origin along the axis
    l = next(i for i in range(t.num_links) if t.joint_type[i] == 2)
    d = t.dof_index[l]
    k = nd.world_kinematics(t, st["q"], st["dq"], st["ddq"], *_base(t, st, floating))
    sw = np.einsum("sij,j->si", k["R"][l], t.axis[l])
    acc = k["a"][l] - np.array([0, 0, -9.81])
    row = Y.reshape(25, om.rows, om.P)[:, (6 if floating else 0) + d, 10 * l]
    assert np.abs(row - np.einsum("si,si->s", sw, acc)).max() <= 1e-11 * max(np.abs(acc).max(), 1.0)


def _base(t, st, floating):
    S = st["q"].shape[0]
    if floating:
        return np.transpose(nd.rpy_R(st["rpy"]), (0, 2, 1)), st["base_vel"][:, :3], st["base_vel"][:, 3:], st["base_acc"][:, :3], st["base_acc"][:, 3:]
    z = np.zeros((S, 3))
    return np.tile(np.eye(3), (S, 1, 1)), z, z, z, z


def test_openmp_gram_pass_of_the_cpu_baseline():
    """orc_stack_gram_omp / orc_stack_block_omp (bench.py's all-core CPU baseline): the per-thread rank-1 accumulation equals the stacked
    [Y | tau]^T [Y | tau] of the serial functions, whatever the thread count."""
    t = load_topo("walkman_apriori")
    om = OracleModel(t, floating=True)
    st = random_states(t, 130, np.random.default_rng(8), True)
    A = np.hstack([om.regressor(st), om.inverse_dynamics(st, t.x_std()).reshape(-1, 1)])
    B, thr = om.stack_block(st, t.x_std())
    assert thr >= 1 and np.array_equal(A, B)
    ref = np.triu(A.T @ A)
    for threads in (1, 3):
        G, used = om.stack_gram(st, t.x_std(), threads=threads)
        assert used == threads and np.abs(G - ref).max() <= 1e-12 * np.abs(ref).max() and not np.tril(G, -1).any()
