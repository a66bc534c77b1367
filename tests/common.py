"""Shared helpers for the test-suite."""
from __future__ import annotations

import os

import numpy as np

from flobaroid_amd.topology import Topology

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROBOTS = os.path.join(ROOT, "flobaroid_amd", "robots")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_topo(name: str) -> Topology:
    return Topology.load(os.path.join(ROBOTS, name + ".topology.json"))


def random_states(topo, S, rng, floating, use_limits=False):
    """Random states with the reference's distributions.

    use_limits=False: tests/test_regressors.py:50-66 (U(-pi,pi), base pi*U(0,1), rpy 0.1*U(0,1)).
    use_limits=True : Model.getRandomRegressor, model.py:696-725 (joint limits from the URDF)."""
    n = topo.num_dofs
    if use_limits:
        lo = np.array([topo.limits[j]["lower"] for j in topo.dof_names])
        hi = np.array([topo.limits[j]["upper"] for j in topo.dof_names])
        vm = np.array([topo.limits[j]["velocity"] for j in topo.dof_names])
        st = dict(q=lo + (hi - lo) * rng.random((S, n)), dq=(rng.random((S, n)) - 0.5) * 2 * vm,
                  ddq=(rng.random((S, n)) - 0.5) * 2 * np.pi)
    else:
        st = dict(q=(rng.random((S, n)) * 2 - 1) * np.pi, dq=(rng.random((S, n)) * 2 - 1) * np.pi,
                  ddq=(rng.random((S, n)) * 2 - 1) * np.pi)
    if floating:
        st.update(base_vel=np.pi * rng.random((S, 6)), base_acc=np.pi * rng.random((S, 6)), rpy=0.1 * rng.random((S, 3)))
    return st


# (robot, floating, friction, friction_symmetric, gravity_only, stribeck)
CONFIGS = [
    ("threeLinks", 1, 0, 1, 0, 0.0),
    ("threeLinks", 0, 1, 0, 0, 0.1),
    ("kuka_lwr4", 0, 0, 1, 0, 0.0),
    ("kuka_lwr4", 0, 1, 1, 0, 0.0),
    ("kuka_lwr4", 1, 1, 0, 0, 0.05),
    ("kuka_lwr4", 0, 1, 1, 1, 0.0),
    ("walkman_left_arm", 1, 1, 1, 0, 0.0),
    ("walkman_apriori", 1, 0, 1, 0, 0.0),
    ("walkman_apriori", 0, 1, 1, 0, 0.0),
]


def cfg_id(c):
    return f"{c[0]}-fb{c[1]}-fr{c[2]}{'s' if c[3] else 'a'}-g{c[4]}-st{c[5]}"


def random_topology(rng, num_links, p_fixed=0.2, branchiness=0.5, p_prismatic=0.0):
    """Random kinematic tree (revolute, fixed and -- p_prismatic > 0 -- prismatic joints, random rest transforms and axes) for structure tests of the
    tile program: nothing about it resembles a bundled robot."""
    from scipy.spatial.transform import Rotation

    L = num_links
    parent, jtype, dof = [-1], [0], [-1]
    names = ["l0"]
    n = 0
    for l in range(1, L):
        par = l - 1 if rng.random() > branchiness else int(rng.integers(0, l))
        fixed = rng.random() < p_fixed
        parent.append(par)
        jtype.append(0 if fixed else (2 if (p_prismatic > 0 and rng.random() < p_prismatic) else 1))  # (no draw when 0: the seeded trees of the older tests stay what they were)
        dof.append(-1 if fixed else n)
        n += 0 if fixed else 1
        names.append(f"l{l}")
    rest_R = Rotation.random(L, random_state=int(rng.integers(1 << 30))).as_matrix()
    rest_R[0] = np.eye(3)
    rest_p = rng.standard_normal((L, 3)) * 0.3
    rest_p[0] = 0
    axis = rng.standard_normal((L, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    axis[[i for i in range(L) if dof[i] < 0]] = 0
    params = np.zeros((L, 10))
    params[:, 0] = 1 + rng.random(L)
    params[:, 1:4] = 0.1 * rng.standard_normal((L, 3))
    params[:, [4, 7, 9]] = 0.05 + 0.05 * rng.random((L, 3))
    dof_names = [f"j{d}" for d in range(n)]
    return Topology(name="random", link_names=names, parent=parent, joint_names=[""] + [f"jt{l}" for l in range(1, L)],
                    joint_type=jtype, dof_index=dof, rest_R=rest_R, rest_p=rest_p, axis=axis, params=params, dof_names=dof_names)
