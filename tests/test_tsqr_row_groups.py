"""The algebra behind the tree-structured TSQR (csrc/fbr_api.hip: tsqr_group_plan), checked on the CPU with the
oracle's regressor: the rows grouped along the kinematic tree -- base-wrench rows, one group per unbranched chain of joints --
touch only the columns of their own sub-tree; factorising every group over those columns and folding the embedded group factors
gives the R of the whole matrix; and the dense tile-update count drops as DESIGN states.  `group_plan` below restates the
grouping rule of the library in Python (the library's own plan is exercised by tests/test_gpu_parity.py on the GPU)."""
import numpy as np
import pytest

from common import load_topo, random_states
from oracle.oracle import OracleModel


def group_plan(topo, floating):
    """rows (regressor row indices) and links (sub-tree) of every group"""
    L, n = topo.num_links, topo.num_dofs
    path = [None] * L

    def getpath(l):
        if path[l] is None:
            p = [] if topo.parent[l] < 0 else list(getpath(topo.parent[l]))
            path[l] = p + ([topo.dof_index[l]] if topo.dof_index[l] >= 0 else [])
        return path[l]

    for l in range(L):
        getpath(l)
    pj, depth = [-1] * n, [0] * n
    for l in range(L):
        d = topo.dof_index[l]
        if d >= 0:
            depth[d] = len(path[l])
            pj[d] = path[l][-2] if len(path[l]) >= 2 else -1
    nchild = [0] * (n + 1)
    for d in range(n):
        nchild[pj[d] + 1] += 1
    fb = 6 if floating else 0
    groups, jgroup, base = [], [-1] * n, -1
    if fb:
        base = 0
        groups.append(list(range(fb)))
    for d in sorted(range(n), key=lambda d: depth[d]):
        pg = base if pj[d] < 0 else jgroup[pj[d]]
        if nchild[pj[d] + 1] == 1 and pg < 0:
            groups.append([])
            pg = base = len(groups) - 1
        if nchild[pj[d] + 1] == 1:
            jgroup[d] = pg
        else:
            groups.append([])
            jgroup[d] = len(groups) - 1
        groups[jgroup[d]].append(fb + d)
    out = []
    for rows in groups:
        joints = [r - fb for r in rows if r >= fb]
        links = list(range(L)) if any(r < fb for r in rows) else [l for l in range(L) if any(d in path[l] for d in joints)]
        out.append((rows, links))
    return out


@pytest.mark.parametrize("name,floating,ngroups", [("walkman_apriori", 1, 6), ("walkman_apriori", 0, 5), ("kuka_lwr4", 1, 1),
                                                   ("walkman_left_arm", 1, 1), ("threeLinks", 1, 1)])
def test_group_factors_assemble_the_full_factor(name, floating, ngroups):
    t = load_topo(name)
    om = OracleModel(t, floating=floating)
    rng = np.random.default_rng(3)
    S = 60
    st = random_states(t, S, rng, floating)
    Y = om.regressor(st)
    tau = rng.standard_normal((Y.shape[0], 1))
    A = np.hstack([Y, tau])
    P = Y.shape[1]
    plan = group_plan(t, floating)
    assert len(plan) == ngroups
    A3 = A.reshape(S, om.rows, P + 1)
    stacked, dense_updates = [], 0.0
    seen = np.zeros(om.rows, dtype=bool)
    for rows, links in plan:
        cols = np.array([10 * l + p for l in links for p in range(10)] + [P])
        seen[rows] = True
        sub = A3[:, rows, :].reshape(-1, P + 1)
        others = np.setdiff1d(np.arange(P + 1), cols)
        assert np.all(sub[:, others] == 0.0)                  # the group's rows touch nothing outside its sub-tree's columns
        Rg = np.linalg.qr(sub[:, cols], mode="r")             # the group's own factorisation, over its columns only
        E = np.zeros((Rg.shape[0], P + 1))
        E[:, cols] = Rg                                       # embedded into the caller's column order
        stacked.append(E)
        dense_updates += len(rows) * (len(cols) / (P + 1.0)) ** 2
    assert seen.all()
    R = np.linalg.qr(np.vstack(stacked), mode="r")            # the embedded factors folded like data rows
    Rref = np.linalg.qr(A, mode="r")
    G = A.T @ A
    assert np.linalg.norm(R.T @ R - G) <= 1e-12 * np.linalg.norm(G)
    assert np.linalg.norm(R.T @ R - Rref.T @ Rref) <= 1e-12 * np.linalg.norm(G)
    if name == "walkman_apriori" and floating:
        # 6 x 481, 3 x 311, 2 x (6 x 71), 2 x (7 x 91): 0.23 of the dense row x column^2 work before the in-group first-column skipping
        assert dense_updates / om.rows < 0.24
        assert sorted(len(r) for r, _ in plan) == [3, 6, 6, 6, 7, 7]
