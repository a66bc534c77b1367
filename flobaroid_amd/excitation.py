"""Batched pieces of the trajectory optimiser's inner loop on top of the fused kernels.

The reference evaluates, per candidate trajectory, ``YBase^T YBase`` -> ``eigvalsh`` (excitation/trajectoryOptimizer.py:248-272)
and, for the analytical gradient, 1 + 3 n regressors per sample in a Python/iDynTree loop spread over worker processes
(excitation/analyticalGradient.py:92-185).  Here both are single device passes:

* ``candidate_dopt``      -- D-optimality of many candidate trajectories from ``Engine.gram_grouped``;
* ``dopt_sensitivities``  -- the worker's ``sens_q, sens_dq, sens_ddq`` from ``Engine.fd_scores``.
"""
from __future__ import annotations

import numpy as np

from . import estimation as est


def candidate_dopt(engine, states: dict, num_candidates: int, independent_cols, dopt_regularization: float = 1e-4, w=None,
                   YtY_prior=None) -> np.ndarray:
    """Regularised D-optimality -sum(log(max(eig(YBase^T YBase [+ YtY_prior]) + delta, 1e-300))), delta = doptRegularization *
    lambda_max per candidate, of ``num_candidates`` trajectories stacked along the sample axis (equal length each), one fused
    pass (trajectoryOptimizer.py:259-272 per candidate)."""
    G = engine.gram_grouped(states, int(num_candidates), w=w)
    G = G.cpu().numpy() if hasattr(G, "cpu") else G
    return est.d_optimality_batch(G, independent_cols, dopt_regularization, YtY_prior)


def dopt_sensitivities(engine, states: dict, W_iner, epsilon: float, W_visc=None, reference_state_carryover: bool = False):
    """``_gradient_worker_chunk`` of analyticalGradient.py:92-185 without its loops: returns ``(sens_q, sens_dq, sens_ddq)``,
    each (S, n), with sens[t, d] = (sum(W_t * Y(state_t + eps e_d)) - sum(W_t * Y(state_t))) / eps.

    ``W_iner`` (S * rows, cols): the D-optimality weight rows of the samples (the reference's ``W_iner`` restricted to the
    identified columns).  ``W_visc`` (S * rows,) optional: the analytic viscous-friction term added to ``sens_dq``
    (analyticalGradient.py:141-143: ``W_visc[t * n_out + fb + d]``).  For a floating base pass the states the reference
    uses there (identity base orientation, zero base twist: rpy = 0, base_vel = 0, base_acc = 0).

    ``reference_state_carryover``: the reference's worker does not reset the kinematic state after its velocity sweep
    (analyticalGradient.py:128-165: ``dq_buf`` is restored but ``setRobotState`` is not called again), so the whole
    acceleration sweep runs with dq_{n-1} + eps still active and every ``sens_ddq[t, d]`` carries the extra term
    ``sens_dq_inertial[t, n-1]`` (exactly: the acceleration part of the regressor does not depend on dq).  False (default)
    returns the clean derivative; True reproduces the reference's numbers (pinned by tests/golden/ref_compute_regressors.npz)."""
    sc = engine.fd_scores(states, W_iner, float(epsilon))
    sc = sc.cpu().numpy() if hasattr(sc, "cpu") else sc
    n = engine.topo.num_dofs
    d = (sc[:, 1:] - sc[:, :1]) / float(epsilon)
    sens_q, sens_dq, sens_ddq = d[:, :n], d[:, n:2 * n].copy(), d[:, 2 * n:3 * n].copy()
    if reference_state_carryover:
        sens_ddq += sens_dq[:, n - 1:n]
    if W_visc is not None:
        fb = engine.rows - n
        Wv = np.asarray(W_visc, dtype=float).reshape(sc.shape[0], engine.rows)
        sens_dq += Wv[:, fb:fb + n]
    return sens_q, sens_dq, sens_ddq
