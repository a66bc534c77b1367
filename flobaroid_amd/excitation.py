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


# ------------------------------------------------------------------------------------------------------------------------------------
# Candidate trajectories generated on the device (round 5): the optimiser's search variables are Fourier coefficients; their samples never
# have to exist on the host.
# ------------------------------------------------------------------------------------------------------------------------------------
def fourier_coefficients(a, b, q, nf, wf: float = 1.0, joint_limits=None, use_deg: bool = False):
    """Pack the parameters of ONE candidate as the reference's ``PulsedTrajectory.initWithParams(a, b, q, nf, wf, joint_limits)`` takes
    them (trajectoryGenerator.py:322-383: per-joint coefficient arrays of length nf[i]) into the padded arrays of
    ``Engine.fourier_states``: dict(wf, a (n, nh), b (n, nh), q_offset (n,), q_range (n,) or None).

    classic (``joint_limits`` None, OscillationGenerator 411-460): q_offset = nf * q0;  bounded (BoundedOscillationGenerator 462-510):
    q_offset = q_center = clip(midpoint + q0, lower, upper), q_range = 0.95 * min(q_center - lower, upper - q_center).  ``use_deg``: q
    is given in degrees (the generators convert it)."""
    n = len(nf)
    nh = max(int(k) for k in nf)
    A, B = np.zeros((n, nh)), np.zeros((n, nh))
    for j in range(n):
        A[j, : int(nf[j])] = np.asarray(a[j], dtype=float)[: int(nf[j])]
        B[j, : int(nf[j])] = np.asarray(b[j], dtype=float)[: int(nf[j])]
    q0 = np.deg2rad(np.asarray(q, dtype=float)) if use_deg else np.asarray(q, dtype=float)
    if joint_limits is None:
        return {"wf": float(wf), "a": A, "b": B, "q_offset": np.asarray(nf, dtype=float) * q0, "q_range": None}
    lo = np.array([l[0] for l in joint_limits], dtype=float)
    hi = np.array([l[1] for l in joint_limits], dtype=float)
    qc = np.clip(0.5 * (lo + hi) + q0, lo, hi)
    return {"wf": float(wf), "a": A, "b": B, "q_offset": qc, "q_range": np.minimum(qc - lo, hi - qc) * 0.95}


def candidate_states(engine, candidates: list, T: int, freq: float, device: bool = True, use_deg_vectorised_quirk: bool = False) -> dict:
    """States of ``len(candidates)`` candidate trajectories (dicts of ``fourier_coefficients``), T samples each at ``freq`` Hz, as ONE
    stacked batch ready for ``Engine.gram_grouped`` / ``candidate_dopt`` -- what ``computeTrajectoryDynamics`` builds per candidate on the
    host (trajectoryGenerator.py:83-155): joint states from the Fourier series, a stationary base (zero twist / acceleration / rpy).

    ``use_deg_vectorised_quirk``: with ``useDeg`` the reference's vectorised evaluation converts radians with ``deg2rad`` once more
    (lines 126-128 after a block that never produced degrees), i.e. its q, dq, ddq are pi / 180 of the per-sample generators' values;
    True reproduces that output."""
    C = len(candidates)
    n = engine.n
    nh = max(c["a"].shape[1] for c in candidates)
    A, B = np.zeros((C, n, nh)), np.zeros((C, n, nh))
    for i, c in enumerate(candidates):
        A[i, :, : c["a"].shape[1]] = c["a"]
        B[i, :, : c["b"].shape[1]] = c["b"]
    bounded = [c["q_range"] is not None for c in candidates]
    if any(bounded) and not all(bounded):
        raise ValueError("classic and bounded candidates cannot share one batch")
    st = engine.fourier_states([c["wf"] for c in candidates], A, B, np.stack([c["q_offset"] for c in candidates]), int(T), float(freq),
                               q_range=np.stack([c["q_range"] for c in candidates]) if all(bounded) else None, device=device)
    if use_deg_vectorised_quirk:
        st = {k: v * (np.pi / 180.0) for k, v in st.items()}
    if engine.floating:
        S = C * int(T)
        if device:
            import torch

            z = lambda k: torch.zeros((S, k), dtype=torch.float64, device=st["q"].device)  # noqa: E731
        else:
            z = lambda k: np.zeros((S, k))  # noqa: E731
        st.update(base_vel=z(6), base_acc=z(6), rpy=z(3))
    return st


def candidate_dopt_from_coefficients(engine, candidates: list, T: int, freq: float, independent_cols, dopt_regularization: float = 1e-4,
                                     YtY_prior=None, friction_sign_threshold: float = 0.02) -> np.ndarray:
    """D-optimality of every candidate without its samples ever leaving the device: Fourier coefficients -> states (``fbr_fourier_states``)
    -> one Gram per candidate (``fbr_gram_grouped``) -> eigenvalues on the host (trajectoryOptimizer.py:240-272 per candidate).
    ``friction_sign_threshold``: the Coulomb column is tanh(dq / threshold) -- callers with an option dict pass
    ``opt.get('frictionSignThreshold', 0.02)`` (helpers.py getFrictionSignSeries, model.py:757), the value ``Model`` uses on the host path."""
    st = candidate_states(engine, candidates, T, freq, device=True)
    if engine.friction:
        import torch

        st["sign"] = torch.tanh(st["dq"] / float(friction_sign_threshold))
    return candidate_dopt(engine, st, len(candidates), independent_cols, dopt_regularization, YtY_prior=YtY_prior)
