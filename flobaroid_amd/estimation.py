"""Estimators fed by the fused GPU reductions (consumer side of the hot path).

The reference solves its least-squares problems on the materialised tall matrix
(``identifier.py:683-737``, ``sdp.py:456-487``).  Every quantity it derives there is a function of the
small reductions the GPU pass returns:

    G_aug = [YStd | tau | cf]^T [YStd | tau | cf]           (fbr_gram_accumulate)
    R_aug with R_aug^T R_aug = G_aug, computed without squaring (fbr_tsqr)

so the functions below reproduce the reference's results from (P+k) x (P+k) inputs, with the same
LAPACK semantics (cut-offs relative to the TALL row count, sign conventions of numpy.linalg.qr).
"""
from __future__ import annotations

import numpy as np
import numpy.linalg as la
import scipy.linalg as sla


def select_aug(M_aug: np.ndarray, cols, P: int) -> np.ndarray:
    """Rows/columns of an augmented (P+k) matrix for the identified columns ``cols`` plus all k rhs columns."""
    k = M_aug.shape[0] - P
    sel = np.concatenate((np.asarray(cols, dtype=np.int64), np.arange(P, P + k)))
    return M_aug[np.ix_(sel, sel)]


def r_from_gram(G: np.ndarray) -> np.ndarray:
    """Upper-triangular R with R^T R = G for a (possibly rank-deficient) PSD Gram, via eigen-decomposition
    + QR (no Cholesky breakdown).  Accuracy is limited to sqrt(eps)*||Y|| -- prefer the TSQR factor."""
    w, V = la.eigh((G + G.T) * 0.5)
    w = np.clip(w, 0.0, None)
    return la.qr((V * np.sqrt(w)).T, mode="r")


def qr_subset(R_aug: np.ndarray, cols, P: int) -> np.ndarray:
    """R factor (positive diagonal) of [YStd[:, cols] | rhs] from the full augmented factor:
    qr(R_aug[:, subset]) -- a (P+k) x (nb+k) host problem."""
    k = R_aug.shape[0] - P
    sel = np.concatenate((np.asarray(cols, dtype=np.int64), np.arange(P, P + k)))
    R = la.qr(R_aug[:, sel], mode="r")
    sgn = np.sign(np.diag(R))
    sgn[sgn == 0] = 1.0
    return R * sgn[:, None]


def lstsq_from_R(Rb_aug: np.ndarray, nb: int, rhs_col: int, num_rows: int):
    """``numpy.linalg.lstsq(YBase, tau)`` (identifier.py:712) from the factor of [YBase | rhs...].

    rcond follows NumPy's default on the TALL problem: eps * max(M, N) with M = num_rows."""
    R1 = Rb_aug[:nb, :nb]
    z = Rb_aug[:nb, nb + rhs_col]
    U, s, Vt = la.svd(R1)
    rcond = np.finfo(np.float64).eps * max(num_rows, nb)
    keep = s > rcond * s[0]
    x = Vt[keep].T @ ((U[:, keep].T @ z) / s[keep])
    return x, s


def pinv_apply_from_R(Rb_aug: np.ndarray, nb: int, rhs_col: int):
    """``numpy.linalg.pinv(YBase).dot(v)`` (identifier.py:709,718) for a rhs column v of the augmented matrix
    (pinv default cut-off: 1e-15 * s_max)."""
    R1 = Rb_aug[:nb, :nb]
    z = Rb_aug[:nb, nb + rhs_col]
    U, s, Vt = la.svd(R1)
    keep = s > 1e-15 * s[0]
    return Vt[keep].T @ ((U[:, keep].T @ z) / s[keep])


def identify_base_parameters(R_aug: np.ndarray, independent_cols, P: int, num_rows: int, add_contacts: bool = True):
    """xBase exactly as ``Identification.identifyBaseParameters`` computes it (identifier.py:709-718) from the
    augmented factor of [YStd | tau | contactForcesSum]."""
    nb = len(independent_cols)
    Rb = qr_subset(R_aug, independent_cols, P)
    xBase, s = lstsq_from_R(Rb, nb, 0, num_rows)
    if add_contacts and Rb.shape[1] > nb + 1:
        xBase = xBase - pinv_apply_from_R(Rb, nb, 1)
    return xBase, Rb, s


def residual_sq_from_R(Rb_aug: np.ndarray, nb: int, x: np.ndarray, rhs_cols=(0,), signs=(1.0,)) -> float:
    """|| sum_i signs[i]*rhs_i - YBase x ||^2 from the augmented factor (used for rho2, sdp.py:482-485, and for
    sigma_rho, identifier.py:357-358): the norm of R_aug [ -x ; signs ]."""
    k = Rb_aug.shape[1] - nb
    v = np.zeros(nb + k)
    v[:nb] = -np.asarray(x)
    for c, sg in zip(rhs_cols, signs):
        v[nb + c] = sg
    return float(np.sum((Rb_aug @ v) ** 2))


def std_dev_for_params(Rb_aug: np.ndarray, nb: int, xBase: np.ndarray, rho: float, num_rows: int) -> np.ndarray:
    """Relative standard deviations p_sigma_x (identifier.py:343-370): C_xx = sigma_rho * pinv(YBase^T YBase)."""
    sigma_rho = rho / (num_rows - nb)
    R1 = Rb_aug[:nb, :nb]
    C = sigma_rho * sla.pinv(R1.T @ R1)
    p = np.sqrt(np.diag(C))
    nz = xBase != 0
    p[nz] /= np.abs(xBase[nz])
    return p


def sdp_inputs(R_aug: np.ndarray, independent_cols, K: np.ndarray, P: int, xBase: np.ndarray):
    """What ``SDP.identifyFeasibleStandardParameters`` derives from ``la.qr(YBase)`` (sdp.py:470-487):
    R1 (sign-normalised, the objective is invariant to row sign flips), rho1 = Q1^T tau,
    contactForces = Q1^T cf, rho2_norm_sqr = ||tau - cf - YBase xBase||^2 and R1 @ K."""
    nb = len(independent_cols)
    Rb = qr_subset(R_aug, independent_cols, P)
    R1 = Rb[:nb, :nb].copy()
    rho1 = Rb[:nb, nb].copy()
    k = Rb.shape[1] - nb
    cf = Rb[:nb, nb + 1].copy() if k > 1 else np.zeros(nb)
    if k > 1:
        rho2 = residual_sq_from_R(Rb, nb, xBase, rhs_cols=(0, 1), signs=(1.0, -1.0))
    else:
        rho2 = residual_sq_from_R(Rb, nb, xBase)
    return {"R1": R1, "rho1": rho1, "contactForces": cf, "rho2_norm_sqr": rho2, "R1_K": R1 @ K}


def observability_weights(R1_K: np.ndarray) -> np.ndarray:
    """Per-parameter CAD-pull weights of ``SDP._observabilityWeights`` (sdp.py:295-315), same expression: with the reduced
    normal matrix M = (R1 K)^T (R1 K) and the ridge eps = 1e-6 * trace(M) / n, obs_std = sqrt(clip(diag((M + eps I)^-1), 0)),
    normalised by the median of its positive entries and clipped to [0.1, 100].  Ordered like the columns of ``R1_K``."""
    n = R1_K.shape[1]
    M = R1_K.T @ R1_K
    ridge = 1e-6 * float(np.trace(M)) / n
    # diag((M + ridge I)^-1) from the solution of (M + ridge I) X = I.  The matrix is badly conditioned (2e7 on WALK-MAN: the ridge is
    # 1e-6 of the mean eigenvalue), so the weights depend on the factorisation to ~1e-9: the general LU solve (gesv) is what the
    # reference's la.inv runs (sdp.py:309-315) and reproduces its numbers to the bit; a Cholesky solve differs by 2e-9
    X = la.solve(M + ridge * np.eye(n), np.eye(n))
    obs_std = np.sqrt(np.maximum(np.einsum("ii->i", X), 0.0))
    scale = np.median(obs_std[obs_std > 0]) if np.any(obs_std > 0) else 1.0
    return np.clip(obs_std / float(scale), 0.1, 100.0)


def sdp_regularized_system(sdp_in: dict, xStdModel: np.ndarray, identified_params, non_id, base_error: float, regularization_factor: float,
                           mode: str = "uniform", delete_cols=()) -> dict:
    """The linear system the SDP's Schur block is built from (sdp.py:487-531): the residual is
    e = rho1_hat - contactForces_hat - Y_combined @ x  over the identifiable standard parameters x, with
    [R1 K; diag(w)] x ~ [rho1; diag(w) xStdModel] -- the CAD regularisation rows that keep the null-space part near the a-priori
    model.  ``mode`` = opt['cadRegularizationMode']: 'uniform' pulls the non-identifiable parameters with one weight
    base_error / len * regularizationFactor, 'observability' every identifiable parameter with ``observability_weights``,
    'geometric' / regularisation off (``regularization_factor`` = 0 or None) adds no rows.

    ``sdp_in``: the dict of ``sdp_inputs`` (R1_K's columns must already exclude ``delete_cols``); ``identified_params`` /
    ``non_id`` as on the Model.  Returns Y_combined, rho1_hat, contactForces_hat, reg_params, reg_weights (dict)."""
    delete_cols = set(int(c) for c in delete_cols)
    idable = sorted(set(int(p) for p in identified_params).difference(delete_cols))
    index = {p: i for i, p in enumerate(idable)}
    R1_K, rho1, cf = sdp_in["R1_K"], sdp_in["rho1"], sdp_in["contactForces"]
    if R1_K.shape[1] != len(idable):
        raise ValueError("R1_K must have one column per identifiable parameter (delete_cols removed)")
    reg_params: list[int] = []
    reg_weights: dict[int, float] = {}
    if regularization_factor:
        p_nid = sorted(set(int(p) for p in non_id).difference(delete_cols).intersection(idable))
        if mode == "observability":
            w = observability_weights(R1_K)
            reg_params = idable
            base = (float(base_error) / len(reg_params)) * regularization_factor
            reg_weights = {p: base * float(w[index[p]]) for p in reg_params}
        elif mode == "geometric":
            pass  # a log-det prior in the objective (sdp.py:506-510), no residual rows
        elif p_nid:
            reg_params = p_nid
            base = (float(base_error) / len(p_nid)) * regularization_factor
            reg_weights = {p: base for p in p_nid}
    if not reg_params:
        return {"Y_combined": R1_K, "rho1_hat": rho1, "contactForces_hat": cf, "reg_params": [], "reg_weights": {}}
    Y_bot = np.zeros((len(reg_params), len(idable)))
    rho_bot = np.zeros(len(reg_params))
    for i, p in enumerate(reg_params):
        Y_bot[i, index[p]] = reg_weights[p]
        rho_bot[i] = reg_weights[p] * xStdModel[p]
    return {"Y_combined": np.vstack([R1_K, Y_bot]), "rho1_hat": np.concatenate((rho1, rho_bot)),
            "contactForces_hat": np.concatenate((cf, np.zeros(len(reg_params)))), "reg_params": reg_params, "reg_weights": reg_weights}


def find_std_from_base(K: np.ndarray, xBase: np.ndarray) -> np.ndarray:
    """``findStdFromBaseParameters`` (identifier.py:328-341): xStd = pinv(K) xBase."""
    return la.pinv(K).dot(xBase)


def identify_standard_direct(R_aug: np.ndarray, P: int, num_base_params: int, rhs_col: int = 0):
    """``Identification.identifyStandardParametersDirect`` (identifier.py:792-814) without the tall SVD:
    with [YStd | tau] = Q R_aug, the thin SVD YStd = U S V^T has U = Q[:, :P] U_r, (U_r, S, V) = svd(R_aug[:P, :P]),
    and U_1^T tau = U_r[:, :nb]^T R_aug[:P, P+rhs_col].  Returns x_est = V_1 S_1^-1 U_1^T tau and the singular values."""
    Rpp = R_aug[:P, :P]
    z = R_aug[:P, P + rhs_col]
    U, s, Vt = la.svd(Rpp)
    nb = int(num_base_params)
    x = Vt[:nb].T @ ((U[:, :nb].T @ z) / s[:nb])
    return x, s


def identify_standard_essential(R_aug: np.ndarray, P: int, x_std_essential: np.ndarray, num_essential: int, rhs_col: int = 0):
    """``Identification.identifyStandardEssentialParameters`` (identifier.py:816-838) from the small factor:
    W_e = YStd diag(x_e) = Q (R_pp diag(x_e)), thin SVD (U_r, s, V) of R_pp diag(x_e), and
    x = diag(x_e) V_1e S_1e^-1 U_1e^T tau with U_1e^T tau = U_r[:, :ne]^T R_aug[:P, P+rhs_col]."""
    xe = np.asarray(x_std_essential, dtype=float)
    Re = R_aug[:P, :P] * xe[None, :]
    z = R_aug[:P, P + rhs_col]
    U, s, Vt = la.svd(Re)
    ne = int(num_essential)
    return xe * (Vt[:ne].T @ ((U[:, :ne].T @ z) / s[:ne]))


def wls_row_weights(p_sigma_x: np.ndarray, num_used_samples: int, rows_total: int) -> np.ndarray:
    """Row weights of the reference's IDIM-WLS pass (identifier.py:739-790): the diagonal of
    ``spdiags(np.repeat([1 / p_sigma_x], num_used_samples), 0, r, r)``, verbatim -- i.e. entry i is
    1 / p_sigma_x[i // num_used_samples] (the reference repeats the per-parameter deviations, not per-joint ones;
    off in every shipped config, SURVEY 8(a) flags it unpinned).  Pass as ``w`` to ``Engine.gram`` / ``Engine.tsqr``."""
    d = np.repeat(1.0 / np.asarray(p_sigma_x, dtype=float), int(num_used_samples))
    if d.shape[0] < rows_total:
        raise ValueError("WLS: fewer weights than rows (num_base_params < rows per sample); the reference's expression is undefined here")
    return d[:rows_total].copy()


def wls_reference_weights(p_sigma_x: np.ndarray, num_used_samples: int, rows_total: int) -> np.ndarray:
    """The diagonal the reference's WLS pass really builds (identifier.py:769-774): ``scipy.sparse.spdiags`` of
    ``np.repeat([1 / p_sigma_x], num_used_samples)`` on an r x r matrix -- entries beyond r are ignored, and when the repeated
    vector is SHORTER than r the rest of the diagonal is zero (those rows drop out of the fit)."""
    d = np.repeat(1.0 / np.asarray(p_sigma_x, dtype=float), int(num_used_samples))[:rows_total]
    return np.concatenate((d, np.zeros(rows_total - d.shape[0])))


def wls_reference_compat_rhs(w: np.ndarray, tau: np.ndarray, contact_forces: np.ndarray | None = None) -> np.ndarray:
    """Right-hand sides that make the weighted reductions reproduce the reference's WLS numbers (``opt['wlsReferenceCompat']``).

    identifier.py:777-789 weights YBase (and model.tau) with G = diag(w) but then solves with the function's LOCAL, unweighted
    ``tau``: xBase = lstsq(G YBase, tau) - pinv(G YBase) contactForcesSum.  ``fbr_gram_accumulate`` / ``fbr_tsqr`` apply the
    row weight to Y and to the rhs columns alike, so the same numbers come out of rhs columns divided by w beforehand
    (rows of weight zero contribute nothing on either side: 0)."""
    w = np.asarray(w, dtype=float)
    cols = [np.asarray(tau, dtype=float).reshape(-1)]
    cols.append(np.zeros_like(cols[0]) if contact_forces is None else np.asarray(contact_forces, dtype=float).reshape(-1))
    out = np.zeros((w.shape[0], 2))
    nz = w != 0
    for c, v in enumerate(cols):
        out[nz, c] = v[nz] / w[nz]
    return out


def identify_base_parameters_wls(engine, states: dict, tau: np.ndarray, contact_forces, independent_cols, p_sigma_x: np.ndarray,
                                 reference_compat: bool | None = None, opt: dict | None = None):
    """The IDIM-WLS pass (identifier.py:739-790) as ONE weighted TSQR of [YBase | tau | contactForcesSum] on the device.

    ``reference_compat=False``: the textbook form -- the row weights apply to Y, tau and the contact forces alike
    (``wls_row_weights``).  ``reference_compat=True`` (``opt['wlsReferenceCompat']``): the reference's numbers --
    ``wls_reference_weights`` on Y only (``wls_reference_compat_rhs``).  ``reference_compat=None`` reads the option dictionary.
    Returns (xBase, R_aug_base, w)."""
    if reference_compat is None:
        reference_compat = bool((opt or {}).get("wlsReferenceCompat", 0))
    S = np.asarray(states["q"]).shape[0]
    rows = S * engine.rows
    ic = np.asarray(independent_cols, dtype=np.int32)
    tau = np.asarray(tau, dtype=float).reshape(-1)
    if reference_compat:
        w = wls_reference_weights(p_sigma_x, S, rows)
        rhs = wls_reference_compat_rhs(w, tau, contact_forces)
    else:
        w = wls_row_weights(p_sigma_x, S, rows)
        rhs = np.column_stack((tau, np.zeros(rows) if contact_forces is None else np.asarray(contact_forces, dtype=float).reshape(-1)))
    Rb = np.asarray(engine.tsqr(states, rhs=np.ascontiguousarray(rhs), w=w, cols=ic))
    nb = ic.size
    xBase, _ = lstsq_from_R(Rb, nb, 0, rows)
    if contact_forces is not None:
        xBase = xBase - pinv_apply_from_R(Rb, nb, 1)
    return xBase, Rb, w


def _regularized_neg_log_det(ev: np.ndarray, dopt_regularization: float) -> np.ndarray:
    """-sum(log(max(ev + delta, 1e-300))) along the last axis with the reference's per-trajectory
    delta = doptRegularization * max(lambda_max, 1e-30) (trajectoryOptimizer.py:267-272; eigvalsh returns ascending values)."""
    delta = float(dopt_regularization) * np.maximum(ev[..., -1:], 1e-30)
    return -np.sum(np.log(np.maximum(ev + delta, 1e-300)), axis=-1)


def d_optimality(G_aug: np.ndarray, independent_cols, dopt_regularization: float = 1e-4, YtY_prior: np.ndarray | None = None) -> float:
    """Excitation criterion of the trajectory optimiser (excitation/trajectoryOptimizer.py:259-272), same expression:
    YtY = YBase^T YBase (= G[ic, ic] of the fused Gram) [+ YtY_prior, the sequential-design term :263-265],
    delta = doptRegularization (config default 1e-4) * max(lambda_max, 1e-30), -sum(log(max(eig + delta, 1e-300)))."""
    ic = np.asarray(independent_cols, dtype=np.int64)
    YtY = G_aug[np.ix_(ic, ic)]
    if YtY_prior is not None:
        YtY = YtY + np.asarray(YtY_prior, dtype=np.float64)
    return float(_regularized_neg_log_det(la.eigvalsh(YtY), dopt_regularization))


def d_optimality_batch(G_groups: np.ndarray, independent_cols, dopt_regularization: float = 1e-4,
                       YtY_prior: np.ndarray | None = None) -> np.ndarray:
    """``d_optimality`` of every candidate trajectory of a batch: ``G_groups`` (ngroups, Pa, Pa) from
    ``Engine.gram_grouped`` (one pass over all candidates; the optimiser's inner loop, trajectoryOptimizer.py:248-272).
    delta is taken per candidate from that candidate's own largest eigenvalue, as the reference does per trajectory."""
    ic = np.asarray(independent_cols, dtype=np.int64)
    YtY = G_groups[:, ic[:, None], ic[None, :]]
    if YtY_prior is not None:
        YtY = YtY + np.asarray(YtY_prior, dtype=np.float64)[None]
    return _regularized_neg_log_det(la.eigvalsh(YtY), dopt_regularization)


def n_observable_base_params(G_aug: np.ndarray, independent_cols, dopt_regularization: float = 1e-4) -> int:
    """Eigenvalues of YBase^T YBase above the regularisation threshold (trajectoryOptimizer.py:275; the count
    trajectory.py:225-264 stores as ``n_observable_base_params``)."""
    ic = np.asarray(independent_cols, dtype=np.int64)
    ev = la.eigvalsh(G_aug[np.ix_(ic, ic)])
    return int(np.sum(ev > float(dopt_regularization) * max(float(ev[-1]), 1e-30)))


def base_wrench_row_mask(num_samples: int, rows: int) -> np.ndarray:
    """0/1 row weights selecting the 6 base-wrench rows of every sample (``_extractBaseWrenchRows``,
    identifier.py:629-636) -- pass as ``w`` to ``Engine.gram`` / ``Engine.tsqr``."""
    w = np.zeros((num_samples, rows))
    w[:, :6] = 1.0
    return w.reshape(-1)


def trajectory_row_weights(residual_bw: np.ndarray, file_boundaries, num_used_samples: int, skip: int = 0) -> np.ndarray:
    """Per-(file, wrench component) inverse-noise row weights of ``_extractBaseWrenchRows`` (identifier.py:654-679).

    ``residual_bw`` (S, 6): base-wrench residual tau_bw - YBase_bw x_pre of a cheap OLS pre-pass (obtainable with
    ``Engine.predict``).  Returns the (S, 6) weights (mean ~ 1); files with <= 6 samples keep weight 1."""
    nw = 6  # wrench components
    # file of every used sample (sample i of the used ones is loaded sample i (skip + 1), data.py:50,115)
    owner = np.searchsorted(np.asarray(file_boundaries), np.arange(num_used_samples) * (skip + 1), side="right") - 1
    n_files = len(file_boundaries) - 1
    count = np.bincount(owner, minlength=n_files)[:n_files]
    sumsq = np.zeros((n_files, nw))
    np.add.at(sumsq, owner, np.square(residual_bw))
    # rms residual per (file, component); files too short to estimate one (<= 6 samples) keep sigma = 1
    rms = np.sqrt(sumsq / np.maximum(count, 1)[:, None])
    sigma = np.where((count > nw)[:, None], rms, 1.0)
    return (np.mean(sigma) / np.maximum(sigma, 1e-12))[owner]


def post_identify_friction(tau_residual_2d: np.ndarray, velocities: np.ndarray, velocities_for_sign: np.ndarray,
                           sign_series: np.ndarray, fb: int, deadzone: float = 0.0, lambda_fv: float = 0.0,
                           alpha_fv: float = 0.0, fv_apriori: np.ndarray | None = None) -> dict:
    """Per-joint friction refit on the torque residual after the inertial identification
    (``Identification._postIdentifyFriction``, identifier.py:979-1099).

    ``tau_residual_2d`` (S, fb + n) = tau_measured - YStd[:, :10L] x_inertial (the streaming prediction pass:
    ``Engine.predict`` with the friction slots of x zeroed); ``velocities`` (S, n) the filtered joint velocities,
    ``velocities_for_sign`` / ``sign_series`` (S, n) from ``helpers.getFrictionSignVelocities / getFrictionSignSeries``.
    Fits [Fc, Fv, off] of tau_res = Fc * sign + Fv * v + off by OLS per joint with the reference's Swevers dead zone
    (samples with |v_sign| < deadzone dropped unless < 30 remain or one direction is missing), the Tikhonov pull of Fv
    towards ``fv_apriori`` (weight ``lambda_fv``, or ``alpha_fv`` x median kept velocity energy) and the clamp Fv >= 0.
    Returns {"Fc", "Fv", "off", "deadzone_kept", "lambda_fv", "fv_energy"}."""
    S, n = velocities.shape
    keep_masks = []
    fv_energy = np.zeros(n)
    kept = np.zeros(n)
    for j in range(n):
        vs = velocities_for_sign[:, j]
        if deadzone > 0:
            keep = np.abs(vs) >= deadzone
            if np.count_nonzero(keep) < 10 * 3 or not (vs[keep] > 0).any() or not (vs[keep] < 0).any():
                keep = np.ones(S, dtype=bool)
        else:
            keep = np.ones(S, dtype=bool)
        keep_masks.append(keep)
        kept[j] = np.count_nonzero(keep) / S
        fv_energy[j] = float(np.sum(velocities[keep, j] ** 2))
    lam = alpha_fv * float(np.median(fv_energy)) if alpha_fv > 0 else float(lambda_fv)
    if lam > 0 and fv_apriori is None:
        raise ValueError("Fv regularisation needs the a-priori viscous friction of every joint")
    Fc, Fv, off = np.zeros(n), np.zeros(n), np.zeros(n)
    for j in range(n):
        keep = keep_masks[j]
        A = np.column_stack([sign_series[keep, j], velocities[keep, j], np.ones(np.count_nonzero(keep))])
        b = tau_residual_2d[keep, fb + j]
        if lam > 0:
            wq = np.sqrt(lam)
            A = np.vstack((A, [0.0, wq, 0.0]))
            b = np.append(b, wq * fv_apriori[j])
        p = la.lstsq(A, b, rcond=None)[0]
        Fc[j], Fv[j], off[j] = p[0], max(p[1], 0.0), p[2]
    return {"Fc": Fc, "Fv": Fv, "off": off, "deadzone_kept": kept, "lambda_fv": lam, "fv_energy": fv_energy}
