"""End-to-end parity of the identification pipeline on WALK-MAN (BASELINE configs[3], [4]) on the GPU:

  trajectory samples -> Model.computeRegressors -> fused Gram / Householder TSQR -> base parameters -> standard parameters,
  SDP inputs, direct / essential solves

against the reference's CPU path executed on the oracle's materialised regressor (identifier.py:709-718, 328-341;
sdp.py:470-487; identifier.py:792-855).  north_star's bars: identical base-parameter index sets, identified standard parameters within
1e-6 relative Frobenius.  At sizes the host cannot materialise (4 M samples = 538 GB of YStd) the lazy YStd / YBase objects are
used and the pipeline is held to size-independent properties.
"""
import os

import numpy as np
import numpy.linalg as la
import pytest

from common import load_topo, random_states
from test_gpu_model import _opt, _synth

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reduction_mode")]  # three modes of the column reductions: conftest.py


def _walkman_model(tmp_path, seed=11, **over):
    from flobaroid_amd.model import Model

    path = str(tmp_path / "walkman_apriori.topology.json")
    topo = load_topo("walkman_apriori")
    topo.save_json(path)
    opt = _opt(floatingBase=1, randomSamples=10000, minTol=0.005, **over)  # configs/walkman_full.yaml:243-247
    np.random.seed(seed)
    return topo, opt, Model(opt, path)


def _cpu_structural(topo, model, seed=11):
    """The CPU path's base projection: the same random states (same global-RNG order), regressors from the oracle, Gram summed on the
    host, the same pivoted-QR rule -> independent columns and K (model.py:634-894)."""
    from flobaroid_amd.model import pivoted_qr
    from oracle.oracle import OracleModel

    np.random.seed(seed)
    st = model._random_states(10000)
    om = OracleModel(topo, floating=1)
    G = np.zeros((480, 480))
    for a in range(0, 10000, 500):
        Y = om.regressor({k: v[a:a + 500] for k, v in st.items()})
        G += Y.T @ Y
    Q, R, P = pivoted_qr(G)
    r = int(np.count_nonzero(np.abs(np.diag(R)) > 0.005))
    Pp = np.zeros((480, 480))
    Pp[np.arange(480), P] = 1.0
    import scipy.linalg as sla

    deps = sla.inv(R[:r, :r]).dot(R[:r, r:])
    deps[np.abs(deps) < 0.005] = 0
    K = Pp.T[:, :r].T + deps.dot(Pp.T[:, r:].T)
    return P[:r], K


def test_walkman_xstd_within_1e6_of_the_cpu_path(tmp_path):
    """WALK-MAN floating base, 20 000 synthetic samples (generator of tests/test_identification.py:25-93: seeded states inside the
    joint limits, tau = ID(a-priori) + N(0, 0.05^2); the host still materialises the 2.7 GB YStd): GPU tsqr ->
    identify_base_parameters -> find_std_from_base and sdp_inputs against NumPy on the oracle's matrix."""
    from flobaroid_amd import estimation as est
    from flobaroid_amd.data import Data
    from oracle.oracle import OracleModel

    topo, opt, model = _walkman_model(tmp_path)
    assert model.num_base_params == 213
    ic_cpu, K_cpu = _cpu_structural(topo, model)
    assert np.array_equal(np.asarray(model.independent_cols), ic_cpu)          # identical index sets, entry by entry
    S = 20000
    meas, st, tau_full = _synth(topo, S, 42, 1)
    data = Data(opt)
    data.init_from_data(meas)
    opt["materializeLimitBytes"] = 1e6   # YStd / YBase stay on the device side (LazyRegressor)
    model.computeRegressors(data)
    from flobaroid_amd.model import LazyRegressor

    assert isinstance(model.YStd, LazyRegressor) and isinstance(model.YBase, LazyRegressor)
    assert model.YStd.shape == (S * 35, 480) and model.YBase.shape == (S * 35, 213)
    # ---- CPU path (the reference's calls on the materialised matrix)
    om = OracleModel(topo, floating=1)
    Y = om.regressor(st)
    sim = om.inverse_dynamics(st, topo.x_std())
    tau_cpu = np.concatenate((sim[:, :6], meas["torques"]), axis=1).reshape(-1)   # model.py:398-413: simulated base wrench
    assert la.norm(model.tau - tau_cpu) <= 1e-11 * la.norm(tau_cpu)
    YB = np.ascontiguousarray(Y[:, ic_cpu])
    xBase_cpu = la.lstsq(YB, tau_cpu, rcond=None)[0]                                # identifier.py:712
    xStd_cpu = la.pinv(K_cpu).dot(xBase_cpu)                                        # identifier.py:337
    # ---- GPU path
    rhs = np.stack((model.tau, model.contactForcesSum), axis=1)
    R_aug = model.engine.tsqr(model._states, rhs=rhs)
    xBase, Rb, sv = est.identify_base_parameters(R_aug, model.independent_cols, 480, S * 35)
    xStd = est.find_std_from_base(model.K, xBase)
    assert la.norm(xBase - xBase_cpu) <= 1e-7 * la.norm(xBase_cpu)
    err = la.norm(xStd - xStd_cpu) / la.norm(xStd_cpu)
    print("WALK-MAN 20k: ||xStd_gpu - xStd_cpu|| / ||xStd_cpu|| =", err)
    assert err <= 1e-6                                                             # north_star
    # the lazy regressors behave like the matrices for what the callers do with them
    assert la.norm(model.YStd.dot(xStd) - Y @ xStd) <= 1e-10 * la.norm(Y @ xStd)
    assert la.norm(model.YBase @ xBase - YB @ xBase) <= 1e-10 * la.norm(YB @ xBase)
    assert np.abs(np.asarray(model.YBase)[:700] - YB[:700]).max() <= 1e-11 * np.abs(YB).max()
    # ---- SDP prologue (sdp.py:470-487) from the GPU factor vs numpy.linalg.qr(YBase)
    sin = est.sdp_inputs(R_aug, model.independent_cols, model.K, 480, xBase)
    Qn, Rn = la.qr(YB)
    sg = np.sign(np.diag(Rn))
    R1_cpu, rho1_cpu = Rn * sg[:, None], (Qn.T @ tau_cpu) * sg
    assert la.norm(sin["R1"] - R1_cpu) <= 1e-9 * la.norm(R1_cpu)
    assert la.norm(sin["rho1"] - rho1_cpu) <= 1e-9 * la.norm(rho1_cpu)
    rho2_cpu = la.norm(tau_cpu - YB @ xBase_cpu) ** 2
    assert abs(sin["rho2_norm_sqr"] - rho2_cpu) <= 1e-8 * rho2_cpu
    assert la.norm(sin["R1_K"] - R1_cpu @ K_cpu) <= 1e-6 * la.norm(R1_cpu @ K_cpu)
    # observability weights / regularisation rows of the SDP (sdp.py:295-315, 487-531) straight from the GPU factor
    w_gpu, w_cpu = est.observability_weights(sin["R1_K"]), est.observability_weights(R1_cpu @ K_cpu)
    assert np.abs(w_gpu - w_cpu).max() <= 1e-6 * w_cpu.max()
    reg = est.sdp_regularized_system(sin, model.xStdModel, model.identified_params, model.non_id, 1.0, 1000.0, "uniform")
    assert reg["Y_combined"].shape == (213 + len(model.non_id), 480)
    # ---- post-identification friction refit (identifier.py:979-1099) on the residual of one streaming prediction pass
    # (through xStd = pinv(K) xBase and the full YStd, as estimateRegressorTorques does, identifier.py:135-141: K's entries below
    # minTol are zeroed, model.py:891, so YStd xStd is not YBase xBase -- the CPU path has the same residual)
    resid = (model.tau - model.YStd.dot(xStd)).reshape(S, 35)
    resid_cpu = tau_cpu - Y @ xStd_cpu
    assert la.norm(resid.reshape(-1) - resid_cpu) <= 1e-9 * la.norm(resid_cpu)
    fr = est.post_identify_friction(resid, meas["velocities"], meas["velocities"], np.tanh(meas["velocities"] / 0.02), 6)
    fr_cpu = est.post_identify_friction(resid_cpu.reshape(S, 35), meas["velocities"], meas["velocities"], np.tanh(meas["velocities"] / 0.02), 6)
    for key in ("Fc", "Fv", "off"):
        assert np.abs(fr[key] - fr_cpu[key]).max() <= 1e-8 * max(1.0, np.abs(fr_cpu[key]).max())


def test_xstd_from_the_tie_rule_set_equals_xstd_from_lapacks_own_set(tmp_path):
    """The base-parameter index set chosen with the documented tie rule (pivotTieTolerance = 1e-7, flobaroid_amd/model.py: pivoted_qr)
    differs from the one LAPACK's own last-bit tie breaking gives (pivotTieTolerance = 0: the reference's call, model.py:809,871-884) in
    the tied positions only -- columns that are equal up to a rigid transform, so both sets span the same column space of the
    regressor.  Checked on WALK-MAN, 20 000 synthetic samples:
      * the fitted torques YBase xBase are the same (1e-9): the two base parametrisations describe the same least-squares fit;
      * the identified STANDARD parameters pinv(K) xBase agree within north_star's 1e-6 when K carries the exact linear dependencies
        R1^-1 R2;
      * with the reference's K -- entries below minTol = 0.005 set to zero, model.py:889-891 -- they agree to that threshold only
        (a few 1e-3): the reference's own xStd moves by as much whenever its LAPACK flips one of the 25 coin-flip pivots."""
    import scipy.linalg as sla

    from flobaroid_amd import estimation as est
    from flobaroid_amd.data import Data

    xs, xs_exact, fits, sets = [], [], [], []
    meas = None
    for name, tol in (("rule", None), ("lapack", 0)):
        d = tmp_path / name
        d.mkdir()
        over = {} if tol is None else {"pivotTieTolerance": tol}
        topo, opt, model = _walkman_model(d, seed=11, **over)
        r = model.num_base_params
        assert r == 213
        if meas is None:
            meas, st, tau_full = _synth(topo, 20000, 42, 1)
        data = Data(opt)
        data.init_from_data(meas)
        opt["materializeLimitBytes"] = 1e6
        model.computeRegressors(data)
        rhs = np.stack((model.tau, model.contactForcesSum), axis=1)
        R_aug = model.engine.tsqr(model._states, rhs=rhs)
        xBase, Rb, sv = est.identify_base_parameters(R_aug, model.independent_cols, 480, 20000 * 35)
        xs.append(est.find_std_from_base(model.K, xBase))
        K_exact = model.Pb.T + sla.solve_triangular(model.R[:r, :r], model.R[:r, r:]).dot(model.Pd.T)   # no minTol threshold
        xs_exact.append(la.pinv(K_exact).dot(xBase))
        fits.append(model.YBase @ xBase)
        sets.append(set(int(c) for c in model.independent_cols))
    ndiff = len(sets[0] - sets[1])
    e_fit = la.norm(fits[0] - fits[1]) / la.norm(fits[1])
    e_exact = la.norm(xs_exact[0] - xs_exact[1]) / la.norm(xs_exact[1])
    e_thr = la.norm(xs[0] - xs[1]) / la.norm(xs[1])
    print(f"WALK-MAN: tie-rule set vs LAPACK's own set differ in {ndiff} of 213 columns; fitted torques {e_fit:.1e}; "
          f"xStd with exact K {e_exact:.1e}; xStd with the reference's thresholded K {e_thr:.1e}")
    assert ndiff > 0            # (otherwise the test compares a set with itself)
    assert e_fit <= 1e-9
    assert e_exact <= 1e-6
    assert e_thr <= 2e-2        # minTol-sized entries of K dropped (the reference's own sensitivity to its coin-flip pivots)


def test_direct_and_essential_solves_from_the_gpu_factor():
    """identifyStandardParametersDirect / identifyStandardEssentialParameters (identifier.py:792-855: thin SVD of the tall YStd and of
    YStd diag(x_e)) against estimation.identify_standard_direct / _essential fed with the GPU R_aug (A11)."""
    from flobaroid_amd import estimation as est
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    topo = load_topo("walkman_left_arm")
    rng = np.random.default_rng(8)
    S = 3000
    st = random_states(topo, S, rng, 1, use_limits=True)
    eng = Engine(topo, floating=1)
    om = OracleModel(topo, floating=1)
    Y = om.regressor(st)
    x_true = topo.x_std() * (1.0 + 0.1 * rng.standard_normal(90))
    tau = Y @ x_true + 0.05 * rng.standard_normal(Y.shape[0])
    R_aug = eng.tsqr(st, rhs=tau.reshape(-1, 1))
    nb = 59  # base rank of the floating left arm (tests/golden/structure.json)
    U, s, VH = la.svd(Y, full_matrices=False)                                  # identifier.py:796-809
    x_ref = VH.T[:, :nb] @ ((U[:, :nb].T @ tau) / s[:nb])
    x, sv = est.identify_standard_direct(R_aug, 90, nb)
    assert np.allclose(sv[:nb], s[:nb], rtol=1e-9)
    assert la.norm(x - x_ref) <= 1e-6 * la.norm(x_ref)
    xe = np.zeros(90)
    ess = np.sort(rng.choice(90, 40, replace=False))
    xe[ess] = x_true[ess]
    ne = 25
    Ue, se, VHe = la.svd(Y * xe[None, :], full_matrices=False)                 # identifier.py:816-838
    x_ess_ref = xe * (VHe.T[:, :ne] @ ((Ue[:, :ne].T @ tau) / se[:ne]))
    x_ess = est.identify_standard_essential(R_aug, 90, xe, ne)
    assert la.norm(x_ess - x_ess_ref) <= 1e-6 * la.norm(x_ess_ref)


def test_config5_4M_samples_through_the_lazy_path(tmp_path):
    """BASELINE configs[4]: WALK-MAN, 4 M samples (YStd would be 538 GB): Data -> Model.computeRegressors with lazy YStd / YBase,
    fused Gram, TSQR of [YBase | tau] streamed in four calls through R_in, SDP inputs, xStd.  Checked through properties that do
    not need the tall matrix: R^T R = G, streaming == one call, the OLS residual from the factor == the residual of a streaming
    prediction pass with the identified parameters, and xStd within 1e-6 of the solve from the full-width factor."""
    from flobaroid_amd import estimation as est
    from flobaroid_amd.data import Data
    from flobaroid_amd.model import LazyRegressor

    topo, opt, model = _walkman_model(tmp_path)
    S = 4_000_000
    rng = np.random.default_rng(2024)
    st = random_states(topo, S, rng, 1, use_limits=True)
    tau = model.engine.inverse_dynamics(st, topo.x_std())
    tau += rng.normal(0, 0.05, tau.shape)
    meas = {"positions": st["q"], "velocities": st["dq"], "accelerations": st["ddq"], "torques": tau[:, 6:], "times": np.arange(S) / 200.0,
            "base_velocity": st["base_vel"], "base_acceleration": st["base_acc"], "base_rpy": st["rpy"]}
    data = Data(opt)
    data.init_from_data(meas)
    model.computeRegressors(data)      # default materializeLimitBytes (8 GB) < 538 GB: lazy
    assert isinstance(model.YStd, LazyRegressor) and model.YStd.shape == (S * 35, 480) and model.YBase.shape == (S * 35, 213)
    ic = np.asarray(model.independent_cols)
    rhs = model.tau.reshape(-1, 1)
    eng = model.engine
    # TSQR of the base regressor [YBase | tau], streamed in four quarters through R_in
    Rb = None
    q = S // 4
    for i in range(4):
        sub = {k: v[i * q:(i + 1) * q] for k, v in model._states.items()}
        Rb = eng.tsqr(sub, rhs=rhs[i * q * 35:(i + 1) * q * 35], cols=ic, R_in=Rb)
    sel = np.concatenate((ic, [480]))
    Gb = model.G_aug[np.ix_(sel, sel)]
    assert la.norm(Rb.T @ Rb - Gb) <= 1e-11 * la.norm(Gb)
    Rb1 = eng.tsqr(model._states, rhs=rhs, cols=ic)
    sg, sg1 = np.sign(np.diag(Rb)), np.sign(np.diag(Rb1))
    assert la.norm(Rb * sg[:, None] - Rb1 * sg1[:, None]) <= 1e-9 * la.norm(Rb1)
    # SDP inputs and the OLS solution from the factor; the solve from the full-width factor (no column subset) agrees
    import scipy.linalg as sla

    nb = 213
    sgn = np.where(sg == 0, 1.0, sg)
    R1, rho1 = (Rb * sgn[:, None])[:nb, :nb], (Rb * sgn[:, None])[:nb, nb]
    xBase = sla.solve_triangular(R1, rho1)
    xStd = est.find_std_from_base(model.K, xBase)
    R_full = eng.tsqr(model._states, rhs=rhs)
    xBase2, _, _ = est.identify_base_parameters(R_full, ic, 480, 1, add_contacts=False)  # (num_rows = 1: rcond = eps * nb, nothing cut)
    assert la.norm(est.find_std_from_base(model.K, xBase2) - xStd) <= 1e-6 * la.norm(xStd)
    # residual of the base-parameter fit, two independent ways: the last diagonal entry of the factor vs a streaming prediction pass
    # through the lazy YBase (fbr_predict with the base columns)
    rho2 = float(Rb[nb, nb] ** 2)
    resid = model.tau - model.YBase.dot(xBase)
    assert abs(la.norm(resid) ** 2 - rho2) <= 1e-7 * rho2
    assert np.isfinite(model.YStd.dot(xStd)).all()
    assert 0.9 < rho2 / (0.05 ** 2 * (S * 29)) < 1.1    # the noise that went in (the 6 simulated base-wrench rows carry none)
    x_model = model.K.dot(model.xStdModel)
    assert la.norm(xBase - x_model) <= 0.02 * la.norm(x_model)
