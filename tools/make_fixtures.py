#!/usr/bin/env python3
"""Generate the committed data fixtures from the reference checkout (run in the build container only).

Reads DATA from /root/reference (URDF robot descriptions, a documentation table, one trajectory
npz) and writes small derived fixtures; no reference source code is copied.

  flobaroid_amd/robots/<name>.topology.json   kinematic tree + a-priori parameters extracted from
                                              model/<name>.urdf by flobaroid_amd.topology.parse_urdf
  tests/golden/kuka_tutorial_apriori.json     xStdModel[0:101] (8 decimals) and the nID flags printed in
                                              documentation/TUTORIAL.md:60-160  (known answer F9)
  tests/golden/kuka_trajectory_opt_1.npz      2409x7 positions/velocities/accelerations + times of
                                              model/kuka_lwr4.urdf.trajectory_opt_1.npz and its recorded
                                              n_observable_base_params (= 64)            (known answer F5)
  tests/golden/structure.json                 documented structure counts (links/DOF/base ranks)
  tests/golden/reference_joint_orders.json    the DOF serialisations the reference holds: the <joint> lists of model/*_regressor.xml
                                              ("same order as reported when running without supplying a regressor file",
                                              model/walkman_regressor.xml:1) and the list in configs/walkman_static.yaml:60-64
  tests/golden/ref_host_functions.npz         seeded inputs and the OUTPUTS OF THE REFERENCE'S OWN pure NumPy/SciPy host
                                              functions run here: Data.preprocess (identification/data.py:369-619),
                                              helpers.getFrictionSignVelocities / getFrictionSignSeries
                                              (identification/helpers.py:89-156), Data.init_from_files (data.py:55-146).
                                              Importing identification.* pulls in idyntree / colorama, which the image
                                              lacks: they are replaced by empty placeholder modules for the import only --
                                              none of the executed code paths touches them (joint channels, FT and plain
                                              npz concatenation).
  tests/golden/ref_estimators.npz             outputs of the reference's own estimator code on a seeded KUKA / threeLinks
                                              problem: Model.computeRegressorLinDepsQR (identification/model.py:832-1052),
                                              Identification.identifyBaseParameters / getStdDevForParams /
                                              findStdFromBaseParameters / identifyStandardParametersDirect /
                                              identifyStandardEssentialParameters / _extractBaseWrenchRows
                                              (identifier.py:328-370,617-855), called as unbound methods on plain attribute
                                              holders (the real constructors need iDynTree).  The regressor matrices come
                                              from this repository's CPU oracle; the fixture stores the states, not Y.
  tests/golden/ref_walkman.npz                the same reference code paths on WALK-MAN (48 links, 29 DOF, floating base): computeRegressors with
                                              contacts on both feet + friction + a-priori torques, getRandomRegressor (10000 random states),
                                              computeRegressorLinDepsQR (minTol 0.005: P, rank, independent columns, K, non_id) and
                                              SDP._observabilityWeights (identification/sdp.py:295-315)
  tests/golden/ref_blocks_wls.npz             (round 3) the reference's own block selection -- Data.getBlockStats / hasMoreSamples / getNextSampleBlock /
                                              selectBlocks / assembleSelectedBlocks (identification/data.py:148-345) with
                                              Model.getSubregressorsConditionNumbers (model.py:1054-1086), run as the loop of identifier.py:1564-1586
                                              on a seeded KUKA file -- and its IDIM-WLS pass, Identification.identifyBaseParameters with useWLS = 1
                                              (identifier.py:739-790), with and without a-priori torques
  tests/golden/ref_compute_regressors.npz     the reference's own Model.computeRegressors + simulateDynamicsIDynTree
                                              (identification/model.py:239-632) executed on small sample sets with the
                                              iDynTree calls answered by this repository's CPU oracle (a minimal object shim:
                                              state setters, regressor / inverse dynamics / frame Jacobian getters).  What
                                              this pins is the reference's PYTHON logic around those calls -- stacking order,
                                              friction column blocks, gravity-only column deletion, skipSamples, a-priori
                                              torques, simulated base wrench, contact-force bookkeeping -- not iDynTree's
                                              numerics (those stay unpinned).  Same for
                                              Model.getRandomRegressor (model.py:634-830) and the D-optimality gradient worker
                                              _dopt_gradient_worker_func (excitation/analyticalGradient.py:46-185).
"""
import json
import os
import re
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"

from flobaroid_amd.topology import parse_urdf  # noqa: E402


def main():
    robots = os.path.join(REPO, "flobaroid_amd", "robots")
    golden = os.path.join(REPO, "tests", "golden")
    os.makedirs(robots, exist_ok=True)
    os.makedirs(golden, exist_ok=True)
    for name in ["threeLinks", "kuka_lwr4", "walkman_left_arm", "walkman_apriori"]:
        t = parse_urdf(os.path.join(REF, "model", name + ".urdf"))
        t.save_json(os.path.join(robots, name + ".topology.json"))
        print(name, t.num_links, "links", t.num_dofs, "dofs")

    # --- F9: TUTORIAL table -------------------------------------------------------------
    rows = []
    pat = re.compile(r"^\|\s*(-?\d+\.\d+)\|\s*(-?\d+\.\d+)\|[^|]*\|[^|]*\|([^|]*)\|#(\d+): (\S+) - (.*)$")
    with open(os.path.join(REF, "documentation", "TUTORIAL.md")) as f:
        for line in f:
            m = pat.match(line.rstrip("\n"))
            if m:
                rows.append((int(m.group(4)), float(m.group(1)), "nID" in m.group(3), m.group(5), m.group(6)))
    rows.sort()
    assert [r[0] for r in rows] == list(range(101)), len(rows)
    with open(os.path.join(golden, "kuka_tutorial_apriori.json"), "w") as f:
        json.dump(
            {
                "source": "documentation/TUTORIAL.md:60-160 (A priori column, nID flag)",
                "xStdModel": [r[1] for r in rows],
                "non_id": [r[0] for r in rows if r[2]],
                "symbols": [r[3] for r in rows],
                "descriptions": [r[4] for r in rows],
                "apriori_mass": 16.0,
            },
            f,
            indent=0,
        )

    # --- F5: trajectory fixture ---------------------------------------------------------
    z = np.load(os.path.join(REF, "model", "kuka_lwr4.urdf.trajectory_opt_1.npz"), allow_pickle=True)
    np.savez_compressed(
        os.path.join(golden, "kuka_trajectory_opt_1.npz"),
        positions=z["positions"],
        velocities=z["velocities"],
        accelerations=z["accelerations"],
        times=z["times"],
        frequency=z["frequency"],
        n_observable_base_params=z["n_observable_base_params"],
    )

    # --- reference-held DOF serialisations -----------------------------------------------
    import xml.etree.ElementTree as ET

    orders = {"source": "model/<robot>_regressor.xml <joint> lists (comments skipped); configs/walkman_static.yaml:60-64"}
    for robot, xml in [("threeLinks", "threeLinks_regressor.xml"), ("kuka_lwr4", "kuka_lwr4_regressor.xml"),
                       ("walkman_left_arm", "walkman_left_arm_regressor.xml"), ("walkman_apriori", "walkman_regressor.xml")]:
        tree = ET.parse(os.path.join(REF, "model", xml)).getroot()
        orders[robot] = [(e.text or "").strip() for e in tree.iter() if e.tag == "joint"]
    with open(os.path.join(REF, "configs", "walkman_static.yaml")) as f:
        lines = f.read().split("\n")[59:64]
    orders["walkman_static_yaml"] = re.findall(r"'([A-Za-z0-9_]+)'", " ".join(lines))
    assert orders["walkman_static_yaml"] == orders["walkman_apriori"] and len(orders["walkman_apriori"]) == 29
    with open(os.path.join(golden, "reference_joint_orders.json"), "w") as f:
        json.dump(orders, f, indent=1)

    # --- documented structure counts ----------------------------------------------------
    with open(os.path.join(golden, "structure.json"), "w") as f:
        json.dump(
            {
                "source": "documentation/analysis_findings.md:29,43,45; design_notes.md:98-104; "
                "model/kuka_lwr4.urdf.trajectory_opt_1.npz; SURVEY.md Appendix A probe table",
                "threeLinks": {"links": 3, "dofs": 2, "base_rank_floating": 24},
                "kuka_lwr4": {"links": 8, "dofs": 7, "base_rank_fixed": 43, "base_rank_fixed_friction": 64,
                              "mass": 16.0},
                "walkman_left_arm": {"links": 9, "dofs": 7, "base_rank_floating": 59},
                "walkman_apriori": {"links": 48, "dofs": 29, "base_rank_floating": 213, "mass": 128.0,
                                    "free_links": 30},
            },
            f,
            indent=1,
        )


def reference_host_functions(golden):
    """Golden vectors from the reference's own host functions (see the module docstring)."""
    import tempfile
    import types

    class _Blank:
        def __getattr__(self, k):
            return ""

    for name in ["idyntree", "idyntree.bindings", "colorama", "trimesh"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["colorama"].Fore = sys.modules["colorama"].Back = sys.modules["colorama"].Style = _Blank()
    sys.modules["idyntree"].bindings = sys.modules["idyntree.bindings"]
    sys.path.insert(0, REF)
    import identification.data as rdata
    import identification.helpers as rhelpers

    out = {}
    rng = np.random.default_rng(2024)
    # ---- Data.preprocess, joint channels + FT
    S, n, Fs = 600, 3, 200.0
    T = np.arange(S) / Fs
    T[300:] += 0.0004  # one irregular step
    Q = np.cumsum(rng.standard_normal((S, n)) * 0.01, axis=0) + np.sin(T[:, None] * (1 + np.arange(n)))
    V = rng.standard_normal((S, n))
    Tau = 5 * np.sin(T[:, None] * 3) + 0.3 * rng.standard_normal((S, n))
    Tau[rng.integers(0, S, 12), rng.integers(0, n, 12)] += 8.0  # outliers for the median filter
    FT = [rng.standard_normal((S, 6)), rng.standard_normal((S, 6))]
    opt = {"filterMedianSize": 5, "useDeg": 0, "num_dofs": n, "filterLowPass1": [8.0, 5], "filterLowPass2": [6.0, 5],
           "filterLowPass3": [3.0, 4], "waitForZeroAcc": 0, "zeroAccThresh": 0.1}
    out.update(pre_Q=Q.copy(), pre_V=V.copy(), pre_Tau=Tau.copy(), pre_T=T.copy(), pre_Fs=Fs, pre_FT0=FT[0].copy(), pre_FT1=FT[1].copy(),
               pre_opt=json.dumps(opt))
    d = rdata.Data(opt)
    Vdot = np.zeros_like(Q)
    Qr, Vr, Tr = np.zeros_like(Q), np.zeros_like(Q), np.zeros_like(Q)
    d.preprocess(Q, V, Vdot, Tau, T, Fs, Q_raw=Qr, V_raw=Vr, Tau_raw=Tr, FT=FT)
    out.update(pre_out_Q=Q, pre_out_V=V, pre_out_Vdot=Vdot, pre_out_Tau=Tau, pre_out_Q_raw=Qr, pre_out_V_raw=Vr, pre_out_Tau_raw=Tr,
               pre_out_FT0=FT[0], pre_out_FT1=FT[1])
    # degrees variant (short)
    optd = dict(opt, useDeg=1)
    Qd, Vd, Td = np.rad2deg(out["pre_Q"][:200]).copy(), out["pre_V"][:200].copy(), out["pre_Tau"][:200].copy()
    Vdd = np.zeros_like(Qd)
    rdata.Data(optd).preprocess(Qd, Vd, Vdd, Td, T[:200].copy(), Fs)
    out.update(pre_deg_out_Q=Qd, pre_deg_out_V=Vd, pre_deg_out_Vdot=Vdd)
    # ---- friction sign helpers
    vel = np.sin(T[:, None] * (2 + np.arange(n))) * 0.5
    raw = vel + 0.05 * rng.standard_normal(vel.shape)
    out.update(fs_vel=vel, fs_raw=raw, fs_freq=Fs)
    for tag, samples, o in [("a", {"velocities": vel.copy(), "velocities_raw": raw.copy(), "frequency": Fs}, {"frictionVelocityCutoff": 25.0, "frictionSignThreshold": 0.02}),
                            ("b", {"velocities": vel.copy(), "velocities_raw": raw.copy(), "frequency": Fs}, {"frictionVelocityCutoff": 150.0}),
                            ("c", {"velocities": vel.copy()}, {"frictionSignThreshold": 0.05})]:
        out["fs_%s_velocities" % tag] = rhelpers.getFrictionSignVelocities(samples, o).copy()
        out["fs_%s_series" % tag] = rhelpers.getFrictionSignSeries(samples, o).copy()
    # ---- Data.init_from_files on three small files
    tmp = tempfile.mkdtemp()
    files = []
    for i, S_i in enumerate((40, 25, 33)):
        fn = os.path.join(tmp, "m%d.npz" % i)
        tt = 0.01 * np.arange(S_i) + 3.0 * i
        np.savez(fn, positions=rng.standard_normal((S_i, n)), velocities=rng.standard_normal((S_i, n)), accelerations=rng.standard_normal((S_i, n)),
                 torques=rng.standard_normal((S_i, n)), times=tt, frequency=100.0, target_positions=rng.standard_normal((S_i, n)))
        files.append(fn)
        z = np.load(fn)
        for k in z.files:
            out["iff_in%d_%s" % (i, k)] = z[k]
    optf = {"startOffset": 4, "skipSamples": 1, "verbose": 0, "showTiming": 0, "selectBlocksFromMeasurements": 0}
    d = rdata.Data(optf)
    d.init_from_files([[files[0], files[1]], [files[2]]])
    for k, v in d.measurements.items():
        out["iff_out_" + k] = np.asarray(v)
    out.update(iff_opt=json.dumps(optf), iff_num_loaded=d.num_loaded_samples, iff_num_used=d.num_used_samples,
               iff_file_boundaries=np.array(d.file_boundaries))
    np.savez_compressed(os.path.join(golden, "ref_host_functions.npz"), **out)
    print("ref_host_functions.npz:", len(out), "arrays")


def _import_reference(mod):
    """import a reference module, replacing any third-party module the image lacks by a permissive placeholder"""
    import importlib
    import types

    class _Blank:
        def __getattr__(self, k):
            return ""

        def __call__(self, *a, **k):
            return _Blank()

    class _Mod(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Blank()

    if REF not in sys.path:
        sys.path.insert(0, REF)
    # the console/plot output module of the reference builds colour tables at import time: not on any executed path here
    sys.modules.setdefault("identification.output", _Mod("identification.output"))
    for _ in range(40):
        try:
            return importlib.import_module(mod)
        except ModuleNotFoundError as e:
            missing = e.name
            if missing is None or missing.startswith("identification"):
                raise
            parts = missing.split(".")
            for i in range(1, len(parts) + 1):
                nm = ".".join(parts[:i])
                if nm not in sys.modules:
                    sys.modules[nm] = _Mod(nm)
            print("  (placeholder for missing module %s)" % missing)
    raise RuntimeError("could not import " + mod)


def reference_estimators(golden):
    """Golden vectors of the reference's estimator code (see the module docstring)."""
    from types import SimpleNamespace as NS

    sys.path.insert(0, os.path.join(REPO, "tests"))
    from common import load_topo, random_states
    from oracle.oracle import OracleModel

    rmodel = _import_reference("identification.model")
    rident = _import_reference("identifier")
    out = {}

    def lin_deps(tag, topo_name, floating, fric, S, seed, minTol, basis=0, orth=1):
        t = load_topo(topo_name)
        rng = np.random.default_rng(seed)
        st = random_states(t, S, rng, floating, use_limits=True)
        om = OracleModel(t, floating=bool(floating), fric=bool(fric), fric_sym=True)
        sign = np.tanh(st["dq"] / 0.02)
        Y = om.regressor(st, sign if fric else None)
        L, n = t.num_links, t.num_dofs
        nall = 10 * L + (3 * n if fric else 0)
        fm = NS(opt={"minTol": minTol, "useBasisProjection": basis, "orthogonalizeBasis": orth, "identifyGravityParamsOnly": 0,
                     "identifyFrictionSimultaneously": int(fric), "identifySymmetricVelFriction": 1, "stribeckVelocity": 0, "randomSamples": 0},
                num_dofs=n, num_links=L, num_model_params=10 * L, num_all_params=nall, num_identified_params=nall)
        rmodel.Model.computeRegressorLinDepsQR(fm, regressor=Y)
        for k in ("q", "dq", "ddq", "base_vel", "base_acc", "rpy"):
            if k in st:
                out["%s_st_%s" % (tag, k)] = st[k]
        out.update({tag + "_meta": json.dumps({"robot": topo_name, "floating": int(floating), "friction": int(fric), "S": S, "minTol": minTol}),
                    tag + "_P": np.asarray(fm.P), tag + "_num_base_params": fm.num_base_params, tag + "_independent_cols": np.asarray(fm.independent_cols),
                    tag + "_linear_deps": fm.linear_deps, tag + "_K": fm.K, tag + "_Pb": fm.Pb, tag + "_Rdiag": np.diag(fm.R),
                    tag + "_identified_params": np.asarray(fm.identified_params), tag + "_non_id": np.asarray(fm.non_id, dtype=np.int64),
                    tag + "_identifiable": np.asarray(fm.identifiable, dtype=np.int64)})
        if basis:
            out.update({tag + "_B": fm.B, tag + "_Binv": fm.Binv})
        return t, st, Y, fm, rng

    lin_deps("ldA", "kuka_lwr4", 0, 0, 60, 11, 1e-8)
    lin_deps("ldB", "kuka_lwr4", 0, 1, 60, 12, 1e-8)
    lin_deps("ldD", "kuka_lwr4", 0, 0, 60, 14, 1e-8, basis=1, orth=1)   # useBasisProjection (model.py:896-929), orthogonalised
    lin_deps("ldE", "threeLinks", 1, 0, 50, 15, 1e-8, basis=1, orth=0)  # and with the pseudo-inverse
    t, st, Y, fm, rng = lin_deps("ldC", "threeLinks", 1, 0, 50, 13, 1e-8)

    # ---- Identification methods on the threeLinks floating problem (two "files", contact forces, a-priori vector)
    S = 50
    rows = Y.shape[0] // S
    x_true = t.x_std() * (1.0 + 0.1 * rng.standard_normal(30))
    xStdModel = t.x_std()
    cf = 0.05 * rng.standard_normal(Y.shape[0])
    torques = Y @ x_true + cf + 0.01 * rng.standard_normal(Y.shape[0])
    torquesAP = Y @ xStdModel
    fm.YStd = Y
    fm.YBase = Y @ fm.Pb
    fm.xStdModel = xStdModel
    fm.torques_stack = torques
    fm.torquesAP_stack = torquesAP
    fm.tau = torques - torquesAP
    fm.tauMeasured = torques.reshape(S, rows)
    fm.contactForcesSum = cf
    opt = {"useBasisProjection": 0, "addContacts": 1, "showBaseParams": 0, "verbose": 0, "useRegressorRegularization": 0, "useWLS": 0,
           "useAPriori": 1, "floatingBase": 1, "useTrajectoryWeighting": 1, "skipSamples": 0, "showTiming": 0}
    idf = NS(opt=opt, model=fm, data=NS(num_used_samples=S, file_boundaries=[0, 20, 50]), urdf_file_real=None)
    rident.Identification.identifyBaseParameters(idf)
    out.update(id_x_true=x_true, id_cf=cf, id_torques=torques, id_xBaseModel=fm.xBaseModel, id_xBase=fm.xBase.copy())
    idf.tauEstimated = (fm.YBase @ fm.xBase + torquesAP + cf).reshape(S, rows)
    out["id_p_sigma_x"] = rident.Identification.getStdDevForParams(idf)
    rident.Identification.findStdFromBaseParameters(idf)
    out["id_xStd_from_base"] = fm.xStd.copy()
    rident.Identification.identifyStandardParametersDirect(idf)
    out["id_xStd_direct"] = fm.xStd.copy()
    xe = np.zeros(30)
    ess = np.array([0, 1, 4, 10, 11, 13, 20, 22, 25])
    xe[ess] = x_true[ess]
    idf.xStdEssential = xe
    idf.num_essential_params = 6
    idf.stdEssentialIdx = ess
    rident.Identification.identifyStandardEssentialParameters(idf)
    out.update(id_xStdEssential=xe, id_num_essential=6, id_xStd_essential=fm.xStd.copy())
    YB_bw, tau_bw = rident.Identification._extractBaseWrenchRows(idf)
    out.update(id_bw_YBase=YB_bw, id_bw_tau=tau_bw, id_bw_cf=idf._bw_contactForcesSum, id_file_boundaries=np.array([0, 20, 50]))
    np.savez_compressed(os.path.join(golden, "ref_estimators.npz"), **out)
    print("ref_estimators.npz:", len(out), "arrays")


def reference_compute_regressors(golden):
    """Golden vectors of the reference's computeRegressors logic (see the module docstring)."""
    from types import SimpleNamespace as NS

    sys.path.insert(0, os.path.join(REPO, "tests"))
    from common import load_topo, random_states
    from oracle.oracle import OracleModel

    rmodel = _import_reference("identification.model")

    class Arr:
        def __init__(self, n=6):
            self.a = np.zeros(int(n))

        def setVal(self, i, v):
            self.a[i] = v

        def getVal(self, i):
            return self.a[i]

        def toNumPy(self):
            return self.a

    class Mat:
        def __init__(self, *shape):
            self.m = np.zeros(shape) if shape else None

        def toNumPy(self):
            return self.m

    class Gen:
        def __init__(self, model=None):
            self.v = None

        def jointTorques(self):
            return NS(toNumPy=lambda: self.v[-self.n:].copy())

        def baseWrench(self):
            return NS(toNumPy=lambda: self.v[:6].copy())

    class KinDyn:
        """answers the iDynTree KinDynComputations calls of model.py with the CPU oracle"""

        def __init__(self, topo, floating):
            self.t, self.fl = topo, bool(floating)
            self.om = OracleModel(topo, floating=self.fl)
            self.n = topo.num_dofs

        def setRobotState(self, *a):
            if len(a) == 3:
                self.st = {"q": a[0].a.copy()[None], "dq": a[1].a.copy()[None]}
            else:
                self.st = {"q": a[1].a.copy()[None], "dq": a[3].a.copy()[None], "base_vel": np.asarray(a[2].v, float)[None], "rpy": np.asarray(a[0].rpy, float)[None]}
            return True

        def _full(self, bacc, ddq):
            st = dict(self.st, ddq=ddq.a.copy()[None])
            if self.fl:
                st["base_acc"] = bacc.a.copy()[None]
            return st

        def inverseDynamicsInertialParametersRegressor(self, bacc, ddq, out):
            Y = self.om.regressor(self._full(bacc, ddq))
            out.m = Y if self.fl else np.vstack([np.zeros((6, Y.shape[1])), Y])
            return True

        def inverseDynamics(self, bacc, ddq, ext, gen):
            tau = self.om.inverse_dynamics(self._full(bacc, ddq), self.xstd)[0]
            gen.v = tau if self.fl else np.concatenate([np.zeros(6), tau])
            gen.n = self.n
            return True

        def getFrameFreeFloatingJacobian(self, frame, out):
            st = dict(self.st, ddq=np.zeros((1, self.n)))
            if self.fl:
                st["base_acc"] = np.zeros((1, 6))
            JT = np.column_stack([self.om.contact_torques(st, frame, np.eye(6)[i][None])[0] for i in range(6)])
            out.m = (JT if self.fl else np.vstack([np.zeros((6, 6)), JT])).T
            return True

    shim = NS(JointPosDoubleArray=Arr, JointDOFsDoubleArray=Arr, Vector6=Arr, VectorDynSize=Arr, MatrixDynSize=Mat,
              Rotation=NS(RPY=lambda r, p, y: NS(rpy=(r, p, y))), Position=NS(Zero=lambda: NS()),
              Transform=lambda rot, pos: NS(inverse=lambda: NS(rpy=rot.rpy)), Twist=NS(FromPython=lambda v: NS(v=np.asarray(v, float))),
              LinkWrenches=lambda model: NS(), FreeFloatingGeneralizedTorques=Gen)
    rmodel.iDynTree = shim
    out = out_main = {}

    def run(tag, robot, floating, S, seed, opt_over, contacts=None, torques_with_base=False, dst=None):
        out = out_main if dst is None else dst
        t = load_topo(robot)
        n, L = t.num_dofs, t.num_links
        rng = np.random.default_rng(seed)
        st = random_states(t, S, rng, floating, use_limits=True)
        opt = {"floatingBase": int(floating), "skipSamples": 0, "identifyGravityParamsOnly": 0, "simulateTorques": 0, "useAPriori": 0,
               "useRegressorForSimulation": 0, "identifyFrictionSimultaneously": 0, "identifySymmetricVelFriction": 1, "stribeckVelocity": 0,
               "addContacts": 1, "useStructuralRegressor": 1, "useBasisProjection": 0, "filterRegressor": 0, "showTiming": 0, "verbose": 0,
               "frictionVelocityCutoff": 25.0, "frictionSignThreshold": 0.02}
        opt.update(opt_over)
        fric = opt["identifyFrictionSimultaneously"]
        grav = opt["identifyGravityParamsOnly"]
        nfr = 0
        if fric:
            nfr = n if grav else (3 * n if opt["identifySymmetricVelFriction"] else 4 * n) + (n if opt["stribeckVelocity"] > 0 else 0)
        nall = 10 * L + nfr
        identified = []
        for i in range(L):
            identified += [10 * i + j for j in (range(4) if grav else range(10))]
        identified += list(range(10 * L, 10 * L + nfr))
        inertia_params = [10 * i + j for i in range(L) for j in range(4, 10)]
        xStdModel = np.concatenate([t.x_std(), 0.2 + rng.random(nfr)])
        rows = n + (6 if floating else 0)
        samples = {"positions": st["q"].copy(), "velocities": st["dq"].copy(), "accelerations": st["ddq"].copy(),
                   "torques": rng.standard_normal((S, rows if torques_with_base else n)), "times": 0.01 * np.arange(S), "frequency": np.array(100.0),
                   "velocities_raw": st["dq"] + 0.01 * rng.standard_normal((S, n))}
        if floating:
            samples.update(base_velocity=st["base_vel"].copy(), base_acceleration=st["base_acc"].copy(), base_rpy=st["rpy"].copy())
        if contacts:
            samples["contacts"] = np.array({f: rng.standard_normal((S, 6)) for f in contacts})
        for k, v in samples.items():
            if k == "contacts":
                for f, a in v.item(0).items():
                    out["%s_in_contacts_%s" % (tag, f)] = a.copy()
            else:
                out["%s_in_%s" % (tag, k)] = np.array(v).copy()
        kd = KinDyn(t, floating)
        kd.xstd = xStdModel[: 10 * L]
        used = S // (opt["skipSamples"] + 1)
        nb = 5
        fm = NS(opt=opt, num_dofs=n, num_links=L, num_model_params=10 * L, num_all_params=nall, num_identified_params=len(identified),
                identified_params=identified, inertia_params=inertia_params, xStdModel=xStdModel, friction_params_start=10 * L, kinDyn=kd,
                gravity_vec=None, idyn_model=None, progress=lambda it: it, Pb=np.eye(len(identified))[:, :nb], num_base_inertial_params=nb - 1)
        fm.simulateDynamicsIDynTree = lambda samples_, idx, kinDyn=None, xStdModel=None: rmodel.Model.simulateDynamicsIDynTree(fm, samples_, idx, kinDyn, xStdModel)
        data = NS(samples=samples, num_used_samples=used)
        rmodel.Model.computeRegressors(fm, data)
        out[tag + "_meta"] = json.dumps({"robot": robot, "floating": int(floating), "S": S, "opt": opt, "contacts": contacts or [], "nb": nb})
        out[tag + "_xStdModel"] = xStdModel
        for k in ("YStd", "YBase", "torques_stack", "torquesAP_stack", "tau", "contacts_stack", "contactForcesSum", "tauMeasured", "T", "sim_torq_stack"):
            out["%s_out_%s" % (tag, k)] = np.asarray(getattr(fm, k))
        out[tag + "_out_samples_torques"] = np.asarray(fm.data.samples["torques"])

    run("crA", "kuka_lwr4", 0, 20, 21, {"identifyFrictionSimultaneously": 1, "stribeckVelocity": 0.05, "skipSamples": 1, "useAPriori": 1})
    run("crB", "threeLinks", 1, 10, 22, {}, contacts=["link3"])
    run("crC", "kuka_lwr4", 0, 16, 23, {"identifyGravityParamsOnly": 1, "identifyFrictionSimultaneously": 1})
    run("crF", "kuka_lwr4", 0, 48, 25, {"filterRegressor": 1, "filterRegCutoff": 10.0})  # model.py:608-615 (5th-order Butterworth filtfilt)
    run("crD", "threeLinks", 1, 16, 24, {"simulateTorques": 1, "useAPriori": 1, "identifyFrictionSimultaneously": 1, "identifySymmetricVelFriction": 0},
        contacts=["link2", "link3"], torques_with_base=True)
    # ---- getRandomRegressor (model.py:634-830): global-RNG call order, raw Gram, pivoted QR, cache file keys
    import tempfile

    def run_random(tag, robot, floating, n_samples, seed, opt_over, dst=None, triu_only=False):
        out = out_main if dst is None else dst
        t = load_topo(robot)
        n, L = t.num_dofs, t.num_links
        opt = {"floatingBase": int(floating), "identifyGravityParamsOnly": 0, "identifyFrictionSimultaneously": 0, "identifySymmetricVelFriction": 1,
               "stribeckVelocity": 0, "verbose": 0, "frictionSignThreshold": 0.02}
        opt.update(opt_over)
        nfr = 0
        if opt["identifyFrictionSimultaneously"]:
            nfr = 3 * n if opt["identifySymmetricVelFriction"] else 4 * n
        kd = KinDyn(t, floating)
        tmpd = tempfile.mkdtemp()
        fm = NS(opt=opt, num_dofs=n, num_links=L, num_model_params=10 * L, num_identified_params=10 * L + nfr, N_OUT=n + (6 if floating else 0),
                inertia_params=[], limits={k: dict(v) for k, v in t.limits.items()}, jointNames=list(t.dof_names), kinDyn=kd, gravity_vec=None,
                progress=lambda it: it, urdf_file=os.path.join(tmpd, robot + ".urdf"))
        np.random.seed(seed)
        R, Q, RQ, PQ = rmodel.Model.getRandomRegressor(fm, n_samples=n_samples)
        cache = np.load(fm.urdf_file + ".regressor.npz")
        if triu_only:  # symmetric: the upper triangle is the matrix
            out[tag + "_R_triu"] = R[np.triu_indices(R.shape[0])]
        else:
            out[tag + "_R"] = R
        out.update({tag + "_meta": json.dumps({"robot": robot, "floating": int(floating), "n_samples": n_samples, "seed": seed, "opt": opt}),
                    tag + "_PQ": PQ, tag + "_RQdiag": np.diag(RQ), tag + "_cache_keys": np.array(sorted(cache.files)),
                    tag + "_cache_n": cache["n"], tag + "_cache_fb": cache["fb"], tag + "_cache_fric": cache["fric"]})
        return fm, (R, Q, RQ, PQ)

    # ---- the D-optimality gradient worker (excitation/analyticalGradient.py:46-185), same shim
    rgrad = _import_reference("excitation.analyticalGradient")
    rgrad.iDynTree = NS(JointPosDoubleArray=Arr, JointDOFsDoubleArray=Arr, Vector6=Arr, MatrixDynSize=Mat,
                        Transform=NS(Identity=lambda: NS(rpy=(0.0, 0.0, 0.0))), Twist=NS(Zero=lambda: NS(v=np.zeros(6))))

    def run_gradient(tag, robot, floating, S, seed, eps):
        t = load_topo(robot)
        n, L = t.num_dofs, t.num_links
        rng = np.random.default_rng(seed)
        st = random_states(t, S, rng, 0, use_limits=True)
        kd = KinDyn(t, floating)
        rgrad._worker_state["model"] = NS(num_dofs=n, kinDyn=kd, gravity_vec=None)
        nout = n + (6 if floating else 0)
        W = rng.standard_normal((S * nout, 10 * L))
        W_visc = rng.standard_normal(S * nout)
        args = (np.arange(S), st["q"], st["dq"], st["ddq"], W, np.zeros(10 * L), nout, eps, bool(floating), 0 if floating else 6, True, W_visc,
                np.zeros(n), 6 if floating else 0, -np.ones(n, dtype=int))
        sq, sdq, sddq, _ = rgrad._dopt_gradient_worker_func(args)
        out.update({tag + "_meta": json.dumps({"robot": robot, "floating": int(floating), "S": S, "eps": eps}), tag + "_q": st["q"], tag + "_dq": st["dq"],
                    tag + "_ddq": st["ddq"], tag + "_W": W, tag + "_W_visc": W_visc, tag + "_sens_q": sq, tag + "_sens_dq": sdq, tag + "_sens_ddq": sddq})

    run_gradient("gwA", "kuka_lwr4", 0, 6, 31, 1e-6)
    run_gradient("gwB", "threeLinks", 1, 7, 32, 1e-6)
    run_random("rrA", "kuka_lwr4", 0, 40, 7, {"identifyFrictionSimultaneously": 1})
    run_random("rrB", "threeLinks", 1, 30, 8, {})
    np.savez_compressed(os.path.join(golden, "ref_compute_regressors.npz"), **out)
    print("ref_compute_regressors.npz:", len(out), "arrays")

    # ---- WALK-MAN (48 links, 29 DOF, floating base), the robot of BASELINE configs[3..4], through the same reference code:
    #      computeRegressors with the option set of configs/walkman_full.yaml that matters on this path (floating base, a-priori
    #      torques, simultaneous friction, contacts on the two foot FT frames, joint-only torque measurements -> simulated base wrench),
    #      getRandomRegressor with randomSamples = 10000 and computeRegressorLinDepsQR with minTol = 0.005 (walkman_full.yaml:243-247)
    outW = {}
    run("crW", "walkman_apriori", 1, 14, 41, {"useAPriori": 1, "identifyFrictionSimultaneously": 1, "skipSamples": 1}, contacts=["l_leg_ft", "r_leg_ft"], dst=outW)
    fmr, (R, Q, RQ, PQ) = run_random("rrW", "walkman_apriori", 1, 10000, 42, {}, dst=outW, triu_only=True)
    t = load_topo("walkman_apriori")
    fm = NS(opt={"minTol": 0.005, "useBasisProjection": 0, "orthogonalizeBasis": 1, "identifyGravityParamsOnly": 0, "identifyFrictionSimultaneously": 0,
                 "identifySymmetricVelFriction": 1, "stribeckVelocity": 0, "randomSamples": 10000},
            num_dofs=t.num_dofs, num_links=t.num_links, num_model_params=480, num_all_params=480, num_identified_params=480,
            getRandomRegressor=lambda n_samples=None: (R, Q, RQ, PQ))
    rmodel.Model.computeRegressorLinDepsQR(fm)
    outW.update(ldW_P=np.asarray(fm.P), ldW_num_base_params=fm.num_base_params, ldW_independent_cols=np.asarray(fm.independent_cols),
                ldW_K=fm.K, ldW_Rdiag=np.diag(fm.R), ldW_non_id=np.asarray(fm.non_id, dtype=np.int64),
                ldW_identifiable=np.asarray(fm.identifiable, dtype=np.int64), ldW_minTol=0.005)
    # ---- SDP._observabilityWeights (sdp.py:295-315) on R1 K of a seeded WALK-MAN least-squares problem (R1 from numpy.linalg.qr of
    #      the oracle's YBase, as sdp.py:470-487 builds it)
    rsdp = _import_reference("identification.sdp")
    rng = np.random.default_rng(43)
    stw = random_states(t, 400, rng, 1, use_limits=True)
    Yw = OracleModel(t, floating=True).regressor(stw)
    R1 = np.linalg.qr(Yw[:, fm.independent_cols], mode="r")
    R1_K = R1 @ fm.K
    outW.update(owW_R1_K=R1_K, owW_weights=rsdp.SDP._observabilityWeights(None, R1_K), owW_seed=43, owW_S=400)
    np.savez_compressed(os.path.join(golden, "ref_walkman.npz"), **outW)
    print("ref_walkman.npz:", len(outW), "arrays, rank", fm.num_base_params)


def reference_blocks_and_wls(golden):
    """Golden vectors of the reference's block selection and WLS pass (see the module docstring)."""
    import tempfile
    import types
    from types import SimpleNamespace as NS

    sys.path.insert(0, os.path.join(REPO, "tests"))
    from common import load_topo, random_states
    from oracle.oracle import OracleModel

    rmodel = _import_reference("identification.model")
    rdata = _import_reference("identification.data")
    rident = _import_reference("identifier")
    out = {}

    # ---- block selection on KUKA (fixed base): 640 samples = 6 blocks of 100 + one of 40; block 3 repeats block 1
    t = load_topo("kuka_lwr4")
    n, L = t.num_dofs, t.num_links
    om = OracleModel(t)
    rng = np.random.default_rng(61)
    Ys = om.regressor(random_states(t, 300, rng, 0, use_limits=True))
    fm = NS(opt={"minTol": 1e-4, "useBasisProjection": 0, "orthogonalizeBasis": 1, "identifyGravityParamsOnly": 0, "identifyFrictionSimultaneously": 0,
                 "identifySymmetricVelFriction": 1, "stribeckVelocity": 0, "randomSamples": 0, "verbose": 0},
            num_dofs=n, num_links=L, num_model_params=10 * L, num_all_params=10 * L, num_identified_params=10 * L)
    rmodel.Model.computeRegressorLinDepsQR(fm, regressor=Ys)
    fm.getSubregressorsConditionNumbers = types.MethodType(rmodel.Model.getSubregressorsConditionNumbers, fm)
    S, bs, skip = 640, 100, 1
    st = random_states(t, S, rng, 0, use_limits=True)
    scale = np.ones(S)
    for b, f in enumerate((1.0, 0.5, 0.12, 0.5, 1.0, 0.3, 0.08)):  # differently exciting blocks -> a spread of condition numbers
        scale[b * bs:(b + 1) * bs] = f
    for k in ("dq", "ddq"):
        st[k] = st[k] * scale[:, None]
    for k in ("q", "dq", "ddq"):
        st[k][3 * bs:4 * bs] = st[k][1 * bs:2 * bs]
    meas = {"positions": st["q"], "velocities": st["dq"], "accelerations": st["ddq"], "torques": rng.standard_normal((S, n)),
            "times": 0.005 * np.arange(S), "frequency": np.array(200.0), "aux1d": rng.standard_normal(S)}
    tmp = tempfile.mkdtemp()
    fn = os.path.join(tmp, "blocks.npz")
    np.savez(fn, **meas)
    opt = {"startOffset": 0, "skipSamples": skip, "verbose": 0, "showTiming": 0, "selectBlocksFromMeasurements": 1, "blockSize": bs,
           "selectBestPerenctage": 70}
    d = rdata.Data(opt)
    d.init_from_files([[fn]])
    ic = np.asarray(fm.independent_cols)
    while True:
        used = d.num_used_samples
        idx = np.arange(used) * (skip + 1)
        sb = {"q": d.samples["positions"][idx], "dq": d.samples["velocities"][idx], "ddq": d.samples["accelerations"][idx]}
        fm.YBase = om.regressor(sb)[:, ic]
        d.getBlockStats(fm)
        if d.hasMoreSamples():
            d.getNextSampleBlock()
        else:
            break
    seen = list(d.seenBlocks)
    d.selectBlocks()
    d.assembleSelectedBlocks()
    for k, v in meas.items():
        out["bl_in_" + k] = v
    out.update(bl_opt=json.dumps({"startOffset": 0, "skipSamples": skip, "selectBlocksFromMeasurements": 1, "blockSize": bs, "selectBestPerenctage": 70}),
               bl_independent_cols=ic, bl_K=fm.K, bl_num_base_params=fm.num_base_params,
               bl_seen_pos=np.array([b[0] for b in seen]), bl_seen_size=np.array([b[1] for b in seen]), bl_seen_cond=np.array([b[2] for b in seen]),
               bl_seen_linkconds=np.array([b[3] for b in seen]), bl_used_pos=np.array([b[0] for b in d.usedBlocks]),
               bl_unused_pos=np.array([b[0] for b in d.unusedBlocks]), bl_num_selected=d.num_selected_samples, bl_num_used=d.num_used_samples)
    for k, v in d.samples.items():
        out["bl_out_" + k] = np.asarray(v)
    print("blocks: seen", [(b[0], b[1], round(b[2], 1)) for b in seen], "used", [b[0] for b in d.usedBlocks], "unused", [b[0] for b in d.unusedBlocks])

    # ---- IDIM-WLS (identifier.py:739-790) on the threeLinks floating problem, without and with a-priori torques
    t = load_topo("threeLinks")
    S = 50
    for tag, useAP, seed in (("wlsA", 0, 71), ("wlsB", 1, 72)):
        rng = np.random.default_rng(seed)
        st = random_states(t, S, rng, 1, use_limits=True)
        Y = OracleModel(t, floating=True).regressor(st)
        rows = Y.shape[0] // S
        fmw = NS(opt={"minTol": 1e-8, "useBasisProjection": 0, "orthogonalizeBasis": 1, "identifyGravityParamsOnly": 0, "identifyFrictionSimultaneously": 0,
                      "identifySymmetricVelFriction": 1, "stribeckVelocity": 0, "randomSamples": 0}, num_dofs=t.num_dofs, num_links=t.num_links,
                 num_model_params=30, num_all_params=30, num_identified_params=30)
        rmodel.Model.computeRegressorLinDepsQR(fmw, regressor=Y)
        x_true = t.x_std() * (1.0 + 0.1 * rng.standard_normal(30))
        cf = 0.05 * rng.standard_normal(Y.shape[0])
        torques = Y @ x_true + cf + 0.01 * rng.standard_normal(Y.shape[0])
        torquesAP = Y @ t.x_std()
        fmw.YStd, fmw.YBase, fmw.xStdModel = Y, Y @ fmw.Pb, t.x_std()
        fmw.torques_stack, fmw.torquesAP_stack = torques, torquesAP
        fmw.tau = torques - torquesAP if useAP else torques.copy()
        fmw.tauMeasured = torques.reshape(S, rows)
        fmw.contactForcesSum = cf
        o = {"useBasisProjection": 0, "addContacts": 1, "showBaseParams": 0, "verbose": 0, "useRegressorRegularization": 0, "useWLS": 1,
             "useAPriori": useAP, "floatingBase": 1, "skipSamples": 0, "showTiming": 0, "estimateWith": "base", "identifyFrictionSimultaneously": 1,
             "showErrorHistogram": 0}
        idf = NS(opt=o, model=fmw, data=NS(num_used_samples=S), urdf_file_real=None)
        idf.estimateRegressorTorques = types.MethodType(rident.Identification.estimateRegressorTorques, idf)
        idf.getStdDevForParams = types.MethodType(rident.Identification.getStdDevForParams, idf)
        idf.identifyBaseParameters = types.MethodType(rident.Identification.identifyBaseParameters, idf)
        tau_in = fmw.tau.copy()
        idf.identifyBaseParameters()
        for k in ("q", "dq", "ddq", "base_vel", "base_acc", "rpy"):
            out["%s_st_%s" % (tag, k)] = st[k]
        out.update({tag + "_meta": json.dumps({"robot": "threeLinks", "S": S, "useAPriori": useAP}), tag + "_independent_cols": np.asarray(fmw.independent_cols),
                    tag + "_tau": tau_in, tag + "_cf": cf, tag + "_torques": torques, tag + "_torquesAP": torquesAP, tag + "_p_sigma_x": idf.p_sigma_x,
                    tag + "_xBase": fmw.xBase.copy(), tag + "_model_tau_after": np.asarray(fmw.tau).reshape(-1)})
    np.savez_compressed(os.path.join(golden, "ref_blocks_wls.npz"), **out)
    print("ref_blocks_wls.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 4: the reference's OWN Identification class (identifier.py, unchanged) driven through the floating-base estimator chain
# ---------------------------------------------------------------------------------------------------------------------------------
FB_SCENARIOS = {
    # configs/threeLinks.yaml as shipped (floating base, data-driven pivoted QR: useStructuralRegressor 0), OLS variant: the SDP needs
    # cvxpy, which the image lacks -> constrainToConsistent 0 (identifier.py:944-954)
    "threelinks": dict(robot="threeLinks", yaml="threeLinks.yaml", files=[(2000, 0.05, 42)], contacts=[],
                       over=dict(constrainToConsistent=0, createPlots=0, verbose=0, showTiming=0, showBaseParams=0, showStandardParams=0,
                                 showBaseEqns=0)),
    # configs/walkman_full.yaml as shipped -- useBaseWrenchForBaseParams 1 (Ayusawa's base-link method, identifier.py:888-892 ->
    # _extractBaseWrenchRows :617-681), useTrajectoryWeighting 1 on TWO measurement files of different noise levels, contact wrenches on
    # the two foot FT frames, postIdentifyFriction 1 -- without the SDP (constrainToConsistent 0) and its a-priori feasibility check
    "walkman": dict(robot="walkman_apriori", yaml="walkman_full.yaml", files=[(520, 0.05, 7), (610, 0.4, 8)], contacts=["l_leg_ft", "r_leg_ft"], with_base=True,
                    over=dict(constrainToConsistent=0, checkAPrioriFeasibility=0, createPlots=0, verbose=0, showTiming=0, showBaseParams=0,
                              showStandardParams=0, showBaseEqns=0, randomSamples=3000)),
}


def fb_measurement_files(name, outdir):
    """Seeded synthetic measurement files of a floating-base scenario (the generator of the reference's tests/test_identification.py:25-93
    with the base states of model.py:720-725): [(path, arrays)] -- tau = inverse dynamics of the a-priori parameters (CPU oracle) + noise,
    joint torques only (the base wrench is simulated by the path, model.py:398-413), contact wrenches on the listed frames."""
    from oracle.oracle import OracleModel
    from flobaroid_amd.topology import Topology

    sc = FB_SCENARIOS[name]
    topo = Topology.load(os.path.join(REPO, "flobaroid_amd", "robots", sc["robot"] + ".topology.json"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from common import random_states

    om = OracleModel(topo, floating=True)
    out = []
    for i, (S, noise, seed) in enumerate(sc["files"]):
        rng = np.random.default_rng(seed)
        st = random_states(topo, S, rng, 1, use_limits=True)
        tau = om.inverse_dynamics(st, topo.x_std()) + rng.normal(0, noise, (S, 6 + topo.num_dofs))
        contacts = {f: 0.02 * rng.standard_normal((S, 6)) for f in sc["contacts"]}  # (small: the weighting pre-pass does not subtract them, identifier.py:661)
        for f, wr in contacts.items():
            # the reference's convention (model.py:562-576): measured JOINT torques already contain the contact contribution, the base
            # rows receive it inside the path
            tau[:, 6:] += np.asarray(om.contact_torques(st, f, wr)).reshape(S, -1)[:, 6:]
        # (measured base wrench + joint torques where the scenario says so: the per-file noise then reaches the base rows that the
        # trajectory weighting looks at; else joint torques only and the path simulates the base wrench, model.py:398-413)
        arrs = dict(positions=st["q"], velocities=st["dq"], accelerations=st["ddq"], torques=tau if sc.get("with_base") else tau[:, 6:],
                    times=np.arange(S) / 200.0,
                    base_velocity=st["base_vel"], base_acceleration=st["base_acc"], base_rpy=st["rpy"], frequency=np.array(200.0))
        if sc["contacts"]:
            arrs["contacts"] = np.array(contacts)
        path = os.path.join(outdir, "%s_meas_%d.npz" % (name, i))
        np.savez(path, **arrs)
        out.append((path, arrs))
    return out


def fb_config(name):
    import yaml

    sc = FB_SCENARIOS[name]
    with open(os.path.join(REF, "configs", sc["yaml"])) as f:
        c = yaml.load(f, Loader=yaml.SafeLoader)
    c.update(sc["over"])
    return c


def run_reference_identification_fb(name, workdir):
    """identifier.py's Identification, not a line changed, on a floating-base scenario with Model / Data replaced by this repository's
    work-alikes and the CPU stand-in engine (tests/cpu_engine.py: the oracle) answering the device calls.  Returns (config, files, outputs)."""
    import shutil

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import cpu_engine
    from flobaroid_amd.data import Data
    from flobaroid_amd.model import Model

    rident = _import_reference("identifier")
    saved = (rident.Model, rident.Data, Model.engine)

    def _engine(self):
        if self._engine is None:
            o = self.opt
            self._engine = cpu_engine.NumpyOracleEngine(self.topology, floating=o["floatingBase"], friction=o["identifyFrictionSimultaneously"],
                                                        friction_symmetric=o["identifySymmetricVelFriction"], gravity_only=o["identifyGravityParamsOnly"],
                                                        stribeck_velocity=float(o.get("stribeckVelocity", 0) or 0.0))
        return self._engine

    rident.Model, rident.Data, Model.engine = Model, Data, property(_engine)
    try:
        sc = FB_SCENARIOS[name]
        urdf = os.path.join(workdir, sc["robot"] + ".urdf")
        shutil.copy(os.path.join(REF, "model", sc["robot"] + ".urdf"), urdf)
        files = fb_measurement_files(name, workdir)
        config = fb_config(name)
        np.random.seed(3)
        idf = rident.Identification(config, urdf, None, [[p for p, _ in files]], None, None)
        idf.estimateParameters()
        idf.estimateRegressorTorques()
        m = idf.model
        outs = dict(xBase=np.array(m.xBase), xStd=np.array(m.xStd), tauEstimated=np.array(idf.tauEstimated), base_error=np.array(idf.base_error),
                    independent_cols=np.array(m.independent_cols), num_base_params=np.array(m.num_base_params), tauMeasured=np.array(m.tauMeasured),
                    num_used_samples=np.array(idf.data.num_used_samples), file_boundaries=np.array(getattr(idf.data, "file_boundaries", [0])))
        if hasattr(idf, "postid_friction"):
            for k in ("Fc", "Fv", "off"):
                outs["postid_" + k] = np.array(idf.postid_friction[k])
        return config, files, outs
    finally:
        rident.Model, rident.Data, Model.engine = saved


def reference_identification_fb(golden):
    """tests/golden/ref_identification_fb.npz: inputs (measurement arrays, option overrides) and the outputs of the reference's own
    Identification on threeLinks floating (configs/threeLinks.yaml, OLS) and on WALK-MAN with the configs/walkman_full.yaml option set."""
    import tempfile

    out = {}
    for name in FB_SCENARIOS:
        with tempfile.TemporaryDirectory() as td:
            config, files, outs = run_reference_identification_fb(name, td)
        # (the option set as the reference's constructor left it: yaml values + the overrides + what identifier.py:57-69 forces)
        out[name + "_meta"] = json.dumps({"robot": FB_SCENARIOS[name]["robot"], "yaml": FB_SCENARIOS[name]["yaml"], "over": FB_SCENARIOS[name]["over"],
                                          "contacts": FB_SCENARIOS[name]["contacts"], "files": len(files),
                                          "opt": {k: v for k, v in config.items() if isinstance(v, (int, float, str, list, dict, type(None)))}})
        for i, (_, arrs) in enumerate(files):
            for k, v in arrs.items():
                if k == "contacts":
                    for f, a in v.item().items():
                        out["%s_in%d_contacts_%s" % (name, i, f)] = a
                else:
                    out["%s_in%d_%s" % (name, i, k)] = v
        for k, v in outs.items():
            out["%s_out_%s" % (name, k)] = v
        print("  ", name, "base params", int(outs["num_base_params"]), "base_error", float(outs["base_error"]))
    np.savez_compressed(os.path.join(golden, "ref_identification_fb.npz"), **out)



# ------------------------------------------------------------------------------------------------------------------------------------
# Round 5 (N1): the reference's Fourier-series trajectory generator -- pure NumPy, no iDynTree on this path
# ------------------------------------------------------------------------------------------------------------------------------------
def reference_trajectories(golden):
    """tests/golden/ref_trajectories.npz: positions / velocities / accelerations of candidate trajectories from the reference's own
    excitation/trajectoryGenerator.py -- PulsedTrajectory.initWithParams (322-383) with OscillationGenerator (411-460, classic, rad and
    useDeg) and BoundedOscillationGenerator (462-560, joint limits), evaluated (i) through the vectorised block of
    computeTrajectoryDynamics (83-128: run unmodified with a model whose computeRegressors does nothing and the reference's own Data)
    and (ii) sample by sample through getAngle / getVelocity / getAcceleration (the fallback path, 129-150)."""
    tg = _import_reference("excitation.trajectoryGenerator")

    class NoModel:  # computeTrajectoryDynamics only calls model.computeRegressors(data) and reads data.samples["torques"] afterwards
        def computeRegressors(self, data, only_simulate=False):
            pass

    rng = np.random.default_rng(515)
    n, freq = 7, 25.0
    out = {"freq": freq, "num_dofs": n}
    limits = [(-1.0 - 0.2 * i, 0.8 + 0.3 * i) for i in range(n)]
    cases = []
    for c in range(4):
        nf = rng.integers(1, 6, n)
        a = [rng.standard_normal(int(k)) * 0.6 for k in nf]
        b = [rng.standard_normal(int(k)) * 0.6 for k in nf]
        q0 = rng.uniform(-0.4, 0.4, n)
        wf = float(rng.uniform(0.6, 1.6))
        cases.append((nf, a, b, q0, wf))
    out["num_cases"] = len(cases)
    for c, (nf, a, b, q0, wf) in enumerate(cases):
        for mode in ("classic", "classic_deg", "bounded"):
            use_deg = mode == "classic_deg"
            qq = np.rad2deg(q0) if use_deg else q0
            tr = tg.PulsedTrajectory(n, use_deg=use_deg).initWithParams(a, b, qq, nf, wf, joint_limits=limits if mode == "bounded" else None)
            cfg = {"simulateTorques": False, "floatingBase": 0, "excitationFrequency": freq, "num_dofs": n, "useDeg": use_deg, "skipSamples": 0,
                   "startOffset": 0, "verbose": 0, "showTiming": 0, "selectBlocksFromMeasurements": 0}
            td, data = tg.computeTrajectoryDynamics(cfg, tr, model=NoModel())
            T = td["positions"].shape[0]
            assert T == int(tr.getPeriodLength() * freq)
            # the same numbers sample by sample through the generator objects
            P2, V2, A2 = np.empty((T, n)), np.empty((T, n)), np.empty((T, n))
            for t in range(T):
                tr.setTime(t / freq)
                for d in range(n):
                    P2[t, d], V2[t, d], A2[t, d] = tr.getAngle(d), tr.getVelocity(d), tr.getAcceleration(d)
            if use_deg:
                P2, V2, A2 = np.deg2rad(P2), np.deg2rad(V2), np.deg2rad(A2)
            for nm, X, Y in (("q", td["positions"], P2), ("dq", td["velocities"], V2), ("ddq", td["accelerations"], A2)):
                err = np.abs(X - Y).max() / max(np.abs(Y).max(), 1.0)
                # (useDeg: the vectorised block never applies the generators' rad2deg but still converts with deg2rad -- its output is the
                # per-sample path's times pi / 180; both are stored, excitation.fourier_coefficients reproduces either)
                assert err <= 1e-11 or (use_deg and abs(err - (1 - np.pi / 180)) < 1e-9), (c, mode, nm, err)
            tag = f"c{c}_{mode}_"
            out.update({tag + "positions": td["positions"], tag + "velocities": td["velocities"], tag + "accelerations": td["accelerations"],
                        tag + "times": td["times"]})
            if use_deg:
                out.update({tag + "persample_positions": P2, tag + "persample_velocities": V2, tag + "persample_accelerations": A2})
            if mode == "bounded":
                out[tag + "q_center"] = np.array([o.q_center for o in tr.oscillators])
                out[tag + "q_range"] = np.array([o.q_range for o in tr.oscillators])
        amax = max(int(k) for k in nf)
        A = np.zeros((n, amax))
        B = np.zeros((n, amax))
        for j in range(n):
            A[j, : len(a[j])] = a[j]
            B[j, : len(b[j])] = b[j]
        out.update({f"c{c}_a": A, f"c{c}_b": B, f"c{c}_nf": np.asarray(nf), f"c{c}_q0": q0, f"c{c}_wf": wf})
    out["joint_limits"] = np.array(limits)
    np.savez_compressed(os.path.join(golden, "ref_trajectories.npz"), **out)
    print("ref_trajectories.npz:", len(out), "arrays")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "blocks_wls":
        reference_blocks_and_wls(os.path.join(REPO, "tests", "golden"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "trajectories":
        reference_trajectories(os.path.join(REPO, "tests", "golden"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fb":
        reference_identification_fb(os.path.join(REPO, "tests", "golden"))
        sys.exit(0)
    main()
    reference_host_functions(os.path.join(REPO, "tests", "golden"))
    reference_estimators(os.path.join(REPO, "tests", "golden"))
    reference_compute_regressors(os.path.join(REPO, "tests", "golden"))
    reference_blocks_and_wls(os.path.join(REPO, "tests", "golden"))
    reference_identification_fb(os.path.join(REPO, "tests", "golden"))
    reference_trajectories(os.path.join(REPO, "tests", "golden"))
