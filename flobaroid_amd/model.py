"""``Model`` work-alike of the reference's ``identification/model.py`` class (drop-in for identifier.py).

Same constructor, method names and attribute surface; the per-sample iDynTree
calls and the Python sample loop are replaced by batched HIP kernels reached through the C-ABI
(``include/fbr.h``), the small P x P reductions stay on the host with the same SciPy/NumPy calls the
reference makes.  Reference line numbers are cited at each step.

Deviations that a maintainer should know about (all documented in DESIGN.md §7):
* links and DOFs are serialised in iDynTree's traversal order (``topology.py``: depth-first, LIFO over the joints in
  document order), which reproduces every joint list the reference holds (``model/*_regressor.xml``); a regressor XML,
  ``opt['jointNames']`` / ``opt['linkNames']`` (explicit lists) or ``opt['dofOrder']`` / ``opt['linkOrder']``
  (``"traversal"`` | ``"document"``) override it;
* errors raise instead of ``sys.exit()`` / printed warnings;
* ``YStd`` is materialised on request; for very large runs it is a lazy object backed by the kernels.
"""
from __future__ import annotations

import os
from typing import Any

import numpy as np
import scipy.linalg as sla

from . import helpers
from .topology import Topology, parse_urdf


def pivoted_qr(A: np.ndarray, tie_eps: float = 1e-7):
    """``scipy.linalg.qr(A, pivoting=True, mode="economic")`` (model.py:809,841) with a DETERMINISTIC rule for ties.

    The structural regressor has columns whose pivoting norms are equal in exact arithmetic (e.g. inertia columns of a link that
    the joint symmetry maps onto each other): which one LAPACK's dgeqp3 takes is decided by the last bits of the Gram, i.e. by the
    summation order -- it differs between the reference's sample loop, the CPU oracle and the GPU reduction, and with it the
    base-parameter index set.  Rule: norms within a relative ``tie_eps`` count as tied and the LOWEST column index wins.  It is
    imposed without leaving LAPACK: the pivots are taken from dgeqp3 on A with column j scaled by 1 + tie_eps * (P - j) / P
    (column scaling multiplies the trailing column norms of every step by exactly that factor, so it only reorders ties), then the
    factor is recomputed from the UNSCALED matrix with that column order.  Outside ties the pivots, and to rounding R, are what
    the reference's call returns.  The default 1e-7 sits above the accuracy of dgeqp3's down-dated partial column norms
    (~sqrt(eps) = 1.5e-8 before LAPACK recomputes them), so a structural tie is always resolved by the rule, never by rounding;
    the WALK-MAN pivots are the same for every tolerance from 1e-9 to 1e-6.

    A tall A (the data regressor of useStructuralRegressor = 0, model.py:841) is first reduced by one unpivoted QR: pivots and R
    of a column-pivoted QR depend on A only through its triangular factor, so the pivoting runs on the small n x n matrix
    (one tall dgeqrf + one tall GEMM instead of two tall factorisations, one of them the slow dgeqp3)."""
    A = np.asarray(A)
    Pn = A.shape[1]
    if tie_eps and Pn > 0:
        d = 1.0 + float(tie_eps) * (Pn - np.arange(Pn)) / Pn
        if A.shape[0] > 2 * Pn:
            Q0, R0 = sla.qr(A, mode="economic")
            piv = sla.qr(R0 * d[None, :], pivoting=True, mode="r")[1]
            Qs, R = sla.qr(R0[:, piv], mode="economic")
            return Q0 @ Qs, R, piv
        piv = sla.qr(A * d[None, :], pivoting=True, mode="r")[1]
        Q, R = sla.qr(A[:, piv], mode="economic")
        return Q, R, piv
    return sla.qr(A, pivoting=True, mode="economic")


class LazyRegressor:
    """Stand-in for a (S*rows) x cols regressor that is too large to hold on the host.

    Supports what the consumers of ``YStd`` do (Appendix B): ``shape``, ``dot(x)`` / ``@ x`` with a
    parameter vector (-> ``fbr_predict``) and explicit materialisation (``np.asarray``)."""

    def __init__(self, engine, states: dict, cols: list[int] | np.ndarray | None = None):
        self._eng = engine
        self._st = states
        self._cols = None if cols is None else np.asarray(cols, dtype=np.int64)
        S = np.asarray(states["q"]).shape[0]
        self.shape = (S * engine.rows, engine.cols if cols is None else len(self._cols))
        self.ndim = 2
        self.dtype = np.dtype(np.float64)

    def dot(self, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim != 1 or x.shape[0] != self.shape[1]:
            raise ValueError("LazyRegressor.dot supports a parameter vector of matching length")
        xf = x
        if self._cols is not None:
            xf = np.zeros(self._eng.cols)
            xf[self._cols] = x
        return self._eng.predict(self._st, xf).reshape(-1)

    __matmul__ = dot

    def __array__(self, dtype=None, copy=None):
        Y = self._eng.regressor(self._st)
        if self._cols is not None:
            Y = Y[:, self._cols]
        return Y if dtype is None else Y.astype(dtype)

    def __getitem__(self, idx):
        return np.asarray(self)[idx]


class Model:
    def __init__(self, opt: dict[str, Any], urdf_file: str, regressor_file: str | None = None,
                 regressor_init: bool = True, device: int = 0) -> None:
        self.urdf_file = urdf_file
        self.opt = opt
        self.device = device

        # attributes set by Identification after construction (model.py:33-43)
        self.xBase = np.array([])
        self.xBaseModel = np.array([])
        self.YBaseInv = np.array([])
        self.xStd = np.array([])
        self.contactForcesSum = np.array([])
        self.non_id: list[int] = []
        self.identifiable: list[int] = []
        self.progress = helpers.Progress(opt).progress

        # defaults / forced debug options (model.py:48-57)
        opt.setdefault("orthogonalizeBasis", 1)
        opt.setdefault("useBasisProjection", 0)
        opt["useRegressorForSimulation"] = 0
        opt["addContacts"] = 1
        for k, v in (("verbose", 0), ("showTiming", 0), ("skipSamples", 0), ("identifyFrictionSimultaneously", 0),
                     ("identifySymmetricVelFriction", 1), ("identifyGravityParamsOnly", 0), ("simulateTorques", 0),
                     ("useAPriori", 0), ("useStructuralRegressor", 1), ("filterRegressor", 0), ("minTol", 1e-4),
                     ("randomSamples", 5000), ("estimateWith", "std")):
            opt.setdefault(k, v)

        # load the robot description (model.py:60-68; raises instead of sys.exit())
        # serialisation (model.py:73-98,121-127: jointNames / linkNames as iDynTree reports them): traversal order by default,
        # opt['linkOrder'] / opt['dofOrder'] = "traversal" | "document", explicit opt['linkNames'] / opt['jointNames'] lists,
        # and the joint list of a regressor XML (model.py:74-85) win in that order
        topo = Topology.load(urdf_file, link_order=opt.get("linkOrder"), dof_order=opt.get("dofOrder"))
        if opt.get("linkNames"):
            topo = topo.reordered_links([str(n) for n in opt["linkNames"]])
        if opt.get("jointNames"):
            topo = topo.reordered_dofs([str(n) for n in opt["jointNames"]])
        if regressor_file:
            import xml.etree.ElementTree as ET

            with open(regressor_file) as f:
                tree = ET.fromstring(f.read())
            names = [(e.text or "").strip() for e in tree.iter() if e.tag == "joint"]
            topo = topo.reordered_dofs(names)
        self.topology = topo
        self.jointNames = list(topo.dof_names)
        self.num_dofs = topo.num_dofs
        # handles of the iDynTree objects some out-of-scope callers touch (Appendix B): not provided
        self.loader = self.idyn_model = self.kinDyn = None

        self.N_OUT = self.num_dofs + 6 if opt["floatingBase"] else self.num_dofs  # model.py:102-105
        self.num_links = topo.num_links
        self.inertia_params: list[int] = []
        self.mass_params: list[int] = []
        for i in range(self.num_links):  # model.py:116-120
            self.mass_params.append(i * 10)
            self.inertia_params.extend([i * 10 + 4, i * 10 + 5, i * 10 + 6, i * 10 + 7, i * 10 + 8, i * 10 + 9])
        self.linkNames = list(topo.link_names)
        self.limits = {k: dict(v) for k, v in topo.limits.items()}

        # parameter counts (model.py:130-168)
        nd = self.num_dofs
        self.num_model_params = self.num_links * 10
        self.num_all_params = self.num_model_params
        if opt["identifyFrictionSimultaneously"]:
            self.num_identified_params = self.num_model_params + nd
            self.num_all_params += nd
            if not opt["identifyGravityParamsOnly"]:
                nv = nd if opt["identifySymmetricVelFriction"] else 2 * nd
                self.num_identified_params += nv + nd
                self.num_all_params += nv + nd
                if opt.get("stribeckVelocity", 0) > 0:
                    self.num_identified_params += nd
                    self.num_all_params += nd
        else:
            self.num_identified_params = self.num_model_params
        self.friction_params_start = self.num_model_params
        if opt["identifyGravityParamsOnly"]:
            self.num_identified_params -= len(self.inertia_params)
            self.friction_params_start = self.num_model_params - len(self.inertia_params)

        self.baseNames = ["base f_x", "base f_y", "base f_z", "base m_x", "base m_y", "base m_z"]
        self.gravity = [0, 0, -9.81, 0, 0, 0]
        self.gravity_twist = self.gravity_vec = None

        # a-priori parameters (model.py:189-208 + helpers.ParamHelpers.addFrictionFromURDF, helpers.py:438-471)
        self.xStdModel = topo.x_std()
        if opt["identifyFrictionSimultaneously"]:
            x = np.concatenate((self.xStdModel, np.zeros(self.num_all_params - self.num_model_params)))
            start = self.num_model_params
            for i, j in enumerate(self.jointNames):
                fr = topo.friction.get(j, {"f_constant": 0.0, "f_velocity": 0.0})
                x[start + i] = fr["f_constant"]
                if not opt["identifyGravityParamsOnly"]:
                    x[start + nd + i] = fr["f_velocity"]
                    if not opt["identifySymmetricVelFriction"]:
                        x[start + 2 * nd + i] = fr["f_velocity"]
            if opt.get("stribeckVelocity", 0) > 0 and not opt["identifyGravityParamsOnly"]:
                fs = self.num_all_params - nd
                for i in range(nd):
                    fc = x[start + i]
                    x[fs + i] = abs(fc) * 0.6 if abs(fc) > 0 else 0.0
            self.xStdModel = x
        if opt["estimateWith"] == "urdf":
            self.xStd = self.xStdModel

        self._engine = None
        if regressor_init:
            self.computeRegressorLinDepsQR()  # model.py:213-216

    # ------------------------------------------------------------------------------------ engine
    @property
    def engine(self):
        """The C-ABI handle (created on first use, after fork; fails loudly without a HIP device)."""
        if self._engine is None:
            from ._lib import Engine

            o = self.opt
            self._engine = Engine(self.topology, floating=o["floatingBase"], friction=o["identifyFrictionSimultaneously"],
                                  friction_symmetric=o["identifySymmetricVelFriction"],
                                  gravity_only=o["identifyGravityParamsOnly"],
                                  stribeck_velocity=float(o.get("stribeckVelocity", 0) or 0.0),
                                  gravity=self.gravity[:3], device=self.device, options=o.get("engineOptions"))
            assert self._engine.rows == self.N_OUT and self._engine.cols == self.num_identified_params
        return self._engine

    def getDescriptionOfParameters(self) -> str:  # model.py:218-237
        names = ["mass", "first moment of mass (x)", "first moment of mass (y)", "first moment of mass (z)",
                 "moment of inertia (xx)", "moment of inertia (xy)", "moment of inertia (xz)", "moment of inertia (yy)",
                 "moment of inertia (yz)", "moment of inertia (zz)"]
        desc = ""
        for i in range(self.num_links):
            for j, pname in enumerate(names):
                desc += f"Parameter {i * 10 + j}: {pname} of link {self.linkNames[i]}\n"
        return desc

    # ------------------------------------------------------------------------------------ states
    def _states_from_samples(self, samples: dict, idx) -> dict:
        st = {"q": samples["positions"][idx], "dq": samples["velocities"][idx], "ddq": samples["accelerations"][idx]}
        if self.opt["floatingBase"]:
            st["base_vel"] = samples["base_velocity"][idx]
            st["base_acc"] = samples["base_acceleration"][idx]
            st["rpy"] = samples["base_rpy"][idx]
        if self.opt["identifyFrictionSimultaneously"]:
            st["sign"] = helpers.getFrictionSignSeries(samples, self.opt)[idx]
        return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in st.items()}

    def _x_for_simulation(self, xStdModel=None) -> np.ndarray:
        """Standard vector in the index layout simulateDynamicsIDynTree reads (model.py:299-326): inertial
        parameters at 0..10L, friction parameters at friction_params_start.. ."""
        x = self.xStdModel if xStdModel is None else np.asarray(xStdModel, dtype=np.float64)
        if not self.opt["identifyFrictionSimultaneously"]:
            return np.ascontiguousarray(x[: self.num_model_params])
        return np.ascontiguousarray(x)

    def simulateDynamicsIDynTree(self, samples: dict[str, np.ndarray], sample_idx: int, kinDyn: Any = None,
                                 xStdModel: np.ndarray | None = None) -> np.ndarray:
        """Torques (and base wrench) of ONE sample (model.py:239-331).  Batched use: ``simulateDynamicsBatch``."""
        return self.simulateDynamicsBatch(samples, np.array([sample_idx]), xStdModel)[0]

    def simulateDynamicsBatch(self, samples: dict[str, np.ndarray], idx, xStdModel=None) -> np.ndarray:
        st = self._states_from_samples(samples, idx)
        vel_sign = None
        if self.opt["identifyFrictionSimultaneously"] and self.opt.get("stribeckVelocity", 0) > 0:
            vel_sign = np.ascontiguousarray(helpers.getFrictionSignVelocities(samples, self.opt)[idx], dtype=np.float64)
        return self.engine.inverse_dynamics(st, self._x_for_simulation(xStdModel), vel_sign=vel_sign)

    # ------------------------------------------------------------------------------------ computeRegressors
    def computeRegressors(self, data, only_simulate: bool = False) -> None:
        """Regressors, torques and contact terms of all used samples (model.py:333-632), batched."""
        self.data = data
        opt = self.opt
        fb = 6 if opt["floatingBase"] else 0
        nd = self.num_dofs
        S = data.num_used_samples
        dim = nd + fb
        samples = data.samples
        idx = np.arange(S) * (opt["skipSamples"] + 1)  # m_idx, model.py:371
        num_time = simulate_time = 0.0

        if opt["identifyGravityParamsOnly"]:  # model.py:382-385 (mutates the samples like the reference)
            samples["velocities"][idx] = 0.0
            samples["accelerations"][idx] = 0.0

        st = self._states_from_samples(samples, idx)
        self._states = st
        torq = np.array(samples["torques"][idx], dtype=np.float64)

        self.torquesAP_stack = np.zeros(dim * S)
        self.sim_torq_stack = np.zeros(dim * S)
        with helpers.Timer() as t:
            if opt["simulateTorques"] or opt["useAPriori"] or opt["floatingBase"]:  # model.py:398-413
                sim = np.nan_to_num(self.simulateDynamicsBatch(samples, idx))
                if opt["useAPriori"]:
                    self.torquesAP_stack = sim.reshape(-1).copy()
                if opt["simulateTorques"]:
                    torq = sim
                elif opt["floatingBase"] and torq.shape[1] < dim:
                    torq = np.concatenate((sim[:, 0:6], torq), axis=1)
        simulate_time += t.interval
        self.torques_stack = np.ascontiguousarray(torq).reshape(-1)

        # contacts (model.py:359-361, 535-560)
        contact_dict = samples["contacts"].item(0) if "contacts" in samples else {}
        frames = list(contact_dict.keys())
        self.contacts_stack = np.zeros((len(frames), dim * S))
        for c, frame in enumerate(frames):
            w = np.ascontiguousarray(contact_dict[frame][idx], dtype=np.float64)
            try:
                self.contacts_stack[c] = self.engine.contact_torques(st, str(frame), w).reshape(-1)
            except KeyError:
                continue  # unknown frame: the reference skips it (model.py:543-544)
        self.contactForcesSum = np.sum(self.contacts_stack, axis=0)

        if opt["floatingBase"]:  # model.py:562-576
            if opt["simulateTorques"]:
                if opt["addContacts"]:
                    self.torques_stack = self.torques_stack + self.contactForcesSum
            else:
                t2 = self.torques_stack.reshape(S, dim)
                self.contactForcesSum_2dim = self.contactForcesSum.reshape(S, dim)
                if opt["addContacts"]:
                    t2[:, :6] += self.contactForcesSum_2dim[:, :6]
                self.torques_stack = t2.flatten()
        if opt["addContacts"]:
            self.sim_torq_stack = self.sim_torq_stack + self.contactForcesSum
        if len(frames) or opt["simulateTorques"]:  # model.py:581-583
            self.data.samples["torques"] = np.reshape(self.torques_stack, (S, dim))

        self.tau = self.torques_stack - self.torquesAP_stack if opt["useAPriori"] else self.torques_stack

        if not only_simulate:
            with helpers.Timer() as t:
                limit = float(opt.get("materializeLimitBytes", 8e9))
                if dim * S * self.num_identified_params * 8.0 <= limit:
                    self.regressor_stack = self.engine.regressor(st)
                else:
                    self.regressor_stack = LazyRegressor(self.engine, st)
            num_time += t.interval
        else:
            self.regressor_stack = np.zeros((dim * S, self.num_identified_params))
        self.YStd = self.regressor_stack

        if not only_simulate:
            # fused reductions of [YStd | tau | contactForcesSum]: everything the estimators need
            rhs = np.stack((self.tau, self.contactForcesSum), axis=1)
            self.G_aug = self.engine.gram(st, rhs=rhs)

        if not opt["useStructuralRegressor"] and not only_simulate:  # model.py:598-601
            self.computeRegressorLinDepsQR(self.YStd)

        if not only_simulate:
            if opt["useBasisProjection"]:
                if isinstance(self.YStd, LazyRegressor):
                    raise ValueError("useBasisProjection needs the materialised YStd (raise opt['materializeLimitBytes'])")
                self.YBase = np.dot(self.YStd, self.B)  # model.py:603-604
            elif isinstance(self.YStd, LazyRegressor):
                self.YBase = LazyRegressor(self.engine, st, cols=self.independent_cols)
            else:
                # YStd @ Pb is a column gather (Pb[:, i] = e_{P[i]}, model.py:876-880, 606)
                self.YBase = np.ascontiguousarray(self.YStd[:, self.independent_cols])
            if opt["filterRegressor"]:
                # model.py:608-615, literally (including its stride of num_dofs where the rows of a sample are N_OUT apart: the two
                # agree for fixed-base models, the only ones the option makes sense for)
                if isinstance(self.YBase, LazyRegressor):
                    raise ValueError("filterRegressor needs the materialised YBase (raise opt['materializeLimitBytes'])")
                from scipy import signal

                b, a = signal.butter(5, opt["filterRegCutoff"] / (samples["frequency"] / 2), btype="low", analog=False)
                for j in range(self.num_base_inertial_params):
                    for i in range(self.num_dofs):
                        self.YBase[i::self.num_dofs, j] = signal.filtfilt(b, a, self.YBase[i::self.num_dofs, j])

        self.sample_end = samples["positions"].shape[0]
        if opt["skipSamples"] > 0:
            self.sample_end -= opt["skipSamples"]
        self.tauMeasured = np.reshape(self.torques_stack, (S, dim))
        if "times" in samples:
            self.T = samples["times"][0 : self.sample_end : opt["skipSamples"] + 1]
        if opt["showTiming"]:
            print(f"(simulation for regressors took {simulate_time:.3f} sec.)")
            print(f"(getting regressors took {num_time:.3f} sec.)")

    # ------------------------------------------------------------------------------------ structural regressor
    def _random_states(self, n_samples: int) -> dict:
        """Random states in the reference's global-RNG call order (model.py:690-735)."""
        nd = self.num_dofs
        q = np.empty((n_samples, nd))
        dq = np.empty((n_samples, nd))
        ddq = np.empty((n_samples, nd))
        fbase = bool(self.opt["floatingBase"])
        bv = np.empty((n_samples, 6))
        ba = np.empty((n_samples, 6))
        rpy = np.empty((n_samples, 3))
        have_limits = len(self.limits) > 0
        if have_limits:
            jn = self.jointNames
            hi = np.array([self.limits[j]["upper"] for j in jn])
            lo = np.array([self.limits[j]["lower"] for j in jn])
            vm = np.array([self.limits[j]["velocity"] for j in jn])
            rng_q = hi - lo
        grav = bool(self.opt["identifyGravityParamsOnly"])
        for i in range(n_samples):
            if have_limits:
                q[i] = lo + rng_q * np.random.rand(nd)
                if grav:
                    dq[i] = 0.0
                    ddq[i] = 0.0
                else:
                    dq[i] = (np.random.rand(nd) - 0.5) * 2 * vm
                    ddq[i] = (np.random.rand(nd) - 0.5) * 2 * np.pi
            else:
                q[i] = (np.random.ranf(nd) * 2 - 1) * np.pi
                dq[i] = (np.random.ranf(nd) * 2 - 1) * np.pi
                ddq[i] = (np.random.ranf(nd) * 2 - 1) * np.pi
            if fbase:
                bv[i] = np.pi * np.random.rand(6)
                ba[i] = np.pi * np.random.rand(6)
                if grav:
                    bv[i] = 0.0
                    ba[i] = 0.0
                rpy[i] = np.random.ranf(3) * 0.1
        st = {"q": q, "dq": dq, "ddq": ddq}
        if fbase:
            st.update(base_vel=bv, base_acc=ba, rpy=rpy)
        if self.opt["identifyFrictionSimultaneously"]:
            st["sign"] = np.tanh(dq / float(self.opt.get("frictionSignThreshold", 0.02)))  # model.py:757-758
        return st

    CACHE_PRODUCER = "flobaroid_amd/3"  # /2: pivots follow the lowest-index tie rule; /3: traversal serialisation, tie tolerance 1e-7

    def _tie_eps(self) -> float:
        """opt['pivotTieTolerance'] (default 1e-7; 0 = LAPACK's own last-bit tie breaking, as the reference)."""
        return float(self.opt.get("pivotTieTolerance", 1e-7))

    def _dof_hash(self) -> str:
        import hashlib

        return hashlib.sha1("\n".join(self.jointNames + ["|"] + self.linkNames).encode()).hexdigest()

    def getRandomRegressor(self, n_samples=None):
        """Structural Gram R = sum A^T A over random states + its pivoted QR, with the reference's npz
        cache (model.py:634-830).  The Gram is the raw sum (never normalised): minTol applies to it."""
        opt = self.opt
        suffix = ".gravity_regressor.npz" if opt["identifyGravityParamsOnly"] else ".regressor.npz"
        regr_filename = self.urdf_file + suffix
        fb = opt["floatingBase"]
        # The file name and the reference's keys are kept (model.py:811-822) so the tools around it find the cache, but a cache
        # is only trusted when it was written by this implementation for the same link / DOF serialisation and friction layout
        # (the reference's own validity check ignores the serialisation and stribeckVelocity).  A cache somebody else wrote is
        # never overwritten: ours then goes to <name>.fbr.npz beside it.
        dof_hash = self._dof_hash()
        stribeck = float(opt.get("stribeckVelocity", 0) or 0.0)
        own_filename = regr_filename[:-4] + ".fbr.npz"
        generate_new = True
        foreign = False
        for fn in (regr_filename, own_filename):
            try:
                f = np.load(fn)
                if "producer" not in f:
                    foreign = foreign or fn == regr_filename
                    # opt['useReferenceRegressorCache'] = 1: a cache the reference itself wrote is adopted when it passes the
                    # reference's own validity check (model.py:663-671) -- its Gram, its pivots, hence ITS choice among tied pivots and its
                    # base-parameter basis; the caller vouches for the serialisation (links and DOFs in iDynTree's traversal order, which
                    # is also ours) and for stribeckVelocity, which that check does not look at
                    if fn == regr_filename and opt.get("useReferenceRegressorCache", 0):
                        R, Q, RQ, PQ = f["R"], f["Q"], f["RQ"], f["PQ"]
                        if not (f["n"] != n_samples or f["fb"] != fb or R.shape[0] != self.num_identified_params
                                or opt["identifyGravityParamsOnly"] != f["grav_only"] or f["fric"] != opt["identifyFrictionSimultaneously"]
                                or f["fric_sym"] != opt["identifySymmetricVelFriction"]):
                            generate_new = False
                            break
                    continue
                R, Q, RQ, PQ = f["R"], f["Q"], f["RQ"], f["PQ"]
                if not (f["n"] != n_samples or f["fb"] != fb or R.shape[0] != self.num_identified_params
                        or opt["identifyGravityParamsOnly"] != f["grav_only"] or f["fric"] != opt["identifyFrictionSimultaneously"]
                        or f["fric_sym"] != opt["identifySymmetricVelFriction"]
                        or str(f["producer"]) != self.CACHE_PRODUCER or str(f["dof_hash"]) != dof_hash
                        or float(f["stribeck"]) != stribeck):
                    generate_new = False
                    break
            except (OSError, KeyError, ValueError):
                continue
        if generate_new:
            if not n_samples:
                n_samples = self.num_dofs * 1000
            st = self._random_states(int(n_samples))
            R = self.engine.gram(st)
            Q, RQ, PQ = pivoted_qr(R, self._tie_eps())  # model.py:809
            try:
                np.savez(own_filename if foreign else regr_filename, R=R, Q=Q, RQ=RQ, PQ=PQ, n=n_samples, fb=opt["floatingBase"],
                         grav_only=opt["identifyGravityParamsOnly"], fric=opt["identifyFrictionSimultaneously"],
                         fric_sym=opt["identifySymmetricVelFriction"], producer=self.CACHE_PRODUCER, dof_hash=dof_hash,
                         stribeck=stribeck, jointNames=np.array(self.jointNames), linkNames=np.array(self.linkNames))
            except OSError:
                pass  # read-only model directory: skip the cache
        return R, Q, RQ, PQ

    # ------------------------------------------------------------------------------------ base parameters
    def computeRegressorLinDepsQR(self, regressor=None):
        """Independent columns, permutation and the base projection K (model.py:832-1052)."""
        opt = self.opt
        if regressor is not None:
            Y = regressor
            if isinstance(Y, LazyRegressor):
                # pivots and |diag R| of a pivoted QR depend on Y only through R with R^T R = Y^T Y
                Rt = self.engine.tsqr(Y._st)
                self.Q, self.R, self.P = pivoted_qr(Rt, self._tie_eps())
            else:
                self.Q, self.R, self.P = pivoted_qr(Y, self._tie_eps())  # model.py:841
        else:
            Y, self.Q, self.R, self.P = self.getRandomRegressor(n_samples=opt["randomSamples"])

        r = int(np.where(np.abs(self.R.diagonal()) > opt["minTol"])[0].size)  # model.py:871
        self.num_base_params = r
        self.num_base_inertial_params = r - self.num_dofs
        self.Pp = np.zeros((self.P.size, self.P.size))
        for i in self.P:
            self.Pp[i, self.P[i]] = 1
        self.Pb = self.Pp.T[:, 0:r]
        self.Pd = self.Pp.T[:, r:]
        self.independent_cols = self.P[0:r]
        R1 = self.R[0:r, 0:r]
        R2 = self.R[0:r, r:]
        self.linear_deps = sla.inv(R1).dot(R2)
        self.linear_deps[np.abs(self.linear_deps) < opt["minTol"]] = 0
        self.Kd = self.linear_deps
        self.K = self.Pb.T + self.Kd.dot(self.Pd.T)
        if opt["useBasisProjection"]:
            # grouped columns of every independent column -> basis B (model.py:896-929; identifier.py:57 forces the option off, the
            # method still supports it for the other callers)
            self.B = np.zeros((self.num_identified_params, self.num_base_params))
            big = np.abs(self.linear_deps) > opt["minTol"]
            for j in range(self.linear_deps.shape[0]):
                kk = np.flatnonzero(big[j])
                self.B[self.P[r + kk], j] = self.linear_deps[j, kk]
                self.B[self.independent_cols[j], j] = 1
            if opt["orthogonalizeBasis"]:
                import numpy.linalg as la

                Q_B_qr, R_B_qr = la.qr(self.B)
                Q_B_qr[np.abs(Q_B_qr) < opt["minTol"]] = 0
                d = np.diag(R_B_qr)
                sgn = np.where(np.abs(d) < opt["minTol"], 0.0, np.sign(d))
                self.B = Q_B_qr.dot(np.diag(sgn))
                self.Binv = self.B.T
            else:
                import numpy.linalg as la

                self.Binv = la.pinv(self.B)

        # indices of the identified parameters within the full vector (model.py:936-1022)
        self.identified_params = []
        for i in range(self.num_links):
            self.identified_params.extend([i * 10, i * 10 + 1, i * 10 + 2, i * 10 + 3])
            if not opt["identifyGravityParamsOnly"]:
                self.identified_params.extend([i * 10 + 4, i * 10 + 5, i * 10 + 6, i * 10 + 7, i * 10 + 8, i * 10 + 9])
        if opt["identifyFrictionSimultaneously"]:
            mp, nd = self.num_model_params, self.num_dofs
            self.identified_params.extend(range(mp, mp + nd))
            if not opt["identifyGravityParamsOnly"]:
                nv = nd if opt["identifySymmetricVelFriction"] else 2 * nd
                self.identified_params.extend(range(mp + nd, mp + nd + nv + nd))
                if opt.get("stribeckVelocity", 0) > 0:
                    fs = self.num_all_params - nd
                    self.identified_params.extend(range(fs, fs + nd))
        # parameters without any influence on a base parameter (model.py:1043-1052): a symbol is free in
        # base_deps = K * syms iff its column of K is not identically zero
        if opt["useBasisProjection"]:  # base_deps = syms . B, or syms . pinv(B)^T thresholded (model.py:1029-1037)
            if opt["orthogonalizeBasis"]:
                used = np.any(self.B != 0, axis=1)
            else:
                import numpy.linalg as la

                Bz = la.pinv(self.B)
                Bz[np.abs(Bz) < opt["minTol"]] = 0
                used = np.any(Bz != 0, axis=0)
        else:
            used = np.any(self.K != 0, axis=0)
        idp = np.array(self.identified_params)
        ident = set(idp[used].tolist())
        self.non_id = [p for p in range(self.num_all_params) if p not in ident]
        self.identifiable = [p for p in range(self.num_all_params) if p not in self.non_id]
        self._syms_ready = False

    # symbolic bookkeeping (model.py:931-1041) is built on first use: sympy on a 213x480 K is slow
    def _build_symbols(self) -> None:
        import sympy
        from sympy import Matrix, symbols

        self.base_syms = sympy.Matrix([sympy.Symbol("beta" + str(i), real=True) for i in range(self.num_base_params)])
        syms: list[Any] = []
        self.mass_syms, self.friction_syms = [], []
        for i in range(self.num_links):
            m = symbols(f"m_{i}")
            syms.append(m)
            self.mass_syms.append(m)
            syms.extend([symbols(f"c_{i}x"), symbols(f"c_{i}y"), symbols(f"c_{i}z")])
            p = f"I_{i}"
            syms.extend([symbols(p + "xx"), symbols(p + "xy"), symbols(p + "xz"), symbols(p + "yy"), symbols(p + "yz"),
                         symbols(p + "zz")])
        if self.opt["identifyFrictionSimultaneously"]:
            nd = self.num_dofs
            names = ["Fc"]
            if not self.opt["identifyGravityParamsOnly"]:
                names += ["Fv"] if self.opt["identifySymmetricVelFriction"] else ["Fv+", "Fv-"]
                names += ["off"]
                if self.opt.get("stribeckVelocity", 0) > 0:
                    names += ["Fs"]
            for nm in names:
                for i in range(nd):
                    s = symbols(f"{nm}_{i}")
                    syms.append(s)
                    self.friction_syms.append(s)
        self._param_syms = np.array(syms)
        if self.opt["useBasisProjection"]:  # model.py:1029-1037
            if self.opt["orthogonalizeBasis"]:
                self._base_deps = np.dot(self._param_syms[self.identified_params], self.B)
            else:
                import numpy.linalg as la

                Bz = la.pinv(self.B)
                Bz[np.abs(Bz) < self.opt["minTol"]] = 0
                self._base_deps = np.dot(self._param_syms[self.identified_params], Bz.T)
        else:
            self._base_deps = Matrix(self.K) * Matrix(self._param_syms[self.identified_params])
        self._syms_ready = True

    @property
    def param_syms(self):
        if not getattr(self, "_syms_ready", False):
            self._build_symbols()
        return self._param_syms

    @property
    def base_deps(self):
        """sympy Matrix(K) * Matrix(param_syms[identified_params]) (model.py:1041), built lazily."""
        if not hasattr(self, "K"):
            return np.array([])
        if not getattr(self, "_syms_ready", False):
            self._build_symbols()
        return self._base_deps

    def base_columns_of_links(self) -> list[list[int]]:
        """Per link the base-regressor columns that depend on one of its standard parameters: row j of K has a non-zero in the
        link's columns (the reference asks sympy whether a parameter symbol is free in base_deps[j], model.py:1069-1076)."""
        idp = {k: i for i, k in enumerate(self.identified_params)}
        out = []
        for i in range(self.num_links):
            cols = [idp[k] for k in range(i * 10, i * 10 + 10) if k in idp]
            out.append([j for j in range(self.num_base_params) if np.any(self.K[j, cols] != 0)])
        return out

    def base_factor(self) -> np.ndarray:
        """Upper-triangular R with R^T R = YBase^T YBase (same singular values as YBase): the TSQR of the path when YBase is
        the plain column gather of the regressor, a host QR when it was post-processed on the host (filterRegressor,
        useBasisProjection) or no states are attached."""
        plain = not (self.opt.get("filterRegressor") or self.opt.get("useBasisProjection"))
        if plain and getattr(self, "_states", None) is not None:
            return np.asarray(self.engine.tsqr(self._states, cols=np.asarray(self.independent_cols, dtype=np.int32)))
        return np.linalg.qr(np.asarray(self.YBase), mode="r")

    def getSubregressorsConditionNumbers(self, R=None):  # model.py:1054-1086
        """Condition number of the base sub-regressor of every link.  cond(YBase[:, cols]) = cond(R[:, cols]) for the
        triangular factor R of YBase, so nothing tall is touched (R: ``base_factor()``, computed when not given)."""
        import numpy.linalg as la

        if R is None:
            R = self.base_factor()
        return [1e16 if not cols else float(la.cond(R[:, cols])) for cols in self.base_columns_of_links()]
