"""ctypes front-end of the CPU ORACLE (oracle/fbr_oracle.c) -- test infrastructure only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``flobaroid_amd``) never does.  Parity status: see the header of
``fbr_oracle.c`` ("parity unpinned" at the iDynTree boundary; pinned on the documented
known answers in ``tests/golden``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libfbr_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fbr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _d(a):
    if a is None:
        return None
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _c(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == shape, (a.shape, shape)
    return a


class OracleModel:
    """Bundles a Topology with the layout options the reference's ``opt`` dict carries."""

    def __init__(self, topo, floating=False, fric=False, fric_sym=True, grav_only=False, stribeck=0.0,
                 gravity=(0.0, 0.0, -9.81)):
        self.topo = topo
        self.L = topo.num_links
        self.n = topo.num_dofs
        self.floating = int(bool(floating))
        self.fric = int(bool(fric))
        self.fric_sym = int(bool(fric_sym))
        self.grav_only = int(bool(grav_only))
        self.stribeck = float(stribeck)
        self.order = np.array(topo.traversal(), dtype=np.int32)
        self.parent = np.array(topo.parent, dtype=np.int32)
        self.dof = np.array(topo.dof_index, dtype=np.int32)
        self.restR = np.ascontiguousarray(topo.rest_R, dtype=np.float64).reshape(-1)
        self.restp = np.ascontiguousarray(topo.rest_p, dtype=np.float64).reshape(-1)
        self.axis = np.ascontiguousarray(topo.axis, dtype=np.float64).reshape(-1)
        jt = getattr(topo, "joint_type", None)
        self.jtype = np.array([1 if d >= 0 else 0 for d in topo.dof_index] if jt is None else list(jt), dtype=np.int32)  # 2: prismatic
        self.gravity = np.array(gravity, dtype=np.float64)
        self.rows = self.n + (6 if self.floating else 0)
        P = 4 * self.L if self.grav_only else 10 * self.L
        if self.fric:
            P += self.n
            if not self.grav_only:
                P += self.n if self.fric_sym else 2 * self.n
                P += self.n
                if self.stribeck > 0:
                    P += self.n
        self.P = P

    def _model_args(self):
        return (self.L, self.n, _i(self.order), _i(self.parent), _i(self.dof), _d(self.restR), _d(self.restp),
                _d(self.axis), _i(self.jtype), self.floating, _d(self.gravity))

    def _states(self, st):
        q = _c(st["q"])
        S = q.shape[0]
        dq = _c(st["dq"], (S, self.n))
        ddq = _c(st["ddq"], (S, self.n))
        if self.floating:
            bv = _c(st["base_vel"], (S, 6))
            ba = _c(st["base_acc"], (S, 6))
            rpy = _c(st["rpy"], (S, 3))
        else:
            bv = ba = rpy = None
        return S, q, dq, ddq, bv, ba, rpy

    def regressor(self, st, sign=None):
        """Stacked standard regressor (S*rows, P) -- the reference's ``regressor_stack``."""
        S, q, dq, ddq, bv, ba, rpy = self._states(st)
        sign = _c(sign, (S, self.n)) if self.fric else None
        Y = np.empty((S * self.rows, self.P))
        rc = lib().orc_regressor_batch(*self._model_args(), self.fric, self.fric_sym, self.grav_only,
                                       ctypes.c_double(self.stribeck), ctypes.c_long(S), _d(q), _d(dq), _d(ddq),
                                       _d(bv), _d(ba), _d(rpy), _d(sign), _d(Y))
        assert rc == 0
        return Y

    def inverse_dynamics(self, st, x_std, sign=None, vel_sign=None):
        """(S, rows) generalized torques [base wrench; joint torques] incl. the friction model."""
        S, q, dq, ddq, bv, ba, rpy = self._states(st)
        x_std = _c(x_std)
        sign = _c(sign, (S, self.n)) if self.fric else None
        vel_sign = _c(vel_sign, (S, self.n)) if (self.fric and self.stribeck > 0) else None
        tau = np.zeros((S, self.rows))
        rc = lib().orc_inverse_dynamics_batch(*self._model_args(), self.fric, self.fric_sym, self.grav_only,
                                              ctypes.c_double(self.stribeck), ctypes.c_long(S), _d(q), _d(dq),
                                              _d(ddq), _d(bv), _d(ba), _d(rpy), _d(sign), _d(vel_sign),
                                              _d(x_std), _d(tau))
        assert rc == 0
        return tau

    def stack_block(self, st, x_std, out=None, threads=0):
        """[Y | tau] of the samples, (S*rows, P+1), the reference's loop body dealt to OpenMP threads (``orc_stack_block_omp``; models
        without friction columns).  Returns (block, threads that took part).  Only bench.py's all-core CPU baseline uses it."""
        assert not self.fric and not self.grav_only
        S, q, dq, ddq, bv, ba, rpy = self._states(st)
        A = out if out is not None else np.empty((S * self.rows, self.P + 1))
        assert A.shape == (S * self.rows, self.P + 1) and A.flags.c_contiguous
        rc = lib().orc_stack_block_omp(*self._model_args(), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy),
                                       _d(_c(x_std)), int(threads), _d(A))
        assert rc > 0, rc
        return A, int(rc)

    def stack_gram(self, st, x_std, threads=0):
        """Upper triangle of [Y | tau]^T [Y | tau] over the samples, accumulated inside the OpenMP loop of ``orc_stack_gram_omp`` (per-thread
        sums, added at the end).  Returns (G, threads).  Only bench.py's all-core CPU baseline and its test use it."""
        assert not self.fric and not self.grav_only
        S, q, dq, ddq, bv, ba, rpy = self._states(st)
        G = np.zeros((self.P + 1, self.P + 1))
        rc = lib().orc_stack_gram_omp(*self._model_args(), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(_c(x_std)),
                                      int(threads), _d(G))
        assert rc > 0, rc
        return G, int(rc)

    def contact_torques(self, st, frame, wrench):
        """(S, rows): J_frame^T w per sample; ``frame`` = link name or a frame name of the topology."""
        S, q, dq, ddq, bv, ba, rpy = self._states(st)
        t = self.topo
        if frame in t.frames:
            fl, fR, fp = t.frames[frame]["link"], t.frames[frame]["R"], t.frames[frame]["p"]
        else:
            fl, fR, fp = t.link_names.index(frame), np.eye(3), np.zeros(3)
        fR = _c(np.asarray(fR).reshape(-1))
        fp = _c(np.asarray(fp).reshape(-1))
        w = _c(wrench, (S, 6))
        out = np.zeros((S, self.rows))
        rc = lib().orc_contact_torques(*self._model_args(), ctypes.c_long(S), _d(q), _d(rpy), int(fl), _d(fR),
                                       _d(fp), _d(w), _d(out))
        assert rc == 0
        return out


def gram(A, rhs=None):
    """[A|rhs]^T [A|rhs] by the plain triple loop (raw sum, model.py:803-806)."""
    A = _c(A)
    M, P = A.shape
    k = 0
    if rhs is not None:
        rhs = _c(rhs).reshape(M, -1)
        k = rhs.shape[1]
    G = np.zeros((P + k, P + k))
    rc = lib().orc_gram(ctypes.c_long(M), P, _d(A), k, _d(rhs), _d(G))
    assert rc == 0
    return G


# ---------------------------------------------------------------------------------------------
# host-side restatement of the reference's reductions (numpy/scipy == what the reference calls)
# ---------------------------------------------------------------------------------------------
def lin_deps_qr(regressor_or_gram, min_tol):
    """``Model.computeRegressorLinDepsQR`` core (model.py:841-894): pivoted QR, rank, Pb, K."""
    import scipy.linalg as sla

    Q, R, P = sla.qr(regressor_or_gram, pivoting=True, mode="economic")
    r = int(np.where(np.abs(R.diagonal()) > min_tol)[0].size)
    n = P.size
    Pp = np.zeros((n, n))
    for i in P:
        Pp[i, P[i]] = 1
    Pb = Pp.T[:, 0:r]
    Pd = Pp.T[:, r:]
    R1 = R[0:r, 0:r]
    R2 = R[0:r, r:]
    lin = sla.inv(R1).dot(R2)
    lin[np.abs(lin) < min_tol] = 0
    K = Pb.T + lin.dot(Pd.T)
    return {"Q": Q, "R": R, "P": P, "r": r, "Pb": Pb, "Pd": Pd, "K": K, "linear_deps": lin,
            "independent_cols": P[0:r]}
