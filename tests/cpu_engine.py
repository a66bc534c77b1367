"""Test-suite stand-in for ``flobaroid_amd._lib.Engine`` on a box without a GPU.

bench.py's launch / sharding / reduction logic (``--gpus N`` re-exec under torchrun, shard ranges, Gram all-reduce, TSQR rank
tree, checks inside the bench) has to be exercised by the CPU test-suite, but the product has no CPU compute path.  The tests
therefore inject this object with ``bench.py --backend gloo --engine cpu_engine:make_engine``: the same method surface, answered
by the CPU oracle (test infrastructure) on torch CPU tensors.  Nothing under flobaroid_amd/ imports it.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle.oracle import OracleModel


def _np(x):
    return x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class OracleEngine:
    def __init__(self, topo, floating=False, device=0, **kw):
        if kw.get("friction"):
            raise NotImplementedError("stand-in: inertial columns only")
        self.topo = topo
        self.om = OracleModel(topo, floating=bool(floating))
        self.rows, self.cols = self.om.rows, self.om.P
        self.n = topo.num_dofs
        self.device = device
        self._prof = {c: [0.0, 0] for c in ("kin", "regressor", "gram", "reduce", "id", "tsqr", "pack", "h2d", "tree")}

    def close(self):
        pass

    def _st(self, st):
        return {("rpy" if k == "base_rpy" else k): np.ascontiguousarray(_np(v)) for k, v in st.items()}

    def _aug(self, st, rhs, w=None):
        Y = self.om.regressor(self._st(st))
        if rhs is not None:
            Y = np.hstack([Y, _np(rhs).reshape(Y.shape[0], -1)])
        if w is not None:
            Y = Y * _np(w).reshape(-1, 1)
        return Y

    def regressor(self, st, out=None):
        Y = torch.from_numpy(self.om.regressor(self._st(st)))
        if out is not None:
            out.copy_(Y)
            return out
        return Y

    def inverse_dynamics(self, st, x_std, vel_sign=None, out=None):
        return torch.from_numpy(self.om.inverse_dynamics(self._st(st), np.asarray(x_std)))

    def gram(self, st, rhs=None, w=None, out=None, accumulate=False):
        A = self._aug(st, rhs, w)
        G = torch.from_numpy(A.T @ A)
        self._prof["gram"][0] += 1.0
        self._prof["gram"][1] += 1
        if out is None:
            return G
        if isinstance(out, np.ndarray):
            out[...] = (out if accumulate else 0) + G.numpy()
            return out
        if accumulate:
            out += G
        else:
            out.copy_(G)
        return out

    def tsqr(self, st, rhs=None, w=None, R_in=None, out=None, cols=None):
        A = self._aug(st, rhs, w)
        if cols is not None:
            k = A.shape[1] - self.cols
            A = A[:, list(np.asarray(cols)) + list(range(self.cols, self.cols + k))]
        if R_in is not None:
            A = np.vstack([_np(R_in), A])
        R = np.linalg.qr(A, mode="r") if A.shape[0] >= A.shape[1] else np.linalg.qr(np.vstack([A, np.zeros((A.shape[1] - A.shape[0], A.shape[1]))]), mode="r")
        return torch.from_numpy(np.triu(R))

    def tsqr_merge(self, R_a, R_b, out=None):
        return torch.from_numpy(np.triu(np.linalg.qr(np.vstack([_np(R_a), _np(R_b)]), mode="r")))

    def tsqr_work_info(self, num_samples, k=0, cols=None):
        n = ((self.cols if cols is None else len(cols)) + k + 15) // 16 * 16
        return {"mfma_level0": 0, "mfma_tree": 0, "block_rows": 64, "n_padded": n, "flop": 0}

    def gram_program_info(self, k=0):
        return {"tiles": 0, "pairs": 0, "mfma_per_sample": 0, "parts": 0}

    def profile_enable(self, on=True):
        pass

    def profile_get(self):
        p = {c: (v[0], v[1]) for c, v in self._prof.items()}
        for v in self._prof.values():
            v[0], v[1] = 0.0, 0
        return p


class NumpyOracleEngine:
    """The same stand-in for host-side callers (``Model`` / ``Data`` with NumPy arrays): NumPy in, NumPy out, friction layouts included.
    Used by the drop-in test that runs the REFERENCE'S identifier.py against the work-alike classes in the build container."""

    def __init__(self, topo, floating=False, friction=False, friction_symmetric=True, gravity_only=False, stribeck_velocity=0.0, **kw):
        self.topo = topo
        self.om = OracleModel(topo, floating=bool(floating), fric=bool(friction), fric_sym=bool(friction_symmetric), grav_only=bool(gravity_only),
                              stribeck=float(stribeck_velocity))
        self.rows, self.cols, self.n = self.om.rows, self.om.P, topo.num_dofs
        self.friction = bool(friction)

    def _st(self, st):
        return {("rpy" if k == "base_rpy" else k): np.ascontiguousarray(v, dtype=np.float64) for k, v in st.items() if k != "sign"}

    def regressor(self, st, out=None):
        return self.om.regressor(self._st(st), st.get("sign"))

    def inverse_dynamics(self, st, x_std, vel_sign=None, out=None):
        return self.om.inverse_dynamics(self._st(st), np.asarray(x_std), st.get("sign"), vel_sign)

    def predict(self, st, x, out=None):
        return (self.regressor(st) @ np.asarray(x)).reshape(-1, self.rows)

    def contact_torques(self, st, frame, wrench, out=None):
        s2 = self._st(st)
        S = s2["q"].shape[0]
        s2.setdefault("dq", np.zeros_like(s2["q"]))
        s2.setdefault("ddq", np.zeros_like(s2["q"]))
        if self.om.floating:
            s2.setdefault("base_vel", np.zeros((S, 6)))
            s2.setdefault("base_acc", np.zeros((S, 6)))
        return self.om.contact_torques(s2, frame, np.asarray(wrench))

    def gram(self, st, rhs=None, w=None, out=None, accumulate=False):
        A = self.regressor(st)
        if rhs is not None:
            A = np.hstack([A, np.asarray(rhs).reshape(A.shape[0], -1)])
        if w is not None:
            A = A * np.asarray(w).reshape(-1, 1)
        return A.T @ A

    def tsqr(self, st, rhs=None, w=None, R_in=None, out=None, cols=None):
        A = self.regressor(st)
        if cols is not None:
            A = A[:, np.asarray(cols)]
        if rhs is not None:
            A = np.hstack([A, np.asarray(rhs).reshape(A.shape[0], -1)])
        if w is not None:
            A = A * np.asarray(w).reshape(-1, 1)
        if R_in is not None:
            A = np.vstack([np.asarray(R_in), A])
        return np.triu(np.linalg.qr(A, mode="r"))


def make_engine(topo, floating=False, device=0, **kw):
    return OracleEngine(topo, floating=floating, device=device, **kw)
