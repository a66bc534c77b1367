"""ctypes loader for the CPU emulation of the HIP kernels (tests/emul/fbr_emul.cpp) -- test-only."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emul", "fbr_emul.cpp")
_OUT = os.path.join(_HERE, "emul", "_build", "libfbr_emul.so")
_CSRC = os.path.join(_HERE, "..", "flobaroid_amd", "csrc")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


class EmulTopo(ctypes.Structure):
    _fields_ = [("L", ctypes.c_int), ("n", ctypes.c_int), ("parent", _ip), ("dof", _ip), ("restR", _dp),
                ("restp", _dp), ("axis", _dp), ("floating", ctypes.c_int), ("gravity", ctypes.c_double * 3),
                ("fric", ctypes.c_int), ("fric_sym", ctypes.c_int), ("grav_only", ctypes.c_int),
                ("stribeck", ctypes.c_double), ("masks", ctypes.POINTER(ctypes.c_uint16)), ("jtype", _ip)]


def lib():
    global _lib
    if _lib is None:
        deps = [_SRC, os.path.join(_CSRC, "fbr_math.h"), os.path.join(_CSRC, "fbr_program.h"), os.path.join(_CSRC, "fbr_reduce.h"), os.path.join(_CSRC, "fbr_kinid.h"), os.path.join(_CSRC, "fbr_gram64.h")]
        if not os.path.exists(_OUT) or any(os.path.getmtime(d) > os.path.getmtime(_OUT) for d in deps):
            os.makedirs(os.path.dirname(_OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", _OUT, _SRC])
        _lib = ctypes.CDLL(_OUT)
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Emul:
    def __init__(self, topo, floating=False, fric=False, fric_sym=True, grav_only=False, stribeck=0.0, masks=None):
        """topo: anything with num_links, num_dofs, parent, dof_index, rest_R, rest_p, axis; masks: per link, the identified parameters
        (bit p = parameter p), the column masks of the regrouped model (csrc/fbr_program.h FbrHostModel::linkmask)"""
        self.keep = [np.array(topo.parent, dtype=np.int32), np.array(topo.dof_index, dtype=np.int32),
                     np.ascontiguousarray(topo.rest_R, dtype=np.float64).reshape(-1),
                     np.ascontiguousarray(topo.rest_p, dtype=np.float64).reshape(-1),
                     np.ascontiguousarray(topo.axis, dtype=np.float64).reshape(-1),
                     None if masks is None else np.ascontiguousarray(masks, dtype=np.uint16),
                     None if getattr(topo, "joint_type", None) is None else np.array(topo.joint_type, dtype=np.int32)]
        self.opts = dict(floating=floating, fric=fric, fric_sym=fric_sym, grav_only=grav_only, stribeck=stribeck)
        self.t = EmulTopo(topo.num_links, topo.num_dofs, self.keep[0].ctypes.data_as(_ip),
                          self.keep[1].ctypes.data_as(_ip), _d(self.keep[2]), _d(self.keep[3]), _d(self.keep[4]),
                          int(floating), (ctypes.c_double * 3)(0.0, 0.0, -9.81), int(fric), int(fric_sym),
                          int(grav_only), float(stribeck),
                          None if masks is None else self.keep[5].ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                          None if self.keep[6] is None else self.keep[6].ctypes.data_as(_ip))
        self.num_links = topo.num_links
        r, c, rec = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().emul_dims(ctypes.byref(self.t), ctypes.byref(r), ctypes.byref(c), ctypes.byref(rec))
        self.rows, self.cols, self.rec = r.value, c.value, rec.value
        self.n = topo.num_dofs
        self.floating = bool(floating)

    def reduction(self, which):
        """(Emul of the reduced robot, E [Pr x P]) of this model's column reductions -- which = 0: fixed links merged into the bodies they
        ride on; 1: merged and the joint-invariant parameters regrouped (column masks) -- or None when there is nothing to reduce.
        [Y | rhs] = [Y_red | rhs] blockdiag(E, 1): the same code the library runs (csrc/fbr_reduce.h)."""
        import types
        L = self.num_links
        parent, dof = np.zeros(L, np.int32), np.zeros(L, np.int32)
        restR, restp, axis = np.zeros(9 * L), np.zeros(3 * L), np.zeros(3 * L)
        masks = np.zeros(L, np.uint16)
        jtype = np.zeros(L, np.int32)
        masked, Pr = ctypes.c_int(), ctypes.c_int()
        cap = (10 * L + 8 * self.n) * self.cols
        E = np.zeros(cap)
        Lr = lib().emul_reduction(ctypes.byref(self.t), int(which), parent.ctypes.data_as(_ip), dof.ctypes.data_as(_ip), _d(restR), _d(restp),
                                  _d(axis), masks.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), ctypes.byref(masked), ctypes.byref(Pr),
                                  _d(E), ctypes.c_long(cap), jtype.ctypes.data_as(_ip))
        if Lr == 0:
            return None
        assert Lr > 0
        topo = types.SimpleNamespace(num_links=Lr, num_dofs=self.n, parent=parent[:Lr], dof_index=dof[:Lr], rest_R=restR[: 9 * Lr],
                                     rest_p=restp[: 3 * Lr], axis=axis[: 3 * Lr], joint_type=jtype[:Lr])
        red = Emul(topo, masks=masks[:Lr] if masked.value else None, **self.opts)
        assert red.cols == Pr.value
        return red, E[: Pr.value * self.cols].reshape(Pr.value, self.cols).copy()

    def _st(self, st):
        c = lambda k: np.ascontiguousarray(st[k], dtype=np.float64) if (k in st and st[k] is not None) else None
        q = c("q")
        return q.shape[0], q, c("dq"), c("ddq"), c("base_vel"), c("base_acc"), c("rpy")

    def kin(self, st):
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        out = np.zeros((S, self.rec))
        lib().emul_kin(ctypes.byref(self.t), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(out))
        return out

    def regressor(self, st, sign=None):
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        sign = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
        Y = np.zeros((S * self.rows, self.cols))
        lib().emul_regressor(ctypes.byref(self.t), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy),
                             _d(sign), _d(Y))
        return Y

    def inverse_dynamics(self, st, x, sign=None, vel_sign=None, mode=0):
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        sign = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
        vel_sign = None if vel_sign is None else np.ascontiguousarray(vel_sign, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        tau = np.zeros((S, self.rows))
        lib().emul_id(ctypes.byref(self.t), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(sign),
                      _d(vel_sign), _d(x), int(mode), _d(tau))
        return tau

    def fused_inverse_dynamics(self, st, x, sign=None, vel_sign=None, mode=0, flink=0, fp=None):
        """(tau, (nsteps, maxlvl, nslots)) from the emulation of fbr_kinid_kernel (csrc/fbr_kinid.h): the device kernel's own step program
        and lane body, one lane per sample.  Raises if a regressor row is written twice or never."""
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        sign = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
        vel_sign = None if vel_sign is None else np.ascontiguousarray(vel_sign, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        tau = np.zeros((S, self.rows))
        info = (ctypes.c_int * 3)()
        fp = np.zeros(3) if fp is None else np.ascontiguousarray(fp, dtype=np.float64)
        rc = lib().emul_kinid(ctypes.byref(self.t), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(sign),
                              _d(vel_sign), _d(x), int(mode), _d(tau), info, int(flink), _d(fp))
        if rc != 0:
            raise RuntimeError(f"emul_kinid: rc {rc}" + (f" (row {-2 - rc} not written exactly once)" if rc <= -2 else " (tree too deep)"))
        return tau, tuple(info)

    def fused_parts_inverse_dynamics(self, st, x, nparts):
        """(tau, steps per part): the tree cut into ``nparts`` programs (fbr_kinid_build_parts), every link's wrench added by its owner only"""
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        x = np.ascontiguousarray(x, dtype=np.float64)
        tau = np.zeros((S, self.rows))
        steps = (ctypes.c_int * 8)()
        rc = lib().emul_kinid_parts(ctypes.byref(self.t), int(nparts), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(x), _d(tau), steps)
        if rc <= 0:
            raise RuntimeError(f"emul_kinid_parts: rc {rc}")
        return tau, list(steps)[:rc]

    def fused_fd_scores(self, st, W, eps, sign=None):
        """emulation of fbr_kinfd_kernel: scores [S][1 + 3 n] of the finite-difference sweep, one lane per evaluation"""
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        sign = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
        W = np.ascontiguousarray(W, dtype=np.float64)
        out = np.zeros((S, 1 + 3 * self.n))
        rc = lib().emul_kinfd(ctypes.byref(self.t), ctypes.c_long(S), ctypes.c_double(eps), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy),
                              _d(sign), _d(W), _d(out))
        if rc != 0:
            raise RuntimeError(f"emul_kinfd: rc {rc}")
        return out

    def program_info(self, k):
        NT, npairs, T, img, items = (ctypes.c_int() for _ in range(5))
        mfma, uni = ctypes.c_long(), ctypes.c_long()
        lib().emul_program_info(ctypes.byref(self.t), int(k), ctypes.byref(NT), ctypes.byref(npairs), ctypes.byref(mfma),
                                ctypes.byref(T), ctypes.byref(img), ctypes.byref(items), ctypes.byref(uni))
        return dict(NT=NT.value, npairs=npairs.value, mfma=mfma.value, T=T.value, part_image_max=img.value,
                    dma_doubles=items.value, mfma_uniform=uni.value)

    def gram64(self, st, rhs=None, w=None, sign=None):
        """(G, stats) of the pass over sample-contiguous images (csrc/fbr_gram64.h), or None when the model is outside it."""
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        k = 0
        if rhs is not None:
            rhs = np.ascontiguousarray(rhs, dtype=np.float64).reshape(S * self.rows, -1)
            k = rhs.shape[1]
        w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        G = np.zeros((self.cols + k, self.cols + k))
        stats = (ctypes.c_long * 10)()
        sign = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
        rc = lib().emul_gram64(ctypes.byref(self.t), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(sign), _d(rhs), int(k),
                               _d(w), _d(G), stats)
        if rc == -1:
            return None
        assert rc == 0, rc
        keys = ("tile_rows", "mfma_per_block", "levels", "max_slabs", "busiest_wave_pair_levels", "balanced_pair_levels", "parts", "pairs", "force_tiles", "stages")
        return G, dict(zip(keys, (int(v) for v in stats)))

    def gram(self, st, rhs=None, sign=None, w=None):
        S, q, dq, ddq, bv, ba, rpy = self._st(st)
        sign = None if sign is None else np.ascontiguousarray(sign, dtype=np.float64)
        k = 0
        if rhs is not None:
            rhs = np.ascontiguousarray(rhs, dtype=np.float64).reshape(S * self.rows, -1)
            k = rhs.shape[1]
        w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        G = np.zeros((self.cols + k, self.cols + k))
        rc = lib().emul_gram(ctypes.byref(self.t), ctypes.c_long(S), _d(q), _d(dq), _d(ddq), _d(bv), _d(ba), _d(rpy), _d(sign),
                             _d(rhs), int(k), _d(w), _d(G))
        assert rc == 0, rc
        return G
