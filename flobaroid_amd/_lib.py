"""Thin ctypes layer over the C-ABI in ``include/fbr.h`` (``libfbr.so``, HIP, gfx950).

There is no CPU fallback: if the shared library is missing or no HIP device is usable the calls
raise.  Arrays may be NumPy (host) or torch CUDA tensors (device, float64, contiguous); all state
arrays of one call must live in the same memory space.
"""
from __future__ import annotations

import ctypes
import os
from typing import Any

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FBR_LIB_PATH") or os.path.join(_HERE, "libfbr.so")  # (FBR_LIB_PATH: kernel-shape experiments, tools/)

FBR_HOST = 0
FBR_DEVICE = 1

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


class FbrError(RuntimeError):
    pass


class fbr_topology(ctypes.Structure):
    _fields_ = [
        ("num_links", ctypes.c_int32),
        ("num_dofs", ctypes.c_int32),
        ("parent", _ip),
        ("dof_index", _ip),
        ("rest_R", _dp),
        ("rest_p", _dp),
        ("axis", _dp),
        ("floating_base", ctypes.c_int32),
        ("gravity", ctypes.c_double * 3),
        ("friction", ctypes.c_int32),
        ("friction_symmetric", ctypes.c_int32),
        ("gravity_only", ctypes.c_int32),
        ("stribeck_velocity", ctypes.c_double),
        ("joint_type", _ip),
    ]


class fbr_states(ctypes.Structure):
    _fields_ = [
        ("num_samples", ctypes.c_int64),
        ("mem", ctypes.c_int32),
        ("q", ctypes.c_void_p),
        ("dq", ctypes.c_void_p),
        ("ddq", ctypes.c_void_p),
        ("base_vel", ctypes.c_void_p),
        ("base_acc", ctypes.c_void_p),
        ("base_rpy", ctypes.c_void_p),
        ("sign", ctypes.c_void_p),
    ]


_lib = None

# name -> (restype, argtypes); also the list of symbols tests check against include/fbr.h
_SIGNATURES = {
    "fbr_version": (ctypes.c_int, []),
    "fbr_device_count": (ctypes.c_int, []),
    "fbr_last_error": (ctypes.c_char_p, []),
    "fbr_model_create": (ctypes.c_int, [ctypes.POINTER(fbr_topology), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "fbr_model_destroy": (None, [ctypes.c_void_p]),
    "fbr_model_dims": (ctypes.c_int, [ctypes.c_void_p, _ip, _ip]),
    "fbr_model_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fbr_regressor_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_void_p, ctypes.c_int32]),
    "fbr_inverse_dynamics_batch": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), _dp, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32],
    ),
    "fbr_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(fbr_states), _dp, ctypes.c_void_p, ctypes.c_int32]),
    "fbr_contact_torques": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_int32, _dp, _dp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32],
    ),
    "fbr_gram_accumulate": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_int32, ctypes.c_int32],
    ),
    "fbr_gram_submit": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)],
    ),
    "fbr_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "fbr_gram_grouped": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_int32],
    ),
    "fbr_fd_scores": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_int32],
    ),
    "fbr_fourier_states": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, _dp, _dp, _dp, _dp, _dp, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_int32],
    ),
    "fbr_tsqr": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_int32],
    ),
    "fbr_tsqr_cols": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), _ip, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32],
    ),
    "fbr_tsqr_submit": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(fbr_states), _ip, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
    ),
    "fbr_tsqr_merge": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "fbr_tsqr_work_info": (
        ctypes.c_int,
        [ctypes.c_void_p, _ip, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
         ctypes.POINTER(ctypes.c_int64), _ip, _ip],
    ),
    "fbr_filtfilt": (ctypes.c_int, [ctypes.c_void_p, _dp, _dp, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "fbr_medfilt": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "fbr_central_diff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]),
    "fbr_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "fbr_profile_get": (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.POINTER(ctypes.c_int64)]),
    "fbr_gram_program_info": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, _ip, _ip, ctypes.POINTER(ctypes.c_int64), _ip],
    ),
    "fbr_gram_lane_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "fbr_model_link_merge_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _ip, _ip]),
    "fbr_model_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]),
    "fbr_model_get_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]),
    "fbr_model_option_name": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p)]),
}

# Options every new Engine starts with (``fbr_model_set_option``, include/fbr.h lists the keys), on top of the library's defaults and below
# the ``options`` argument of the constructor.  A plain Python dict: the library itself never reads the process environment.  The test
# suite uses it to run whole modules with the column reductions forced / switched off (tests/conftest.py: reduction_mode).
FBR_VERSION = 102  # include/fbr.h FBR_VERSION: the C-ABI these ctypes signatures describe
DEFAULT_OPTIONS: dict = {}


def load_library():
    """Load libfbr.so (fails loudly; never substitutes a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FbrError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  flobaroid_amd has no CPU fallback."
        )
    # torch ships its own libamdhip64; if libfbr pulled in /opt/rocm's copy first, a later torch.cuda
    # initialisation in the same process would see "No HIP GPUs".  Load torch's runtime first when torch
    # is installed so both share one HIP runtime (torch is plumbing here, not a dependency of the C-ABI).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    got = int(lib.fbr_version())
    if got != FBR_VERSION:
        raise FbrError(f"{LIB_PATH} reports C-ABI version {got}, this binding was written for {FBR_VERSION} (include/fbr.h FBR_VERSION): "
                       "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'` -- signatures have changed between the two")
    _lib = lib
    return lib


def device_count() -> int:
    return int(load_library().fbr_device_count())


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().fbr_last_error()
        raise FbrError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def _is_torch(x: Any) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class _Ref:
    """Pointer + memory space of one array argument (keeps the backing object alive)."""

    def __init__(self, x: Any, shape: tuple | None = None, name: str = "array"):
        self.obj = None
        self.ptr = None
        self.mem = None
        if x is None:
            return
        if _is_torch(x):
            import torch

            if x.dtype != torch.float64:
                raise TypeError(f"{name}: torch tensors must be float64")
            if x.device.type == "cuda":
                t = x.contiguous()
                self.obj, self.ptr, self.mem = t, t.data_ptr(), FBR_DEVICE
                got = tuple(t.shape)
            else:
                a = np.ascontiguousarray(x.numpy(), dtype=np.float64)
                self.obj, self.ptr, self.mem = a, a.ctypes.data, FBR_HOST
                got = a.shape
        else:
            a = np.ascontiguousarray(x, dtype=np.float64)
            self.obj, self.ptr, self.mem = a, a.ctypes.data, FBR_HOST
            got = a.shape
        if shape is not None and tuple(got) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(got)}")


def _same_space(refs: list[_Ref]) -> int:
    mems = {r.mem for r in refs if r.mem is not None}
    if len(mems) > 1:
        raise ValueError("all arrays of one call must live in the same memory space (all NumPy or all CUDA tensors)")
    return mems.pop() if mems else FBR_HOST


class Engine:
    """One ``fbr_model`` handle: device tables of a robot + identified-column layout."""

    def __init__(self, topo, floating=False, friction=False, friction_symmetric=True, gravity_only=False,
                 stribeck_velocity=0.0, gravity=(0.0, 0.0, -9.81), device: int = 0, options: dict | None = None):
        lib = load_library()
        self._lib = lib
        self.topo = topo
        self._parent = np.array(topo.parent, dtype=np.int32)
        self._dof = np.array(topo.dof_index, dtype=np.int32)
        self._restR = np.ascontiguousarray(topo.rest_R, dtype=np.float64).reshape(-1)
        self._restp = np.ascontiguousarray(topo.rest_p, dtype=np.float64).reshape(-1)
        self._axis = np.ascontiguousarray(topo.axis, dtype=np.float64).reshape(-1)
        t = fbr_topology()
        t.num_links = topo.num_links
        t.num_dofs = topo.num_dofs
        t.parent = self._parent.ctypes.data_as(_ip)
        t.dof_index = self._dof.ctypes.data_as(_ip)
        t.rest_R = self._restR.ctypes.data_as(_dp)
        t.rest_p = self._restp.ctypes.data_as(_dp)
        t.axis = self._axis.ctypes.data_as(_dp)
        t.floating_base = int(bool(floating))
        t.gravity = (ctypes.c_double * 3)(*[float(g) for g in gravity])
        t.friction = int(bool(friction))
        t.friction_symmetric = int(bool(friction_symmetric))
        t.gravity_only = int(bool(gravity_only))
        t.stribeck_velocity = float(stribeck_velocity)
        self._jtype = np.array(topo.joint_type, dtype=np.int32)  # 0 fixed, 1 revolute, 2 prismatic
        t.joint_type = self._jtype.ctypes.data_as(_ip)
        h = ctypes.c_void_p()
        _check(lib.fbr_model_create(ctypes.byref(t), int(device), ctypes.byref(h)), "fbr_model_create")
        self._h = h
        r, c = ctypes.c_int32(), ctypes.c_int32()
        _check(lib.fbr_model_dims(h, ctypes.byref(r), ctypes.byref(c)), "fbr_model_dims")
        self.rows, self.cols = int(r.value), int(c.value)
        self.n = topo.num_dofs
        self.L = topo.num_links
        self.floating = bool(floating)
        self.friction = bool(friction)
        self.device = int(device)
        for key, val in {**DEFAULT_OPTIONS, **(options or {})}.items():
            self.set_option(key, val)

    # ------------------------------------------------------------------ options (fbr_model_set_option)
    def set_option(self, key: str, value: float) -> None:
        _check(self._lib.fbr_model_set_option(self._h, key.encode(), float(value)), f"fbr_model_set_option({key})")

    def get_option(self, key: str) -> float:
        v = ctypes.c_double()
        _check(self._lib.fbr_model_get_option(self._h, key.encode(), ctypes.byref(v)), f"fbr_model_get_option({key})")
        return float(v.value)

    def options(self) -> dict:
        """Every option of the handle with its current value."""
        out, i, name = {}, 0, ctypes.c_char_p()
        while self._lib.fbr_model_option_name(i, ctypes.byref(name)) == 0:
            out[name.value.decode()] = self.get_option(name.value.decode())
            i += 1
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.fbr_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def set_stream(self, stream_ptr: int | None) -> None:
        _check(self._lib.fbr_model_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)), "fbr_model_set_stream")
        self._stream_handle = int(stream_ptr or 0)  # 0: the model's own stream

    def use_torch_stream(self) -> None:
        """Run on torch's current stream when that is a real stream object.  torch's DEFAULT stream is the null stream (handle 0),
        which the C-ABI reads as "the model's own stream": the engine then keeps its own non-blocking stream and every call with
        CUDA tensors first waits for torch's stream (``_sync_torch``), so inputs produced by torch kernels are complete."""
        import torch

        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def _sync_torch(self) -> None:
        # calls are blocking and the library synchronises its own stream on return; what is missing is the other direction:
        # torch kernels still writing the inputs on a stream the library does not run on.  Checked at EVERY call: the caller may have
        # switched torch streams (or called set_stream) since use_torch_stream().
        import torch

        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == 0 or cur.cuda_stream != getattr(self, "_stream_handle", 0):
            cur.synchronize()

    def _states(self, st: dict, need_vel: bool = True):
        q = _Ref(st["q"], name="q")
        S = q.obj.shape[0]
        n = self.n
        if tuple(q.obj.shape) != (S, n):
            raise ValueError(f"q: expected (S,{n}), got {tuple(q.obj.shape)}")
        refs = [q]
        dq = _Ref(st.get("dq"), (S, n), "dq") if need_vel or st.get("dq") is not None else _Ref(None)
        ddq = _Ref(st.get("ddq"), (S, n), "ddq") if need_vel or st.get("ddq") is not None else _Ref(None)
        bv = ba = rpy = sg = _Ref(None)
        if self.floating:
            rpy = _Ref(st.get("rpy", st.get("base_rpy")), (S, 3), "base_rpy")
            if need_vel:
                bv = _Ref(st.get("base_vel", st.get("base_velocity")), (S, 6), "base_vel")
                ba = _Ref(st.get("base_acc", st.get("base_acceleration")), (S, 6), "base_acc")
        if self.friction and need_vel:
            sg = _Ref(st.get("sign"), (S, n), "sign")
        refs += [dq, ddq, bv, ba, rpy, sg]
        mem = _same_space(refs)
        if mem == FBR_DEVICE:
            self._sync_torch()
        s = fbr_states()
        s.num_samples = S
        s.mem = mem
        s.q, s.dq, s.ddq = q.ptr, dq.ptr, ddq.ptr
        s.base_vel, s.base_acc, s.base_rpy, s.sign = bv.ptr, ba.ptr, rpy.ptr, sg.ptr
        return s, refs, S, mem

    def _out(self, out, shape, mem):
        if out is not None:
            if _is_torch(out):
                import torch

                if out.dtype != torch.float64 or not out.is_contiguous():
                    raise ValueError("out must be a contiguous float64 tensor")
            elif not (isinstance(out, np.ndarray) and out.flags.c_contiguous and out.dtype == np.float64):
                raise ValueError("out must be a C-contiguous float64 ndarray")
            return _Ref(out, shape, "out"), out
        if mem == FBR_DEVICE:
            import torch

            t = torch.empty(shape, dtype=torch.float64, device=f"cuda:{self.device}")
            return _Ref(t), t
        a = np.empty(shape, dtype=np.float64)
        return _Ref(a), a

    # ------------------------------------------------------------------ C-ABI calls
    def regressor(self, st: dict, out=None):
        """Stacked standard regressor (S*rows, cols)."""
        s, keep, S, mem = self._states(st)
        r, ret = self._out(out, (S * self.rows, self.cols), mem)
        _check(self._lib.fbr_regressor_batch(self._h, ctypes.byref(s), r.ptr, r.mem), "fbr_regressor_batch")
        return ret

    def inverse_dynamics(self, st: dict, x_std, vel_sign=None, out=None):
        s, keep, S, mem = self._states(st)
        x = np.ascontiguousarray(x_std, dtype=np.float64)
        vs = _Ref(vel_sign, (S, self.n), "vel_sign") if vel_sign is not None else _Ref(None)
        if vs.mem is not None and vs.mem != mem:
            raise ValueError("vel_sign must live in the same memory space as the states")
        r, ret = self._out(out, (S, self.rows), mem)
        _check(
            self._lib.fbr_inverse_dynamics_batch(self._h, ctypes.byref(s), x.ctypes.data_as(_dp), int(x.size), vs.ptr,
                                                 r.ptr, r.mem),
            "fbr_inverse_dynamics_batch",
        )
        return ret

    def predict(self, st: dict, x, out=None):
        s, keep, S, mem = self._states(st)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.size != self.cols:
            raise ValueError(f"x must have {self.cols} entries")
        r, ret = self._out(out, (S, self.rows), mem)
        _check(self._lib.fbr_predict(self._h, ctypes.byref(s), x.ctypes.data_as(_dp), r.ptr, r.mem), "fbr_predict")
        return ret

    def contact_torques(self, st: dict, frame: str, wrench, out=None):
        t = self.topo
        if frame in t.frames:
            link, fR, fp = t.frames[frame]["link"], t.frames[frame]["R"], t.frames[frame]["p"]
        elif frame in t.link_names:
            link, fR, fp = t.link_names.index(frame), np.eye(3), np.zeros(3)
        else:
            raise KeyError(f"unknown frame '{frame}'")
        s, keep, S, mem = self._states(st, need_vel=False)
        w = _Ref(wrench, (S, 6), "wrench")
        if w.mem != mem:
            raise ValueError("wrench must live in the same memory space as the states")
        fR = np.ascontiguousarray(fR, dtype=np.float64).reshape(-1)
        fp = np.ascontiguousarray(fp, dtype=np.float64).reshape(-1)
        r, ret = self._out(out, (S, self.rows), mem)
        _check(
            self._lib.fbr_contact_torques(self._h, ctypes.byref(s), int(link), fR.ctypes.data_as(_dp),
                                          fp.ctypes.data_as(_dp), w.ptr, r.ptr, r.mem),
            "fbr_contact_torques",
        )
        return ret

    def _rhs(self, rhs, w, S, mem):
        k = 0
        rr = _Ref(None)
        if rhs is not None:
            if not _is_torch(rhs):
                rhs = np.asarray(rhs, dtype=np.float64)
            ncol = int(rhs.shape[-1]) if rhs.ndim >= 2 else 1  # (an empty batch cannot infer the column count by reshape)
            rhs2 = rhs.reshape(S * self.rows, ncol)
            k = int(rhs2.shape[1])
            rr = _Ref(rhs2, (S * self.rows, k), "rhs")
            if rr.mem != mem:
                raise ValueError("rhs must live in the same memory space as the states")
        wr = _Ref(None)
        if w is not None:
            w2 = w.reshape(S * self.rows) if _is_torch(w) else np.asarray(w, dtype=np.float64).reshape(S * self.rows)
            wr = _Ref(w2, (S * self.rows,), "w")
            if wr.mem != mem:
                raise ValueError("w must live in the same memory space as the states")
        return rr, wr, k

    def gram(self, st: dict, rhs=None, w=None, out=None, accumulate: bool = False):
        """Raw Gram [Y|rhs]^T diag(w)^2 [Y|rhs], shape (cols+k, cols+k)."""
        s, keep, S, mem = self._states(st)
        rr, wr, k = self._rhs(rhs, w, S, mem)
        Pa = self.cols + k
        r, ret = self._out(out, (Pa, Pa), mem)
        _check(
            self._lib.fbr_gram_accumulate(self._h, ctypes.byref(s), rr.ptr, k, wr.ptr, r.ptr, r.mem, int(bool(accumulate))),
            "fbr_gram_accumulate",
        )
        return ret

    def gram_submit(self, st: dict, out, rhs=None, w=None, accumulate: bool = False) -> int:
        """``gram`` without waiting (``fbr_gram_submit``): ``out`` is a CUDA tensor, the inputs CUDA tensors or PINNED host tensors
        (staged chunk by chunk on a copy stream); result in ``out`` after ``wait(ticket)``.  The caller keeps ``st`` / ``rhs`` / ``w`` /
        ``out`` alive and untouched until then; at most two submissions are in flight."""
        s, keep, S, mem = self._states(st)
        rr, wr, k = self._rhs(rhs, w, S, mem)
        Pa = self.cols + k
        r, _ = self._out(out, (Pa, Pa), mem)
        if r.mem != FBR_DEVICE:
            raise ValueError("gram_submit needs a CUDA tensor for the output")
        t = ctypes.c_int64(-1)
        _check(self._lib.fbr_gram_submit(self._h, ctypes.byref(s), rr.ptr, k, wr.ptr, r.ptr, int(bool(accumulate)), ctypes.byref(t)),
               "fbr_gram_submit")
        # the GPU reads the arrays handed to the library until wait(): contiguous copies / reshapes made on the way (``_Ref``, ``_rhs``)
        # would otherwise go back to torch's allocator on return and could be reused while still being read
        self._keep_inflight(int(t.value), (keep, rr, wr, r))
        return int(t.value)

    def _keep_inflight(self, ticket: int, refs) -> None:
        infl = self.__dict__.setdefault("_inflight", {})
        infl[ticket] = refs
        for old in [k_ for k_ in infl if k_ < ticket - 1]:  # (a third submission has waited for the oldest inside the library)
            del infl[old]

    def wait(self, ticket: int = -1) -> None:
        """Block until the submission ``ticket`` (default: everything submitted) is complete."""
        try:
            _check(self._lib.fbr_wait(self._h, int(ticket)), "fbr_wait")
        finally:
            infl = self.__dict__.get("_inflight", {})
            for k_ in [k_ for k_ in infl if ticket < 0 or k_ <= ticket]:
                del infl[k_]

    def tsqr_submit(self, st: dict, out, rhs=None, w=None, R_in=None, cols=None) -> int:
        """``tsqr`` without waiting (``fbr_tsqr_submit``): CUDA tensors only; result in ``out`` after ``wait(ticket)``.  At most two
        submissions (of either kind) are in flight; consecutive TSQR submissions overlap the next one's kinematics / first regressor
        chunk with the merge trees of the one before."""
        s, keep, S, mem = self._states(st)
        rr, wr, k = self._rhs(rhs, w, S, mem)
        ca = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        Pa = (self.cols if ca is None else int(ca.size)) + k
        r, _ = self._out(out, (Pa, Pa), mem)
        rin = _Ref(R_in, (Pa, Pa), "R_in") if R_in is not None else _Ref(None)
        if mem != FBR_DEVICE or r.mem != FBR_DEVICE or (rin.mem is not None and rin.mem != FBR_DEVICE):
            raise ValueError("tsqr_submit needs CUDA tensors for the states, rhs, weights, R_in and the output")
        t = ctypes.c_int64(-1)
        _check(self._lib.fbr_tsqr_submit(self._h, ctypes.byref(s), None if ca is None else ca.ctypes.data_as(_ip), 0 if ca is None else int(ca.size),
                                         rr.ptr, k, wr.ptr, rin.ptr, r.ptr, ctypes.byref(t)), "fbr_tsqr_submit")
        self._keep_inflight(int(t.value), (keep, rr, wr, rin, r, ca))
        return int(t.value)

    def gram_grouped(self, st: dict, ngroups: int, rhs=None, w=None, out=None):
        """One raw Gram per group of S / ngroups consecutive samples, shape (ngroups, cols+k, cols+k), in one pass."""
        s, keep, S, mem = self._states(st)
        if ngroups < 1 or S % ngroups:
            raise ValueError("the number of samples must be a multiple of ngroups")
        rr, wr, k = self._rhs(rhs, w, S, mem)
        Pa = self.cols + k
        r, ret = self._out(out, (int(ngroups), Pa, Pa), mem)
        _check(self._lib.fbr_gram_grouped(self._h, ctypes.byref(s), int(ngroups), rr.ptr, k, wr.ptr, r.ptr, r.mem), "fbr_gram_grouped")
        return ret

    def fd_scores(self, st: dict, W, eps: float, out=None):
        """Weighted regressor scores sum(W_s * Y) of every sample's baseline state and its 3n states perturbed by +eps
        in q_d, dq_d, ddq_d: shape (S, 1 + 3n) (analyticalGradient.py:92-185 without the per-sample Python loop)."""
        s, keep, S, mem = self._states(st)
        Wr = _Ref(W if _is_torch(W) else np.asarray(W, dtype=np.float64).reshape(S * self.rows, self.cols), (S * self.rows, self.cols), "W")
        if Wr.mem != mem:
            raise ValueError("W must live in the same memory space as the states")
        r, ret = self._out(out, (S, 1 + 3 * self.topo.num_dofs), mem)
        _check(self._lib.fbr_fd_scores(self._h, ctypes.byref(s), Wr.ptr, float(eps), r.ptr, r.mem), "fbr_fd_scores")
        return ret

    def fourier_states(self, wf, a, b, q_offset, T: int, freq: float, q_range=None, device: bool = True) -> dict:
        """q, dq, ddq of C candidate trajectories x T samples from their Fourier coefficients (``fbr_fourier_states``): ``wf`` (C,),
        ``a`` / ``b`` (C, n, nharm), ``q_offset`` (C, n), ``q_range`` (C, n) or None.  Returns {"q", "dq", "ddq"} of shape (C * T, n) --
        CUDA tensors (``device=True``: they stay in HBM for ``gram_grouped``) or NumPy arrays."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        C, n, nh = a.shape
        if n != self.n or b.shape != a.shape:
            raise ValueError(f"a / b: expected (C, {self.n}, nharm)")
        wf = np.ascontiguousarray(np.broadcast_to(np.asarray(wf, dtype=np.float64), (C,)))
        qo = np.ascontiguousarray(q_offset, dtype=np.float64).reshape(C, n)
        qr = None if q_range is None else np.ascontiguousarray(q_range, dtype=np.float64).reshape(C, n)
        outs, refs = [], []
        for _ in range(3):
            r, ret = self._out(None, (C * int(T), n), FBR_DEVICE if device else FBR_HOST)
            outs.append(ret)
            refs.append(r)
        _check(self._lib.fbr_fourier_states(self._h, C, int(T), nh, float(freq), wf.ctypes.data_as(_dp), a.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                            qo.ctypes.data_as(_dp), None if qr is None else qr.ctypes.data_as(_dp), refs[0].ptr, refs[1].ptr, refs[2].ptr,
                                            refs[0].mem), "fbr_fourier_states")
        return {"q": outs[0], "dq": outs[1], "ddq": outs[2]}

    def tsqr(self, st: dict, rhs=None, w=None, R_in=None, out=None, cols=None):
        """Upper-triangular R with R^T R = [Y[:, cols]|rhs]^T [Y[:, cols]|rhs] (blocked Householder TSQR);
        ``cols=None`` takes every identified column."""
        s, keep, S, mem = self._states(st)
        rr, wr, k = self._rhs(rhs, w, S, mem)
        ca = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        Pa = (self.cols if ca is None else int(ca.size)) + k
        r, ret = self._out(out, (Pa, Pa), mem)
        rin = _Ref(R_in, (Pa, Pa), "R_in") if R_in is not None else _Ref(None)
        if rin.mem is not None and rin.mem != r.mem:
            raise ValueError("R_in must live in the same memory space as the output")
        if ca is None:
            _check(self._lib.fbr_tsqr(self._h, ctypes.byref(s), rr.ptr, k, wr.ptr, rin.ptr, r.ptr, r.mem), "fbr_tsqr")
        else:
            _check(self._lib.fbr_tsqr_cols(self._h, ctypes.byref(s), ca.ctypes.data_as(_ip), int(ca.size), rr.ptr, k, wr.ptr,
                                           rin.ptr, r.ptr, r.mem), "fbr_tsqr_cols")
        return ret

    def tsqr_merge(self, R_a, R_b, out=None):
        a = _Ref(R_a, name="R_a")
        n = a.obj.shape[0]
        b = _Ref(R_b, (n, n), "R_b")
        if a.mem != b.mem:
            raise ValueError("R_a and R_b must live in the same memory space")
        if a.mem == FBR_DEVICE:
            self._sync_torch()  # (a factor just received by torch.distributed lands on torch's stream)
        r, ret = self._out(out, (n, n), a.mem)
        _check(self._lib.fbr_tsqr_merge(self._h, n, a.ptr, b.ptr, r.ptr, a.mem), "fbr_tsqr_merge")
        return ret

    # ------------------------------------------------------------------ signal conditioning (Data.preprocess on the device)
    def _sig_array(self, X, ncols):
        """(pointer, mem, S, ld, keep-alive) of a 2-D C-contiguous float64 array (NumPy, changed in place, or CUDA tensor)."""
        if _is_torch(X):
            if X.dim() != 2 or not X.is_contiguous() or str(X.dtype) != "torch.float64" or X.device.type != "cuda":
                raise ValueError("expected a contiguous 2-D float64 CUDA tensor")
            self._sync_torch()
            return X.data_ptr(), FBR_DEVICE, int(X.shape[0]), int(X.shape[1]), X
        if not (isinstance(X, np.ndarray) and X.ndim == 2 and X.flags.c_contiguous and X.dtype == np.float64):
            raise ValueError("expected a C-contiguous 2-D float64 ndarray (it is filtered in place)")
        return X.ctypes.data, FBR_HOST, int(X.shape[0]), int(X.shape[1]), X

    def filtfilt(self, b, a, X, ncols=None):
        """In place ``X[:, :ncols] = scipy.signal.filtfilt(b, a, X[:, :ncols], axis=0)`` (zero-phase low-pass of every channel)."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        a = np.ascontiguousarray(a, dtype=np.float64)
        if b.size != a.size:  # scipy pads the shorter one with zeros
            n = max(b.size, a.size)
            b, a = np.r_[b, np.zeros(n - b.size)], np.r_[a, np.zeros(n - a.size)]
        ptr, mem, S, ld, keep = self._sig_array(X, ncols)
        nc = ld if ncols is None else int(ncols)
        _check(self._lib.fbr_filtfilt(self._h, b.ctypes.data_as(_dp), a.ctypes.data_as(_dp), int(b.size), ptr, S, nc, ld, mem), "fbr_filtfilt")
        return X

    def medfilt(self, k: int, X, ncols=None):
        """In place ``X[:, :ncols] = scipy.signal.medfilt(X[:, :ncols], (k, 1))``."""
        ptr, mem, S, ld, keep = self._sig_array(X, ncols)
        nc = ld if ncols is None else int(ncols)
        _check(self._lib.fbr_medfilt(self._h, int(k), ptr, S, nc, ld, mem), "fbr_medfilt")
        return X

    def central_diff(self, A, times, out=None):
        """4th-order central difference of the columns of A over the time stamps (data.py:396-418)."""
        ptr, mem, S, ld, keep = self._sig_array(A, None)
        t = _Ref(times, (S,), "times")
        if t.mem != mem:
            raise ValueError("times must live in the same memory space as A")
        r, ret = self._out(out, (S, ld), mem)
        _check(self._lib.fbr_central_diff(self._h, ptr, t.ptr, r.ptr, S, ld, mem), "fbr_central_diff")
        return ret

    def tsqr_work_info(self, num_samples: int, k: int = 0, cols=None) -> dict:
        """Executed MFMA instructions of ``tsqr(..)`` for ``num_samples`` samples (level-0 folds, merge tree), 2 * 16 * 16 * 4 = 2048 flop each."""
        ca = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        l0, tr = ctypes.c_int64(), ctypes.c_int64()
        mb, npad = ctypes.c_int32(), ctypes.c_int32()
        _check(
            self._lib.fbr_tsqr_work_info(self._h, None if ca is None else ca.ctypes.data_as(_ip), 0 if ca is None else int(ca.size), int(k),
                                         int(num_samples), ctypes.byref(l0), ctypes.byref(tr), ctypes.byref(mb), ctypes.byref(npad)),
            "fbr_tsqr_work_info",
        )
        return {"mfma_level0": l0.value, "mfma_tree": tr.value, "block_rows": mb.value, "n_padded": npad.value,
                "flop": 2048 * (l0.value + tr.value)}

    PROF_CLASSES = ("kin", "regressor", "gram", "reduce", "id", "tsqr", "pack", "h2d", "tree")

    def profile_enable(self, on: bool = True) -> None:
        _check(self._lib.fbr_profile_enable(self._h, int(bool(on))), "fbr_profile_enable")

    def profile_get(self) -> dict:
        """{class: (device ms, launches)} since the last call (resets the counters)."""
        n = len(self.PROF_CLASSES)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        _check(self._lib.fbr_profile_get(self._h, ms, cnt), "fbr_profile_get")
        return {c: (float(ms[i]), int(cnt[i])) for i, c in enumerate(self.PROF_CLASSES)}

    def link_merge_info(self, num_samples: int = -1) -> dict:
        """What a Gram pass over ``num_samples`` samples runs on (-1: a batch large enough for the column reductions): the moving bodies
        (links attached by fixed joints are merged into the body they ride on and the result expanded, fbr.h fbr_model_link_merge_info)."""
        ml, rc = ctypes.c_int32(), ctypes.c_int32()
        _check(self._lib.fbr_model_link_merge_info(self._h, int(num_samples), ctypes.byref(ml), ctypes.byref(rc)), "fbr_model_link_merge_info")
        return {"moving_links": ml.value, "reduced_cols": rc.value, "links": self.topo.num_links, "cols": self.cols}

    def gram_program_info(self, k: int = 0, num_samples: int = -1) -> dict:
        """Tile program ``gram(..)`` executes for a batch of ``num_samples`` samples (-1: large batches)."""
        nt, npairs, parts = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        mf = ctypes.c_int64()
        _check(
            self._lib.fbr_gram_program_info(self._h, int(k), int(num_samples), ctypes.byref(nt), ctypes.byref(npairs), ctypes.byref(mf),
                                            ctypes.byref(parts)),
            "fbr_gram_program_info",
        )
        return {"tiles": nt.value, "pairs": npairs.value, "mfma_per_sample": mf.value, "parts": parts.value}

    def gram_lane_info(self, k: int = 0, num_samples: int = -1) -> dict:
        """The sample-contiguous Gram pass (option ``gram_lane``) for such a batch; ``active`` False: the per-sample-image pass runs."""
        a = (ctypes.c_int64 * 12)()
        _check(self._lib.fbr_gram_lane_info(self._h, int(k), int(num_samples), a), "fbr_gram_lane_info")
        keys = ("active", "tile_rows", "block_image_bytes", "mfma_per_block", "levels", "max_slabs", "lds_bytes", "tiles", "force_tiles",
                "busiest_wave_pair_levels", "balanced_pair_levels", "stages")
        d = dict(zip(keys, (int(v) for v in a)))
        d["active"] = bool(d["active"])
        return d
