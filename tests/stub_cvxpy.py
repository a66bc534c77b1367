"""A stand-in for the few pieces of ``cvxpy`` the reference's ``identification/sdp.py`` touches on the path
``SDP.initSDP_LMIs`` -> ``SDP.identifyFeasibleStandardParameters`` (sdp.py:68-290, 450-604) -- TEST INFRASTRUCTURE, used where cvxpy /
CLARABEL are not installed (the build container, the GPU image) to exercise the PLUMBING of ``tools/pin_sdp.py``: that the reference's own,
unmodified function consumes the inputs this repository hands it and that both routes (CPU path: its own ``la.qr(YBase)``; GPU path: the
TSQR factor) present it with the same residual map.  It is not an SDP solver: affine expressions are tracked exactly, the linear matrix
inequalities are recorded and IGNORED, and ``Problem.solve`` returns the minimiser of the Schur block's residual ``|| e(x) ||^2`` (minimum
norm, equality pins honoured).  With real cvxpy the same tool runs the real solve.
"""
from __future__ import annotations

import builtins
import types

import numpy as np

CLARABEL, SCS, MOSEK, CVXOPT, COPT = "CLARABEL", "SCS", "MOSEK", "CVXOPT", "COPT"


class SolverError(Exception):
    pass


error = types.SimpleNamespace(SolverError=SolverError)


class Expression:
    """Affine map of the problem's variables, flattened: value = sum_v A[v] @ v.flat + b, carried with a shape."""

    __array_ufunc__ = None  # NumPy operands defer to the reflected operators below (ndarray @ x, ndarray - x)

    def __init__(self, A: dict, b: np.ndarray, shape: tuple):
        self.A, self.b, self.shape = A, np.asarray(b, dtype=float).reshape(-1), tuple(shape)

    # ---- construction helpers
    @staticmethod
    def const(v) -> "Expression":
        v = np.asarray(v, dtype=float)
        return Expression({}, v.reshape(-1), v.shape)

    @staticmethod
    def lift(v) -> "Expression":
        return v if isinstance(v, Expression) else Expression.const(v)

    @property
    def size(self) -> int:
        return int(self.b.size)

    def _rows(self, fn, shape) -> "Expression":
        return Expression({v: fn(a) for v, a in self.A.items()}, fn(self.b.reshape(-1, 1)).reshape(-1), shape)

    # ---- arithmetic
    def __neg__(self):
        return Expression({v: -a for v, a in self.A.items()}, -self.b, self.shape)

    def __add__(self, o):
        o = Expression.lift(o)
        if o.size != self.size:
            if o.size == 1:
                o = Expression({v: np.repeat(a, self.size, axis=0) for v, a in o.A.items()}, np.repeat(o.b, self.size), self.shape)
            elif self.size == 1:
                return o + self
            else:
                raise ValueError(f"shape mismatch {self.shape} + {o.shape}")
        A = dict(self.A)
        for v, a in o.A.items():
            A[v] = A[v] + a if v in A else a
        return Expression(A, self.b + o.b, self.shape if self.size > 1 or not o.shape else o.shape)

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-Expression.lift(o))

    def __rsub__(self, o):
        return Expression.lift(o) + (-self)

    def __mul__(self, c):
        if isinstance(c, Expression):
            if c.A:
                raise NotImplementedError("products of variables")
            c = c.b.reshape(c.shape) if c.shape else float(c.b[0])
        c = np.asarray(c, dtype=float)
        if c.size != 1:
            raise NotImplementedError("elementwise products")
        return Expression({v: a * float(c) for v, a in self.A.items()}, self.b * float(c), self.shape)

    __rmul__ = __mul__

    def __truediv__(self, c):
        return self * (1.0 / float(c))

    def __rmatmul__(self, M):
        M = np.asarray(M, dtype=float)
        if M.ndim != 2 or M.shape[1] != self.size:
            raise ValueError(f"matmul shapes {M.shape} @ {self.shape}")
        return Expression({v: M @ a for v, a in self.A.items()}, M @ self.b, (M.shape[0],))

    def __getitem__(self, key):
        idx = np.arange(self.size).reshape(self.shape if self.shape else (1,))[key]
        idx = np.atleast_1d(idx).reshape(-1)
        return Expression({v: a[idx] for v, a in self.A.items()}, self.b[idx], () if np.isscalar(key) or isinstance(key, (int, np.integer)) else (len(idx),))

    # ---- constraints
    def __le__(self, o):
        return Constraint(self - o, "<=")

    def __ge__(self, o):
        return Constraint(self - o, ">=")

    def __rshift__(self, o):
        return PSDConstraint(self, o)

    def __rrshift__(self, o):
        return PSDConstraint(Expression.lift(o), self)

    # ---- evaluation
    @property
    def value(self):
        out = self.b.copy()
        for v, a in self.A.items():
            if v.value is None:
                return None
            out = out + a @ np.asarray(v.value, dtype=float).reshape(-1)
        return out.reshape(self.shape) if self.shape else float(out[0])


class Variable(Expression):
    def __init__(self, shape=(), name: str = "var"):
        shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(shape)
        n = int(np.prod(shape)) if shape else 1
        self.name = name
        self._value = None
        Expression.__init__(self, {self: np.eye(n)}, np.zeros(n), shape)

    def __hash__(self):
        return id(self)

    def __eq__(self, o):
        return self is o

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        self._value = None if v is None else np.asarray(v, dtype=float).reshape(self.shape) if self.shape else float(v)


class BMat(Expression):
    """block matrix: only the blocks are kept (the stub never evaluates a matrix inequality)"""

    def __init__(self, blocks):
        self.blocks = [[Expression.lift(b) for b in row] for row in blocks]
        Expression.__init__(self, {}, np.zeros(1), (len(self.blocks), len(self.blocks[0])))


class Constraint:
    def __init__(self, expr: Expression, sense: str):
        self.expr, self.sense = expr, sense

    def violation(self):
        v = np.atleast_1d(self.expr.value)
        return np.maximum(v if self.sense == "<=" else -v, 0.0)


class PSDConstraint:
    def __init__(self, lhs, rhs):
        self.lhs, self.rhs = lhs, rhs

    def violation(self):
        return 0.0


def bmat(blocks):
    return BMat(blocks)


def reshape(expr, shape, order="C"):
    e = Expression.lift(expr)
    return Expression(e.A, e.b, tuple(shape))


def hstack(parts):
    parts = [Expression.lift(p) for p in parts]
    vs = set().union(*[set(p.A) for p in parts])
    A = {v: np.vstack([p.A.get(v, np.zeros((p.size, v.size))) for p in parts]) for v in vs}
    return Expression(A, np.concatenate([p.b for p in parts]), (builtins.sum(p.size for p in parts),))


def sum(terms):  # noqa: A001 (cvxpy's own name)
    out = Expression.const(0.0)
    for t in terms:
        out = out + t
    return out


class Minimize:
    def __init__(self, expr):
        self.expr = Expression.lift(expr)


class Problem:
    def __init__(self, objective, constraints=()):
        self.objective, self.constraints, self.status = objective, list(constraints), None

    def solve(self, solver=None, verbose=False, **opts):
        """min || e(x) ||^2 over the vector variable x of the Schur block [[u - rho2, e^T], [e, I]] >> 0 of the objective's scalar u
        (sdp.py:560-571), every pair x_i <= c, x_i >= c treated as the equality x_i = c; all other constraints are ignored."""
        schur = [c for c in self.constraints if isinstance(c, PSDConstraint) and isinstance(c.lhs, BMat) and len(c.lhs.blocks) == 2
                 and c.lhs.blocks[1][0].A]
        if len(schur) != 1:
            raise SolverError("stub cvxpy: expected exactly one Schur-complement block in the constraints")
        blocks = schur[0].lhs.blocks
        e, top = blocks[1][0], blocks[0][0]
        xs = [v for v in e.A if v.size > 1]
        if len(xs) != 1:
            raise SolverError("stub cvxpy: the residual must depend on exactly one vector variable")
        x = xs[0]
        A, b = e.A[x], e.b
        lo, hi = {}, {}
        for c in self.constraints:
            if isinstance(c, Constraint) and set(c.expr.A) == {x} and c.expr.size == 1:
                row = c.expr.A[x][0]
                nz = np.flatnonzero(row)
                if len(nz) == 1 and row[nz[0]] == 1.0:
                    (hi if c.sense == "<=" else lo)[int(nz[0])] = -float(c.expr.b[0])
        fixed = {i: v for i, v in hi.items() if i in lo and lo[i] == v}
        free = np.array([i for i in range(x.size) if i not in fixed], dtype=int)
        xv = np.zeros(x.size)
        for i, v in fixed.items():
            xv[i] = v
        rhs = -(b + A @ xv)
        xv[free] = np.linalg.lstsq(A[:, free], rhs, rcond=None)[0]
        x.value = xv
        res = float(np.sum((A @ xv + b) ** 2))
        for v in top.A:  # u - rho2 >= ||e||^2  ->  u = ||e||^2 - (constant part of the top-left block)
            v.value = res - float(top.b[0])
        self.status = "optimal"
        self.value = res
        return self.value
