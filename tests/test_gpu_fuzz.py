"""Randomised parity sweep of the hot path ("cover the edge cases"): random kinematic trees (revolute / prismatic / fixed
joints, any branching), every base / friction / gravity-only mode, random batch lengths (odd, shorter than a tile pair, several chunks),
0 ... 3 right-hand sides, row weights with masked rows, and the three column-reduction modes drawn per case -- materialised regressor,
fused Gram (twice: bitwise repeatable), grouped Gram, TSQR and prediction against the CPU oracle.

Further down: the preprocessing kernels against SciPy, the Fourier-series state generator against its NumPy form, factor merges.

FBR_FUZZ_CASES (default 24) and FBR_FUZZ_SEED (default 2025) size the sweep: `FBR_FUZZ_CASES=400 pytest tests/test_gpu_fuzz.py -m gpu`
is the long run; a failing case prints the parameters that reproduce it."""
import os

import numpy as np
import pytest

from common import random_states, random_topology

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("FBR_FUZZ_CASES", "24"))
SEED = int(os.environ.get("FBR_FUZZ_SEED", "2025"))


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _draw(case):
    rng = np.random.default_rng([SEED, case])
    p = dict(case=case, L=int(rng.integers(2, 46)), floating=int(rng.random() < 0.5), fric=int(rng.random() < 0.4), fric_sym=int(rng.random() < 0.5),
             grav=int(rng.random() < 0.1), p_fixed=float(rng.choice([0.0, 0.2, 0.5])), p_prism=float(rng.choice([0.0, 0.0, 0.3])),
             branch=float(rng.choice([0.0, 0.3, 0.7, 1.0])), S=int(rng.choice([1, 2, 3, 7, 64, 129, 256, 500, 777])), k=int(rng.integers(0, 4)),
             weights=int(rng.integers(0, 3)), mode=str(rng.choice(["default", "reduced", "allcols"])), chunk=int(rng.choice([0, 0, 64, 200])),
             groups=int(rng.random() < 0.5))
    # (second generation of the sweep, drawn after the first so that the earlier cases stay what they were)
    if rng.random() < 0.25:
        p["S"] = int(rng.choice([1024, 1536, 2053]))
    p["stribeck"] = float(rng.choice([0.0, 0.0, 0.05])) if p["fric"] else 0.0
    if rng.random() < 0.15:
        p["weights"] = int(rng.choice([3, 4]))  # 3: base-wrench rows only (every joint row masked), 4: joint rows only
    p["grouped_min"] = int(rng.choice([512, 1]))
    return p, rng


def _weights(p, rng, S, rows, fb):
    if not p["weights"]:
        return None
    w = 0.5 + rng.random(S * rows)
    wr = w.reshape(S, rows)
    if p["weights"] == 2:  # masked rows (weight 0): a whole regressor row for every sample, and scattered ones
        wr[:, int(rng.integers(0, rows))] = 0.0
        w[rng.random(S * rows) < 0.1] = 0.0
    elif p["weights"] == 3 and fb:
        wr[:, fb:] = 0.0
    elif p["weights"] == 4 and fb:
        wr[:, :fb] = 0.0
    return w


def _options(p):
    opts = {"default": {}, "reduced": {"reduce_min_work": 0, "tsqr_group_min_samples": 1, "reduce_grouped_min_samples": p["grouped_min"]},
            "allcols": {"link_merge": 0}}[p["mode"]]
    if p["chunk"]:
        opts = dict(opts, chunk_samples=p["chunk"])
    if p["groups"] and p["mode"] != "reduced":
        opts = dict(opts, tsqr_group_min_samples=1)
    return opts


@pytest.mark.parametrize("case", range(CASES))
def test_random_tree_against_the_oracle(case):
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    p, rng = _draw(case)
    t = random_topology(rng, p["L"], p_fixed=p["p_fixed"], branchiness=p["branch"], p_prismatic=p["p_prism"])
    if t.num_dofs == 0 or t.num_dofs + (6 if p["floating"] else 0) > 60:
        pytest.skip("row count outside the fused kernels")
    if p["grav"] and p["fric"]:
        p["fric"] = 0
    om = OracleModel(t, floating=bool(p["floating"]), fric=bool(p["fric"]), fric_sym=bool(p["fric_sym"]), grav_only=bool(p["grav"]), stribeck=p["stribeck"])
    S, k = p["S"], p["k"]
    st = random_states(t, S, rng, p["floating"])
    sign = np.tanh(st["dq"] / 0.02)
    st["sign"] = sign
    Yo = om.regressor(st, sign)
    rows = om.rows
    rhs = rng.standard_normal((S * rows, k)) if k else None
    w = _weights(p, rng, S, rows, 6 if p["floating"] else 0)
    opts = _options(p)
    eng = Engine(t, floating=bool(p["floating"]), friction=bool(p["fric"]), friction_symmetric=bool(p["fric_sym"]), gravity_only=bool(p["grav"]),
                 stribeck_velocity=p["stribeck"], options=opts)
    why = f"reproduce: FBR_FUZZ_SEED={SEED} case {case}: {p}"
    try:
        A = Yo if rhs is None else np.hstack([Yo, rhs])
        if w is not None:
            A = A * w[:, None]
        Go = A.T @ A
        gn = max(np.linalg.norm(Go), 1e-300)
        Y = eng.regressor(st)
        assert Y.shape == Yo.shape and np.abs(Y - Yo).max() <= 1e-11 * max(1.0, np.abs(Yo).max()), why
        G = eng.gram(st, rhs=rhs, w=w)
        assert np.linalg.norm(G - Go) <= 1e-11 * gn, (why, _rel(G, Go))
        assert np.array_equal(G, G.T), why
        assert np.array_equal(G, eng.gram(st, rhs=rhs, w=w)), why  # deterministic to the bit
        R = eng.tsqr(st, rhs=rhs, w=w)
        assert np.all(np.tril(R, -1) == 0), why
        assert np.linalg.norm(R.T @ R - Go) <= 1e-11 * gn, (why, _rel(R.T @ R, Go))
        assert np.array_equal(R, eng.tsqr(st, rhs=rhs, w=w)), why
        for ng in (2, 3):
            if S % ng == 0 and S >= ng:
                Gg = eng.gram_grouped(st, ng, rhs=rhs, w=w)
                h = S // ng * rows
                for g in range(ng):
                    Ag = A[g * h:(g + 1) * h]
                    assert np.linalg.norm(Gg[g] - Ag.T @ Ag) <= 1e-11 * gn, (why, "group", g, ng)
        x = rng.standard_normal(om.P)
        tau = eng.predict(st, x)
        ref = (Yo @ x).reshape(S, rows)
        assert np.abs(tau - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), why
    finally:
        eng.close()


@pytest.mark.parametrize("case", range(max(CASES // 2, 1)))
def test_random_tree_other_entry_points(case):
    """The same draw of trees and modes through the remaining entry points: column subsets of the factorisation (`fbr_tsqr_cols`, with a
    streamed second half), inverse dynamics, finite-difference scores, contact torques, and submissions (bitwise the blocking results)."""
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    p, rng = _draw(10_000 + case)
    t = random_topology(rng, p["L"], p_fixed=p["p_fixed"], branchiness=p["branch"], p_prismatic=p["p_prism"])
    if t.num_dofs == 0 or t.num_dofs + (6 if p["floating"] else 0) > 60:
        pytest.skip("row count outside the fused kernels")
    if p["grav"] and p["fric"]:
        p["fric"] = 0
    om = OracleModel(t, floating=bool(p["floating"]), fric=bool(p["fric"]), fric_sym=bool(p["fric_sym"]), grav_only=bool(p["grav"]), stribeck=p["stribeck"])
    S, k = max(p["S"], 2), max(p["k"], 1) if p["k"] < 3 else 2
    st = random_states(t, S, rng, p["floating"])
    st["sign"] = np.tanh(st["dq"] / 0.02)
    Yo = om.regressor(st, st["sign"])
    rows, P = om.rows, om.P
    rhs = rng.standard_normal((S * rows, k))
    w = _weights(p, rng, S, rows, 6 if p["floating"] else 0)
    opts = _options(p)
    eng = Engine(t, floating=bool(p["floating"]), friction=bool(p["fric"]), friction_symmetric=bool(p["fric_sym"]), gravity_only=bool(p["grav"]),
                 stribeck_velocity=p["stribeck"], options=opts)
    why = f"reproduce: FBR_FUZZ_SEED={SEED} case {10_000 + case}: {p}"
    try:
        A = np.hstack([Yo, rhs]) * (1.0 if w is None else w[:, None])
        # a random subset of the columns (in increasing order, as Model.independent_cols is), whole and streamed in two halves
        ncols = int(rng.integers(1, P + 1))
        cols = np.sort(rng.choice(P, ncols, replace=False))
        sel = list(cols) + [P + i for i in range(k)]
        Gs = A[:, sel].T @ A[:, sel]
        gn = max(np.linalg.norm(Gs), 1e-300)
        Rc = eng.tsqr(st, rhs=rhs, w=w, cols=cols)
        assert Rc.shape == (ncols + k, ncols + k) and np.all(np.tril(Rc, -1) == 0), why
        assert np.linalg.norm(Rc.T @ Rc - Gs) <= 1e-11 * gn, (why, "cols", ncols)
        h = S // 2
        a = {kk: v[:h] for kk, v in st.items()}
        b = {kk: v[h:] for kk, v in st.items()}
        wa, wb = (None, None) if w is None else (w[: h * rows], w[h * rows:])
        R2 = eng.tsqr(b, rhs=rhs[h * rows:], w=wb, cols=cols, R_in=eng.tsqr(a, rhs=rhs[: h * rows], w=wa, cols=cols))
        assert np.linalg.norm(R2.T @ R2 - Gs) <= 1e-11 * gn, (why, "streamed cols")
        G2 = eng.gram(b, rhs=rhs[h * rows:], w=wb, out=eng.gram(a, rhs=rhs[: h * rows], w=wa), accumulate=True)
        assert np.linalg.norm(G2 - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A), (why, "accumulated gram")
        # inverse dynamics with the friction model
        nfric = P - (4 if p["grav"] else 10) * t.num_links
        x_std = np.concatenate([t.x_std(), rng.random(max(nfric, 0) + 4 * t.num_dofs)])
        vel_sign = st["dq"] * 0.9 if p["stribeck"] > 0 else None
        tau = eng.inverse_dynamics(st, x_std, vel_sign=vel_sign)
        tau_o = om.inverse_dynamics(st, x_std, st["sign"], vel_sign)
        assert np.abs(tau - tau_o).max() <= 1e-11 * max(np.abs(tau_o).max(), 1.0), (why, "inverse dynamics")
        # contact torques at the last link
        wr = rng.standard_normal((S, 6))
        ct = eng.contact_torques(st, t.link_names[-1], wr)
        ct_o = om.contact_torques(st, t.link_names[-1], wr)
        assert np.abs(ct - ct_o).max() <= 1e-11 * max(np.abs(ct_o).max(), 1.0), (why, "contact")
        # finite-difference scores on a few samples
        Sf = min(S, 5)
        sf = {kk: v[:Sf] for kk, v in st.items()}
        W = rng.standard_normal((Sf * rows, P))
        eps = 1e-6
        sc = eng.fd_scores(sf, W, eps)
        n = t.num_dofs
        Wb = W.reshape(Sf, rows, P)
        ref = np.empty_like(sc)
        ref[:, 0] = np.einsum("src,src->s", Wb, Yo[: Sf * rows].reshape(Sf, rows, P))
        for kind, key in enumerate(("q", "dq", "ddq")):
            for d in range(n):
                sp = {kk: v.copy() for kk, v in sf.items()}
                sp[key][:, d] += eps
                ref[:, 1 + kind * n + d] = np.einsum("src,src->s", Wb, om.regressor(sp, sf["sign"]).reshape(Sf, rows, P))
        assert np.abs(sc - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1.0), (why, "fd scores")
        # submissions (device-resident operands): two in flight, bitwise the blocking results
        import torch

        dev = torch.device("cuda", 0)
        dv = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
        sd = {kk: dv(v) for kk, v in st.items()}
        Gb, Rb = eng.gram(sd, rhs=dv(rhs), w=dv(w)), eng.tsqr(sd, rhs=dv(rhs), w=dv(w))
        assert np.array_equal(Gb.cpu().numpy(), eng.gram(st, rhs=rhs, w=w)), (why, "device vs host operands")
        Gsub, Rsub = torch.zeros_like(Gb), torch.zeros_like(Rb)
        rd, wd = dv(rhs), dv(w)
        t1 = eng.gram_submit(sd, Gsub, rhs=rd, w=wd)
        t2 = eng.tsqr_submit(sd, Rsub, rhs=rd, w=wd)
        eng.wait(t1)
        eng.wait(t2)
        torch.cuda.synchronize()
        assert torch.equal(Gsub, Gb) and torch.equal(Rsub, Rb), (why, "submissions")
    finally:
        eng.close()


@pytest.mark.parametrize("case", range(max(CASES // 6, 2)))
def test_random_tree_at_the_sizes_where_the_library_switches_paths(case):
    """Default options only, batches long enough that the library itself takes the column reductions (`reduce_min_work`), the row groups of
    the TSQR (`tsqr_group_min_samples`) and several chunks: what a user gets on a robot that is none of the bundled ones.  Gram and
    R^T R against the oracle's Gram accumulated in slices; the fused pass and the factorisation repeat to the bit."""
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    rng = np.random.default_rng([SEED, 20_000 + case])
    L = int(rng.integers(12, 46))
    floating = int(rng.random() < 0.6)
    t = random_topology(rng, L, p_fixed=float(rng.choice([0.0, 0.3, 0.5])), branchiness=float(rng.choice([0.2, 0.5, 0.8])),
                        p_prismatic=float(rng.choice([0.0, 0.2])))
    if t.num_dofs == 0 or t.num_dofs + (6 if floating else 0) > 60:
        pytest.skip("row count outside the fused kernels")
    fric = int(rng.random() < 0.3)
    om = OracleModel(t, floating=bool(floating), fric=bool(fric), fric_sym=True)
    S = int(rng.choice([26_000, 31_111, 48_000]))
    k = int(rng.integers(1, 3))
    st = random_states(t, S, rng, floating)
    st["sign"] = np.tanh(st["dq"] / 0.02)
    rows, P = om.rows, om.P
    rhs = rng.standard_normal((S * rows, k))
    w = 0.5 + rng.random(S * rows) if rng.random() < 0.4 else None
    Go = np.zeros((P + k, P + k))
    for a in range(0, S, 4000):
        b = min(S, a + 4000)
        A = np.hstack([om.regressor({kk: v[a:b] for kk, v in st.items()}, st["sign"][a:b]), rhs[a * rows:b * rows]])
        if w is not None:
            A *= w[a * rows:b * rows, None]
        Go += A.T @ A
    eng = Engine(t, floating=bool(floating), friction=bool(fric), friction_symmetric=True)
    why = f"reproduce: FBR_FUZZ_SEED={SEED} case {20_000 + case}: L={L} floating={floating} fric={fric} S={S} k={k} weights={w is not None}"
    try:
        info = eng.link_merge_info(S)
        gn = np.linalg.norm(Go)
        G = eng.gram(st, rhs=rhs, w=w)
        assert np.linalg.norm(G - Go) <= 1e-11 * gn, (why, info, _rel(G, Go))
        assert np.array_equal(G, eng.gram(st, rhs=rhs, w=w)), why
        R = eng.tsqr(st, rhs=rhs, w=w)
        assert np.all(np.tril(R, -1) == 0) and np.linalg.norm(R.T @ R - Go) <= 1e-11 * gn, (why, info, _rel(R.T @ R, Go))
        assert np.array_equal(R, eng.tsqr(st, rhs=rhs, w=w)), why
    finally:
        eng.close()


@pytest.mark.parametrize("case", range(max(CASES // 2, 1)))
def test_random_signals_through_the_preprocessing_kernels(case):
    """N2 (`Data.preprocess` on the device): fbr_filtfilt / fbr_medfilt / fbr_central_diff on random lengths (around the 2048-sample blocks
    of the state scan), channel counts, filter orders and cut-offs, kernel widths and time grids against SciPy / the host version."""
    import scipy.signal as sig

    from common import load_topo
    from flobaroid_amd._lib import Engine
    from flobaroid_amd.data import Data

    rng = np.random.default_rng([SEED, 30_000 + case])
    eng = Engine(load_topo("threeLinks"))
    try:
        order = int(rng.integers(1, 9))
        fc = float(rng.uniform(0.01, 0.45))
        S = int(rng.choice([3 * (order + 1) + 1, 64, 2047, 2048, 2049, 4097, int(rng.integers(50, 20000))]))
        C = int(rng.integers(1, 40))
        X = np.cumsum(rng.standard_normal((S, C)), axis=0) * 0.1 + 3.0 * rng.standard_normal(C)
        why = f"reproduce: FBR_FUZZ_SEED={SEED} case {30_000 + case}: order={order} fc={fc:.4f} S={S} C={C}"
        b, a = sig.butter(order, fc)
        # (b, a) of a high-order, low cut-off Butterworth is an ill-conditioned description of the filter (sum a ~ prod (1 - pole) against
        # coefficients of order 2^order): SciPy's own result moves with the rounding of its start states (a linear solve there, the
        # closed form here) by about eps * cond -- the comparison allows that much and skips what no double-precision run defines
        cond = float(np.abs(a).sum() / abs(a.sum()))
        if S > 3 * (order + 1) and cond < 1e10:
            want = sig.filtfilt(b, a, X, axis=0)
            got = eng.filtfilt(b, a, X.copy())
            assert np.all(np.isfinite(got)), why
            assert np.abs(got - want).max() <= max(1e-10, 1e-14 * cond) * max(np.abs(want).max(), 1.0), (why, cond)
            nc = int(rng.integers(1, C + 1))
            Y = X.copy()
            eng.filtfilt(b, a, Y, ncols=nc)
            assert np.array_equal(Y[:, :nc], got[:, :nc]) and np.array_equal(Y[:, nc:], X[:, nc:]), why
        k = int(rng.choice([1, 3, 5, 7, 9, 11]))
        assert np.array_equal(eng.medfilt(k, X.copy()), sig.medfilt(X, (k, 1))), (why, k)
        if S >= 3:
            T = np.cumsum(rng.uniform(0.001, 0.02) + 1e-4 * rng.random(S))
            want = Data._central_diff(X, T)
            got = eng.central_diff(X, T)
            assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-300), why
    finally:
        eng.close()


@pytest.mark.parametrize("case", range(max(CASES // 3, 1)))
def test_random_fourier_candidates_and_factor_merges(case):
    """fbr_fourier_states (classic and bounded generators: trajectoryGenerator.py:411-510 written out in NumPy) on random candidate counts,
    lengths, harmonics and rates, host and device outputs; fbr_tsqr_merge of random triangular factors of every width, some rank
    deficient, against the Gram of the stacked pair."""
    from flobaroid_amd._lib import Engine

    rng = np.random.default_rng([SEED, 40_000 + case])
    t = random_topology(rng, int(rng.integers(2, 30)), p_fixed=0.2, branchiness=0.5)
    if t.num_dofs == 0:
        pytest.skip("no joint")
    eng = Engine(t)
    try:
        n = t.num_dofs
        C, T, nh = int(rng.integers(1, 9)), int(rng.choice([1, 2, 37, 256, 1000])), int(rng.integers(1, 7))
        freq = float(rng.choice([50.0, 200.0, 1000.0]))
        wf = rng.uniform(0.3, 3.0, C)
        a, b = rng.standard_normal((C, n, nh)), rng.standard_normal((C, n, nh))
        qo = rng.standard_normal((C, n))
        bounded = bool(rng.random() < 0.5)
        qr = rng.uniform(0.1, 1.5, (C, n)) if bounded else None
        why = f"reproduce: FBR_FUZZ_SEED={SEED} case {40_000 + case}: C={C} T={T} nh={nh} freq={freq} bounded={bounded} n={n}"
        ts = np.arange(T) / freq
        l = np.arange(1, nh + 1)
        want = {k: np.empty((C * T, n)) for k in ("q", "dq", "ddq")}
        for c in range(C):
            wl = wf[c] * l                                              # (nh,)
            ang = ts[:, None] * wl[None, :]                             # (T, nh)
            sn, cs = np.sin(ang), np.cos(ang)
            if not bounded:
                q = sn @ (a[c] / wl).T - cs @ (b[c] / wl).T + qo[c]
                dq = cs @ a[c].T + sn @ b[c].T
                ddq = -sn @ (a[c] * wl).T + cs @ (b[c] * wl).T
            else:
                raw = cs @ b[c].T + sn @ a[c].T
                rd = cs @ (a[c] * wl).T - sn @ (b[c] * wl).T
                rdd = -sn @ (a[c] * wl * wl).T - cs @ (b[c] * wl * wl).T
                th = np.tanh(raw)
                s2 = 1.0 - th * th
                q = qo[c] + qr[c] * th
                dq = qr[c] * s2 * rd
                ddq = qr[c] * (s2 * rdd - 2.0 * th * s2 * rd * rd)
            for k, v in (("q", q), ("dq", dq), ("ddq", ddq)):
                want[k][c * T:(c + 1) * T] = v
        got_h = eng.fourier_states(wf, a, b, qo, T, freq, q_range=qr, device=False)
        got_d = eng.fourier_states(wf, a, b, qo, T, freq, q_range=qr, device=True)
        for k in ("q", "dq", "ddq"):
            scale = max(1.0, np.abs(want[k]).max())
            assert np.abs(got_h[k] - want[k]).max() <= 1e-11 * scale, (why, k)
            assert np.array_equal(got_d[k].cpu().numpy(), got_h[k]), (why, k, "device == host")
        # factor merges
        nn = int(rng.integers(1, 700))
        R1, R2 = np.triu(rng.standard_normal((nn, nn))), np.triu(rng.standard_normal((nn, nn)))
        if rng.random() < 0.4:  # a few dependent / zero columns
            for j in rng.integers(0, nn, size=max(1, nn // 10)):
                R1[:, j] = 0.0
                R2[:, j] = 0.0 if rng.random() < 0.5 else R2[:, j]
        Rm = eng.tsqr_merge(R1, R2)
        G = R1.T @ R1 + R2.T @ R2
        assert np.all(np.tril(Rm, -1) == 0) and np.all(np.isfinite(Rm)), (why, nn)
        assert np.linalg.norm(Rm.T @ Rm - G) <= 1e-11 * max(np.linalg.norm(G), 1e-300), (why, nn, _rel(Rm.T @ Rm, G))
        assert np.array_equal(Rm, eng.tsqr_merge(R1, R2)), (why, nn)
    finally:
        eng.close()


@pytest.mark.parametrize("case", range(max(CASES // 4, 1)))
def test_random_robot_through_the_class_level_pipeline(case, tmp_path):
    """The host layer on robots that are none of the bundled ones: random tree (mixed joint types) -> topology file -> Model (structural
    Gram on the device, pivoted QR, base projection) -> Data -> computeRegressors -> identification from the TSQR factor, against the same
    pipeline on the oracle's materialised regressor: same independent columns, base parameters equal to NumPy's lstsq on YBase, the
    noise-free torques reproduced, K x_true recovered."""
    import numpy.linalg as la

    from flobaroid_amd import estimation as est
    from flobaroid_amd.data import Data
    from flobaroid_amd.model import Model, pivoted_qr
    from oracle.oracle import OracleModel

    rng = np.random.default_rng([SEED, 50_000 + case])
    L = int(rng.integers(3, 16))
    floating = int(rng.random() < 0.5)
    t = random_topology(rng, L, p_fixed=float(rng.choice([0.0, 0.3])), branchiness=float(rng.choice([0.0, 0.5, 1.0])), p_prismatic=float(rng.choice([0.0, 0.3])))
    if t.num_dofs == 0:
        pytest.skip("no joint")
    t.limits = {j: {"lower": -2.0, "upper": 2.0, "velocity": 3.0, "torque": 100.0} for j in t.dof_names}
    path = str(tmp_path / "random.topology.json")
    t.save_json(path)
    mode = str(rng.choice(["default", "reduced", "allcols"]))
    eo = {"default": {}, "reduced": {"reduce_min_work": 0, "tsqr_group_min_samples": 1}, "allcols": {"link_merge": 0}}[mode]
    opt = dict(floatingBase=floating, identifyFrictionSimultaneously=0, identifySymmetricVelFriction=1, identifyGravityParamsOnly=0,
               simulateTorques=0, useAPriori=0, useStructuralRegressor=1, skipSamples=0, startOffset=0, verbose=0, showTiming=0,
               filterRegressor=0, estimateWith="std", randomSamples=1500, minTol=1e-6, selectBlocksFromMeasurements=0, engineOptions=eo)
    why = f"reproduce: FBR_FUZZ_SEED={SEED} case {50_000 + case}: L={L} floating={floating} mode={mode} n={t.num_dofs}"
    np.random.seed(int(rng.integers(1 << 30)))
    state = np.random.get_state()
    model = Model(opt, path)
    om = OracleModel(t, floating=bool(floating))
    # the CPU path's structural Gram on the same random states
    np.random.set_state(state)
    st_r = model._random_states(1500)
    Yr = om.regressor(st_r)
    Q, R, P = pivoted_qr(Yr.T @ Yr)
    r = int(np.count_nonzero(np.abs(np.diag(R)) > opt["minTol"]))
    dg = np.abs(np.diag(R))
    if r < len(dg) and r > 0 and dg[r - 1] < 1e3 * opt["minTol"]:
        pytest.skip("a pivot within three decades of minTol: the rank is a matter of rounding")
    assert model.num_base_params == r, why
    assert np.array_equal(np.sort(np.asarray(model.independent_cols)), np.sort(P[:r])), why
    # identification on noise-free data
    S = 600
    st = random_states(t, S, rng, floating, use_limits=True)
    x_true = t.x_std()
    tau = om.inverse_dynamics(st, x_true)
    meas = {"positions": st["q"], "velocities": st["dq"], "accelerations": st["ddq"], "torques": tau[:, 6:] if floating else tau, "times": np.arange(S) / 200.0}
    if floating:
        meas.update(base_velocity=st["base_vel"], base_acceleration=st["base_acc"], base_rpy=st["rpy"])
    data = Data(opt)
    data.init_from_data(meas)
    model.computeRegressors(data)
    Yo = om.regressor(st)
    assert np.abs(np.asarray(model.YStd) - Yo).max() <= 1e-11 * max(1.0, np.abs(Yo).max()), why
    ic = np.asarray(model.independent_cols)
    xb_ref = la.lstsq(Yo[:, ic], model.tau, rcond=None)[0]
    R_aug = model.engine.tsqr(model._states, rhs=np.stack((model.tau, model.contactForcesSum), axis=1))
    xB, Rb, s = est.identify_base_parameters(R_aug, model.independent_cols, model.num_identified_params, Yo.shape[0])
    assert la.norm(xB - xb_ref) <= 1e-7 * max(la.norm(xb_ref), 1e-300), (why, la.norm(xB - xb_ref) / la.norm(xb_ref))
    assert la.norm(Yo[:, ic] @ xB - model.tau) <= 1e-8 * la.norm(model.tau), why          # noise-free: the torques are reproduced
    xb_true = model.K @ x_true[model.identified_params]
    assert la.norm(xB - xb_true) <= 1e-5 * max(la.norm(xb_true), 1e-300), (why, la.norm(xB - xb_true) / la.norm(xb_true))


@pytest.mark.parametrize("opts", [{}, {"reduce_min_work": 0, "tsqr_group_min_samples": 1}], ids=["default", "reduced"])
def test_non_finite_inputs_come_back_and_leave_the_handle_usable(opts):
    """A NaN, an Inf or an overflowing value in the states of one sample: every call returns (non-finite results, no wait that never ends in
    the flag protocols of the factorisation) and the next call on clean inputs is right."""
    from common import load_topo
    from flobaroid_amd._lib import Engine

    t = load_topo("walkman_apriori")
    rng = np.random.default_rng(0)
    S = 3000
    eng = Engine(t, floating=True, options=opts)
    try:
        st = random_states(t, S, rng, True)
        rhs = rng.standard_normal((S * 35, 1))
        G0 = eng.gram(st, rhs=rhs)
        for bad in (np.nan, np.inf, 1e300):
            s2 = {k: v.copy() for k, v in st.items()}
            s2["ddq"][S // 2, 3] = bad
            s2["q"][7, 0] = bad if np.isfinite(bad) else 1e18
            G = eng.gram(s2, rhs=rhs)
            R = eng.tsqr(s2, rhs=rhs)
            eng.predict(s2, rng.standard_normal(480))
            assert not np.isfinite(G).all() and not np.isfinite(R).all()
            assert np.array_equal(eng.gram(st, rhs=rhs), G0)
    finally:
        eng.close()
