"""GPU parity: every HIP entry point of the C-ABI against the CPU oracle on seeded inputs.

Tolerances (fp64 path, stated by north_star as 1e-6 on identified parameters; the kernels are held to
much tighter bars): regressor / torques entries <= 1e-11 * max|.| of the array, Gram <= 1e-11 relative
Frobenius.
"""
import os

import numpy as np
import pytest

from common import CONFIGS, cfg_id, load_topo, random_states

# every test runs with the product's own choice of the column reductions, with the reductions forced and with them switched off (conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reduction_mode")]


def _engine_oracle(cfg):
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    name, fl, fr, sym, grav, strb = cfg
    t = load_topo(name)
    eng = Engine(t, floating=fl, friction=fr, friction_symmetric=sym, gravity_only=grav, stribeck_velocity=strb)
    om = OracleModel(t, floating=fl, fric=fr, fric_sym=sym, grav_only=grav, stribeck=strb)
    return t, eng, om


def _states(t, cfg, S, seed):
    rng = np.random.default_rng(seed)
    st = random_states(t, S, rng, cfg[1])
    if cfg[4]:
        st["dq"][:] = 0.0
        st["ddq"][:] = 0.0
    st["sign"] = np.tanh(st["dq"] / 0.02)
    return st, rng


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
def test_regressor_matches_oracle(cfg):
    t, eng, om = _engine_oracle(cfg)
    assert (eng.rows, eng.cols) == (om.rows, om.P)
    st, _ = _states(t, cfg, 37, 1)
    Y = eng.regressor(st)
    Yo = om.regressor(st, st["sign"])
    assert Y.shape == Yo.shape
    assert np.abs(Y - Yo).max() <= 1e-11 * np.abs(Yo).max()
    # STRUCTURAL zeros -- decided by the topology: the row of joint d in the columns of a link that does not hang below d, every
    # row but its joint's in a friction column -- are exact zeros.  Entries the oracle evaluates to exactly 0.0 for another reason
    # (an analytic cancellation of its body-frame formulation, e.g. a link origin on its joint axis) only have to vanish to
    # rounding in the base-frame formulation of the kernels.
    fb = 6 if cfg[1] else 0
    cpl = 4 if cfg[4] else 10
    anc = t.ancestors_dofs()
    mask = np.zeros((eng.rows, eng.cols), dtype=bool)
    for l in range(t.num_links):
        for d in range(t.num_dofs):
            if d not in anc[l]:
                mask[fb + d, cpl * l:cpl * (l + 1)] = True
    for c in range(cpl * t.num_links, eng.cols):
        j = (c - cpl * t.num_links) % t.num_dofs
        mask[:, c] = True
        mask[fb + j, c] = False
    Y3, Yo3 = Y.reshape(37, eng.rows, eng.cols), Yo.reshape(37, eng.rows, eng.cols)
    assert np.all(Yo3[:, mask] == 0.0) and np.all(Y3[:, mask] == 0.0)
    assert np.abs(Y[Yo == 0.0]).max(initial=0.0) <= 1e-13 * np.abs(Yo).max()


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
def test_inverse_dynamics_and_predict(cfg):
    t, eng, om = _engine_oracle(cfg)
    st, rng = _states(t, cfg, 41, 2)
    nfric = om.P - (4 if cfg[4] else 10) * t.num_links
    x_std = np.concatenate([t.x_std(), rng.random(max(nfric, 0) + 4 * t.num_dofs)])
    vel_sign = st["dq"] * 0.9
    tau = eng.inverse_dynamics(st, x_std, vel_sign=vel_sign)
    tau_o = om.inverse_dynamics(st, x_std, st["sign"], vel_sign)
    assert np.abs(tau - tau_o).max() <= 1e-11 * np.abs(tau_o).max()
    x = rng.standard_normal(om.P)
    Yo = om.regressor(st, st["sign"])
    pred = eng.predict(st, x)
    ref = (Yo @ x).reshape(pred.shape)
    assert np.abs(pred - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[2], CONFIGS[7]], ids=cfg_id)
def test_contact_torques(cfg):
    t, eng, om = _engine_oracle(cfg)
    st, rng = _states(t, cfg, 29, 3)
    frames = list(t.frames)[:2] + [t.link_names[-1]]
    for fr in frames:
        if t.frames.get(fr, {}).get("link", 0) < 0:
            continue
        w = rng.standard_normal((29, 6))
        out = eng.contact_torques(st, fr, w)
        ref = om.contact_torques(st, fr, w)
        assert np.abs(out - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
@pytest.mark.parametrize("k", [0, 2])
def test_gram_matches_oracle(cfg, k):
    t, eng, om = _engine_oracle(cfg)
    S = 300
    st, rng = _states(t, cfg, S, 4)
    rhs = rng.standard_normal((S * om.rows, k)) if k else None
    G = eng.gram(st, rhs=rhs)
    Yo = om.regressor(st, st["sign"])
    Ya = Yo if rhs is None else np.hstack([Yo, rhs])
    Go = Ya.T @ Ya
    assert G.shape == Go.shape
    assert np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go)
    assert np.array_equal(G, G.T)
    # accumulate + weights (0/1 mask of the base-wrench rows and a random positive weight)
    w = rng.random(S * om.rows) + 0.5
    if cfg[1]:
        w.reshape(S, om.rows)[:, 6:] = 0.0
    G2 = eng.gram(st, rhs=rhs, w=w, out=G.copy(), accumulate=True)
    Yw = Ya * w[:, None]
    assert np.linalg.norm(G2 - (Go + Yw.T @ Yw)) <= 1e-11 * np.linalg.norm(Go)


def test_device_pointers_and_chunking():
    """torch CUDA tensors go through as device pointers; a batch larger than one chunk is split."""
    import torch

    cfg = CONFIGS[2]
    t, eng, om = _engine_oracle(cfg)
    st, rng = _states(t, cfg, 5000, 5)
    dst = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
    eng.use_torch_stream()
    G = eng.gram(dst)
    assert G.is_cuda
    Yo = om.regressor(st, st["sign"])
    Go = Yo.T @ Yo
    assert np.linalg.norm(G.cpu().numpy() - Go) <= 1e-11 * np.linalg.norm(Go)
    Y = eng.regressor(dst)
    assert np.abs(Y.cpu().numpy() - Yo).max() <= 1e-11 * np.abs(Yo).max()


def _aug(om, st, rhs, w=None):
    Yo = om.regressor(st, st["sign"])
    A = Yo if rhs is None else np.hstack([Yo, rhs])
    if w is not None:
        A = A * w[:, None]
    return A


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[3], CONFIGS[6], CONFIGS[7]], ids=cfg_id)
def test_tsqr_matches_householder_reference(cfg):
    """R^T R = A^T A to rounding, R upper triangular, and on a full-rank column subset R agrees with
    numpy.linalg.qr of the materialised matrix (the reference's la.qr(YBase), sdp.py:470) up to row signs."""
    from flobaroid_amd import estimation as est
    from oracle.oracle import lin_deps_qr

    t, eng, om = _engine_oracle(cfg)
    S = 700 if t.num_links > 10 else 1500
    st, rng = _states(t, cfg, S, 11)
    rhs = rng.standard_normal((S * om.rows, 2))
    A = _aug(om, st, rhs)
    R = eng.tsqr(st, rhs=rhs)
    assert np.all(np.tril(R, -1) == 0.0)
    G = A.T @ A
    assert np.linalg.norm(R.T @ R - G) <= 1e-12 * np.linalg.norm(G)
    # full-rank subset: the structural base columns + the rhs columns
    d = lin_deps_qr(G[: om.P, : om.P], 1e-6 * np.abs(G).max())
    ic = d["independent_cols"]
    Rb = est.qr_subset(R, ic, om.P)
    Rn = np.linalg.qr(A[:, np.concatenate((ic, [om.P, om.P + 1]))], mode="r")
    Rn = Rn * np.where(np.diag(Rn) < 0, -1.0, 1.0)[:, None]
    assert np.linalg.norm(Rb - Rn) <= 1e-9 * np.linalg.norm(Rn)


def test_tsqr_streaming_weights_and_merge():
    cfg = CONFIGS[3]
    t, eng, om = _engine_oracle(cfg)
    S = 900
    st, rng = _states(t, cfg, S, 12)
    w = rng.random(S * om.rows) + 0.5
    A = _aug(om, st, None, w)
    G = A.T @ A
    half = S // 2
    st1 = {k: v[:half] for k, v in st.items()}
    st2 = {k: v[half:] for k, v in st.items()}
    R1 = eng.tsqr(st1, w=w[: half * om.rows])
    R12 = eng.tsqr(st2, w=w[half * om.rows:], R_in=R1)
    assert np.linalg.norm(R12.T @ R12 - G) <= 1e-12 * np.linalg.norm(G)
    R2 = eng.tsqr(st2, w=w[half * om.rows:])
    Rm = eng.tsqr_merge(R1, R2)
    assert np.all(np.tril(Rm, -1) == 0.0)
    assert np.linalg.norm(Rm.T @ Rm - G) <= 1e-12 * np.linalg.norm(G)


def test_tsqr_keeps_small_singular_directions():
    """The reason TSQR exists: a column that is nearly dependent (1e-9 relative) keeps its tiny pivot in R,
    while the Gram route loses it (sqrt(eps) * ||A||)."""
    cfg = CONFIGS[2]
    t, eng, om = _engine_oracle(cfg)
    S = 600
    st, rng = _states(t, cfg, S, 13)
    Yo = om.regressor(st, st["sign"])
    base = Yo[:, 29]
    rhs = (base + 1e-9 * rng.standard_normal(base.shape) * np.abs(base).max()).reshape(-1, 1)
    R = eng.tsqr(st, rhs=rhs)
    A = np.hstack([Yo, rhs])
    sel = [29, om.P]
    Rs = np.linalg.qr(R[:, sel], mode="r")
    Rn = np.linalg.qr(A[:, sel], mode="r")
    assert abs(abs(Rs[1, 1]) - abs(Rn[1, 1])) <= 1e-4 * abs(Rn[1, 1])


def test_tsqr_column_subset_is_qr_of_ybase():
    """fbr_tsqr_cols on Model.independent_cols == numpy.linalg.qr(YBase | tau) (sdp.py:470) up to row signs."""
    from oracle.oracle import lin_deps_qr

    cfg = CONFIGS[6]
    t, eng, om = _engine_oracle(cfg)
    S = 800
    st, rng = _states(t, cfg, S, 21)
    rhs = rng.standard_normal((S * om.rows, 1))
    Yo = om.regressor(st, st["sign"])
    d = lin_deps_qr(Yo.T @ Yo, 1e-6 * np.abs(Yo.T @ Yo).max())
    ic = d["independent_cols"]
    R = eng.tsqr(st, rhs=rhs, cols=ic)
    assert R.shape == (len(ic) + 1, len(ic) + 1)
    Rn = np.linalg.qr(np.hstack([Yo[:, ic], rhs]), mode="r")
    sg = lambda M: M * np.where(np.diag(M) < 0, -1.0, 1.0)[:, None]
    assert np.linalg.norm(sg(R) - sg(Rn)) <= 1e-10 * np.linalg.norm(Rn)
    with pytest.raises(Exception):
        eng.tsqr(st, cols=[0, 0, 1])


@pytest.mark.parametrize("name,fl,S", [("kuka_lwr4", 0, 120_000), ("walkman_left_arm", 1, 60_000), ("walkman_apriori", 1, 40_000)])
def test_tsqr_long_streams_agree_with_fused_gram(name, fl, S):
    """Many folds per worker (continuous wave pipeline / wave-private narrow kernels) and a deep merge tree:
    R^T R must equal the fused Gram of the same inputs (itself pinned on the oracle above) -- size-independent property
    at sizes the CPU oracle cannot reach in a test."""
    from flobaroid_amd._lib import Engine

    t = load_topo(name)
    eng = Engine(t, floating=bool(fl))
    rng = np.random.default_rng(77)
    st = random_states(t, S, rng, fl)
    rhs = rng.standard_normal((S * eng.rows, 1))
    R = eng.tsqr(st, rhs=rhs)
    G = eng.gram(st, rhs=rhs)
    assert np.all(np.tril(R, -1) == 0.0) and np.all(np.isfinite(R))
    assert np.linalg.norm(R.T @ R - G) <= 1e-11 * np.linalg.norm(G)
    # streaming in two halves through R_in gives the same Gram
    h = S // 2
    R1 = eng.tsqr({k: v[:h] for k, v in st.items()}, rhs=rhs[: h * eng.rows])
    R2 = eng.tsqr({k: v[h:] for k, v in st.items()}, rhs=rhs[h * eng.rows:], R_in=R1)
    assert np.linalg.norm(R2.T @ R2 - G) <= 1e-11 * np.linalg.norm(G)


@pytest.mark.parametrize("cfg", [CONFIGS[3], CONFIGS[7]], ids=cfg_id)
def test_grouped_gram_equals_one_gram_per_group(cfg):
    """fbr_gram_grouped: one pass over ngroups candidate trajectories == ngroups separate fbr_gram_accumulate calls
    (bitwise up to the fixed slice order: compared to 1e-13), and the batch D-optimality of trajectoryOptimizer.py:263-272."""
    from flobaroid_amd import estimation as est

    t, eng, om = _engine_oracle(cfg)
    ng, Sg = 7, 90
    st, rng = _states(t, cfg, ng * Sg, 31)
    rhs = rng.standard_normal((ng * Sg * om.rows, 1))
    w = rng.random(ng * Sg * om.rows) + 0.5
    Gg = eng.gram_grouped(st, ng, rhs=rhs, w=w)
    assert Gg.shape == (ng, om.P + 1, om.P + 1)
    for g in range(ng):
        sl = slice(g * Sg, (g + 1) * Sg)
        rs = slice(g * Sg * om.rows, (g + 1) * Sg * om.rows)
        G1 = eng.gram({k: v[sl] for k, v in st.items()}, rhs=rhs[rs], w=w[rs])
        assert np.linalg.norm(Gg[g] - G1) <= 1e-13 * np.linalg.norm(G1)
    A = _aug(om, {k: v[:Sg] for k, v in st.items()}, rhs[: Sg * om.rows], w[: Sg * om.rows])
    assert np.linalg.norm(Gg[0] - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A)
    ic = np.flatnonzero(np.diag(Gg.sum(axis=0))[: om.P] > 0)[:20]
    dopt = est.d_optimality_batch(Gg, ic, 1e-4)
    for g in range(ng):
        assert abs(dopt[g] - est.d_optimality(Gg[g], ic, 1e-4)) <= 1e-9 * abs(dopt[g])
    with pytest.raises(Exception):
        eng.gram_grouped(st, 4)  # 630 samples are not a multiple of 4


@pytest.mark.parametrize("Sg", [1, 3, 7])
def test_grouped_gram_odd_group_sizes_floating_base(Sg):
    """The base-wrench rows of sample pairs share MFMA k-steps (even sample: its own rows + two of its partner's): pairs must
    not straddle groups, and the last sample of an odd-sized group has no partner.  Each group against the oracle."""
    cfg = CONFIGS[7]  # WALK-MAN, floating base
    t, eng, om = _engine_oracle(cfg)
    ng = 9
    st, rng = _states(t, cfg, ng * Sg, 57)
    rhs = rng.standard_normal((ng * Sg * om.rows, 2))
    w = rng.random(ng * Sg * om.rows) + 0.5
    for _ in range(2):  # twice: the second pass runs over images whose ghost rows hold the first pass's values
        Gg = eng.gram_grouped(st, ng, rhs=rhs, w=w)
        for g in range(ng):
            sl = slice(g * Sg, (g + 1) * Sg)
            rs = slice(g * Sg * om.rows, (g + 1) * Sg * om.rows)
            A = _aug(om, {k: v[sl] for k, v in st.items()}, rhs[rs], w[rs])
            assert np.linalg.norm(Gg[g] - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A), (Sg, g)
    # a different grouping of the same buffers afterwards (slots change parity)
    G = eng.gram(st, rhs=rhs, w=w)
    A = _aug(om, st, rhs, w)
    assert np.linalg.norm(G - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A)


def test_multi_chunk_paths_at_small_sizes():
    """The option chunk_samples forces the chunked code paths (double-buffered producer stream of the Gram, groups larger /
    smaller than a chunk, several TSQR chunks, chunked regressor / inverse dynamics) at sizes the oracle can check."""
    cfg = CONFIGS[7]
    t, eng, om = _engine_oracle(cfg)
    S = 330
    st, rng = _states(t, cfg, S, 41)
    rhs = rng.standard_normal((S * om.rows, 2))
    A = _aug(om, st, rhs)
    Go = A.T @ A
    x = rng.standard_normal(om.P)
    eng.set_option("chunk_samples", 70)
    G = eng.gram(st, rhs=rhs)
    assert np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go)
    R = eng.tsqr(st, rhs=rhs)
    assert np.linalg.norm(R.T @ R - Go) <= 1e-11 * np.linalg.norm(Go)
    Rw = eng.tsqr(st, rhs=rhs, w=np.full(S * om.rows, 2.0))
    assert np.linalg.norm(Rw.T @ Rw - 4.0 * Go) <= 1e-11 * np.linalg.norm(4.0 * Go)
    Y = eng.regressor(st)
    assert np.abs(Y - A[:, : om.P]).max() <= 1e-11 * np.abs(A).max()
    assert np.abs(eng.predict(st, x) - (A[:, : om.P] @ x).reshape(S, om.rows)).max() <= 1e-10 * np.abs(A @ np.r_[x, 0, 0]).max()
    # groups of 110 samples > chunk of 70 (pieces of one group), and groups of 30 < chunk (two groups per launch)
    for ng in (3, 11):
        Gg = eng.gram_grouped(st, ng, rhs=rhs)
        Sg = S // ng
        for g in range(ng):
            Ag = A[g * Sg * om.rows:(g + 1) * Sg * om.rows]
            assert np.linalg.norm(Gg[g] - Ag.T @ Ag) <= 1e-11 * np.linalg.norm(Ag.T @ Ag)


def test_pinned_host_inputs_are_staged_chunk_by_chunk():
    """Pinned host states / rhs / weights take the chunked staging path of fbr_gram_accumulate (copy stream, double-buffered staging
    buffers, events against the producer stream); pageable ones the up-front copy.  Both must give the device-resident result bit for
    bit (same chunking, same kernels), also for grouped Grams and with a friction layout (sign series staged as well)."""
    import torch

    for cfg in (CONFIGS[7], CONFIGS[6]):
        t, eng, om = _engine_oracle(cfg)
        S = 660
        st, rng = _states(t, cfg, S, 17)
        rhs = rng.standard_normal((S * om.rows, 2))
        w = 0.5 + rng.random(S * om.rows)
        eng.set_option("chunk_samples", 90)   # 8 chunks: both staging buffers are reused several times
        G_pageable = eng.gram(st, rhs=rhs, w=w)
        pin = {k: torch.from_numpy(v).pin_memory() for k, v in st.items()}
        G_pinned = eng.gram(pin, rhs=torch.from_numpy(rhs).pin_memory(), w=torch.from_numpy(w).pin_memory())
        dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
        G_dev = eng.gram(dev, rhs=torch.from_numpy(rhs).cuda(), w=torch.from_numpy(w).cuda()).cpu().numpy()
        assert np.array_equal(G_pinned, G_dev) and np.array_equal(G_pageable, G_dev)
        A = _aug(om, st, rhs, w)
        assert np.linalg.norm(G_dev - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A)
        Gg = eng.gram_grouped(pin, 3, rhs=torch.from_numpy(rhs).pin_memory())
        Gg_dev = eng.gram_grouped(dev, 3, rhs=torch.from_numpy(rhs).cuda()).cpu().numpy()
        assert np.array_equal(Gg, Gg_dev)
        pr = None
        eng.profile_enable(True)
        eng.profile_get()
        eng.gram(pin, rhs=torch.from_numpy(rhs).pin_memory())
        pr = eng.profile_get()
        eng.profile_enable(False)
        assert pr["h2d"][1] == 8 and pr["pack"][1] == 8   # one staging copy group and one packing launch per chunk
        eng.close()


def test_tsqr_column_order_is_internal():
    """The single wide factorisation (short batches, unbranched robots) orders the inertial columns by link depth internally
     and returns the factor in the caller's order: with the reordering switched off the same R^T R and, for a full-rank
    column subset, the same sign-normalised R; an R_in in the caller's order streams through either way; the executed-work
    counter reports the saving."""
    cfg = CONFIGS[7]
    t, eng, om = _engine_oracle(cfg)
    eng.set_option("tsqr_groups", 0)
    eng.set_option("link_merge", 0)  # (the column order of the UNMERGED factorisation is what is looked at)
    S = 1200   # (35 S rows >= 64 n: below that the final re-triangularisation is not worth it and the caller's order is kept)
    st, rng = _states(t, cfg, S, 23)
    rhs = rng.standard_normal((S * om.rows, 1))
    A = _aug(om, st, rhs)
    Go = A.T @ A
    import scipy.linalg as sla

    # 150 linearly independent columns (the first pivots of the Gram): 151 columns with the rhs -> the wide kernels, unique R up to signs
    cols = np.sort(sla.qr(Go[: om.P, : om.P], pivoting=True, mode="r")[1][:150]).astype(np.int32)
    sel = np.r_[cols, om.P]
    h = S // 2
    first = {k: v[:h] for k, v in st.items()}
    second = {k: v[h:] for k, v in st.items()}

    def run():
        R = eng.tsqr(st, rhs=rhs)
        Rc = eng.tsqr(st, rhs=rhs, cols=cols)
        Rs = eng.tsqr(second, rhs=rhs[h * om.rows:], cols=cols, R_in=eng.tsqr(first, rhs=rhs[: h * om.rows], cols=cols))
        return R, Rc, Rs, eng.tsqr_work_info(S, k=1)

    R1, Rc1, Rs1, wi1 = run()
    eng.set_option("tsqr_reorder", 0)
    R0, Rc0, Rs0, wi0 = run()
    for R in (R0, R1):
        assert np.all(np.tril(R, -1) == 0) and np.linalg.norm(R.T @ R - Go) <= 1e-11 * np.linalg.norm(Go)
    norm = lambda R: R * np.where(np.diag(R) < 0, -1.0, 1.0)[:, None]
    Gc = Go[np.ix_(sel, sel)]
    for R in (Rc0, Rc1, Rs0, Rs1):
        assert np.linalg.norm(R.T @ R - Gc) <= 1e-11 * np.linalg.norm(Gc)
        assert np.linalg.norm(norm(R) - norm(Rc0)) <= 1e-9 * np.linalg.norm(Rc0)
    assert wi1["mfma_level0"] < 0.9 * wi0["mfma_level0"] and wi1["mfma_tree"] > wi0["mfma_tree"]


@pytest.mark.parametrize("cfg", [CONFIGS[7], CONFIGS[8], CONFIGS[4]], ids=cfg_id)
def test_tsqr_row_groups_along_the_tree(cfg):
    """Tree-structured TSQR: the base-wrench rows and every unbranched chain of joints are factorised on their own, over
    the columns their rows can touch, and the group factors are folded into the final factor.  Same R^T R as the single
    factorisation, the same sign-normalised R for a full-rank column subset (also with row weights and a streamed R_in), and the
    work counter shows the saving on the branched robots."""
    t, eng, om = _engine_oracle(cfg)
    S = 1500
    st, rng = _states(t, cfg, S, 31)
    rhs = rng.standard_normal((S * om.rows, 2))
    w = 0.5 + rng.random(S * om.rows)
    A = _aug(om, st, rhs) * w[:, None]
    Go = A.T @ A
    import scipy.linalg as sla

    ncol = min(150, int(np.linalg.matrix_rank(Go[: om.P, : om.P])) - 5)
    cols = np.sort(sla.qr(Go[: om.P, : om.P], pivoting=True, mode="r")[1][:ncol]).astype(np.int32)
    sel = np.r_[cols, om.P, om.P + 1]
    h = S // 3
    first = {k: v[:h] for k, v in st.items()}
    second = {k: v[h:] for k, v in st.items()}

    def run():
        R = eng.tsqr(st, rhs=rhs, w=w)
        Rc = eng.tsqr(st, rhs=rhs, w=w, cols=cols)
        Rs = eng.tsqr(second, rhs=rhs[h * om.rows:], w=w[h * om.rows:], cols=cols,
                      R_in=eng.tsqr(first, rhs=rhs[: h * om.rows], w=w[: h * om.rows], cols=cols))
        return R, Rc, Rs, eng.tsqr_work_info(S, k=2), eng.tsqr_work_info(1000000, k=2)

    eng.set_option("tsqr_group_min_samples", 1)
    R1, Rc1, Rs1, wi1, big1 = run()
    eng.set_option("tsqr_groups", 0)
    R0, Rc0, Rs0, wi0, big0 = run()
    for R in (R0, R1):
        assert np.all(np.tril(R, -1) == 0) and np.linalg.norm(R.T @ R - Go) <= 1e-11 * np.linalg.norm(Go)
    norm = lambda R: R * np.where(np.diag(R) < 0, -1.0, 1.0)[:, None]
    Gc = Go[np.ix_(sel, sel)]
    for R in (Rc0, Rc1, Rs0, Rs1):
        assert np.linalg.norm(R.T @ R - Gc) <= 1e-11 * np.linalg.norm(Gc)
        assert np.linalg.norm(norm(R) - norm(Rc0)) <= 1e-9 * np.linalg.norm(Rc0)
    if cfg[0] == "walkman_apriori":   # branched tree: legs / arms / head / waist / base rows are separate factorisations
        assert big1["flop"] < 0.6 * big0["flop"]
    elif cfg[1]:                      # one chain on a floating base: the force rows of the base wrench are a group of their own, over the
        assert big1["flop"] < big0["flop"]   # columns that have a force (option tsqr_force_group)
        eng.set_option("tsqr_groups", 1)
        eng.set_option("tsqr_force_group", 0)
        assert eng.tsqr_work_info(1000000, k=2)["flop"] == big0["flop"]   # without it: one group, the single factorisation
    else:                             # one chain, fixed base: one group, the single factorisation
        assert big1["flop"] == big0["flop"]


@pytest.mark.parametrize("cfg", [CONFIGS[7], CONFIGS[6]], ids=cfg_id)
def test_tsqr_row_mask_skips_masked_rows(cfg):
    """Base-wrench-only identification (identifier.py:629-636: only the 6 base rows of every sample enter the fit) is a 0/1 row
    weight: rows that no sample weights are left out of the factorisation altogether (their groups are not formed).  Same R^T R as
    the weighted Gram, also for a mask that keeps single joint rows, and the same sign-normalised R as the unskipped path."""
    t, eng, om = _engine_oracle(cfg)
    eng.set_option("tsqr_group_min_samples", 1)
    S = 700
    st, rng = _states(t, cfg, S, 41)
    rhs = rng.standard_normal((S * om.rows, 1))
    A0 = _aug(om, st, rhs)
    import scipy.linalg as sla

    for keep in (list(range(6)), list(range(6)) + [6 + 2, om.rows - 1]):
        w = np.zeros((S, om.rows))
        w[:, keep] = 1.0 + rng.random((S, len(keep)))
        w = w.reshape(-1)
        A = A0 * w[:, None]
        Go = A.T @ A
        R = eng.tsqr(st, rhs=rhs, w=w)
        assert np.all(np.tril(R, -1) == 0) and np.linalg.norm(R.T @ R - Go) <= 1e-11 * np.linalg.norm(Go)
        G = eng.gram(st, rhs=rhs, w=w)
        assert np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go)
        ncol = min(60, int(np.linalg.matrix_rank(Go[: om.P, : om.P])) - 3)
        cols = np.sort(sla.qr(Go[: om.P, : om.P], pivoting=True, mode="r")[1][:ncol]).astype(np.int32)
        Rc = eng.tsqr(st, rhs=rhs, w=w, cols=cols)
        eng.set_option("tsqr_groups", 0)
        Rc0 = eng.tsqr(st, rhs=rhs, w=w, cols=cols)
        eng.set_option("tsqr_groups", 1)
        norm = lambda R: R * np.where(np.diag(R) < 0, -1.0, 1.0)[:, None]
        assert np.linalg.norm(norm(Rc) - norm(Rc0)) <= 1e-9 * np.linalg.norm(Rc0)


@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[7]], ids=cfg_id)
def test_fd_sweep_scores_match_oracle_regressors(cfg):
    """fbr_fd_scores == sum(W_t * Y(state_t + eps e_d)) with the oracle's regressor on every perturbed state
    (the per-sample worker of analyticalGradient.py:92-185)."""
    from flobaroid_amd import excitation as exc

    t, eng, om = _engine_oracle(cfg)
    S, eps = 9, 1e-6
    st, rng = _states(t, cfg, S, 51)
    n = t.num_dofs
    W = rng.standard_normal((S * om.rows, om.P))
    sc = eng.fd_scores(st, W, eps)
    assert sc.shape == (S, 1 + 3 * n)
    Wb = W.reshape(S, om.rows, om.P)
    ref = np.empty_like(sc)
    ref[:, 0] = np.einsum("src,src->s", Wb, om.regressor(st, st["sign"]).reshape(S, om.rows, om.P))
    for kind, key in enumerate(("q", "dq", "ddq")):
        for d in range(n):
            sp = {k: v.copy() for k, v in st.items()}
            sp[key][:, d] += eps
            ref[:, 1 + kind * n + d] = np.einsum("src,src->s", Wb, om.regressor(sp, st["sign"]).reshape(S, om.rows, om.P))
    assert np.abs(sc - ref).max() <= 1e-11 * np.abs(ref).max()
    sq, sdq, sddq = exc.dopt_sensitivities(eng, st, W, eps)
    assert sq.shape == sdq.shape == sddq.shape == (S, n)
    # (differences of nearly equal scores: compare through the scores' own scale)
    assert np.abs(sq - (ref[:, 1:1 + n] - ref[:, :1]) / eps).max() <= 1e-10 * np.abs(ref).max() / eps
    # regressor is linear in ddq: the acceleration sensitivity is exact, independent of eps
    sc2 = eng.fd_scores(st, W, 1e-3)
    assert np.abs((sc2[:, 1 + 2 * n:] - sc2[:, :1]) / 1e-3 - sddq).max() <= 1e-6 * np.abs(sddq).max()


@pytest.mark.parametrize("n", [5, 16, 33, 100, 128, 129, 200, 256, 300, 384, 400, 512, 520, 640, 700, 768])
def test_tsqr_merge_every_kernel_instantiation(n):
    """fbr_tsqr_merge for every factor width: wave-private kernels (1..8 column tiles) and the wave-pipelined kernels
    (1..6 tiles per wave, 64 / 48 / 32-row blocks): Rm^T Rm = R1^T R1 + R2^T R2, Rm upper triangular.
    One of the inputs is rank deficient (zero diagonal entries and dependent columns)."""
    from flobaroid_amd._lib import Engine

    t = load_topo("threeLinks")
    eng = Engine(t, floating=True)
    rng = np.random.default_rng(1000 + n)
    R1 = np.triu(rng.standard_normal((n, n)))
    R2 = np.triu(rng.standard_normal((n, n)))
    if n > 4:
        R2[:, n // 2] = R2[:, n // 3]          # dependent column
        R2[n // 2:, n // 2] = 0.0
        R1[:, 1] = 0.0                          # zero column / zero pivot
    Rm = eng.tsqr_merge(R1, R2)
    assert np.all(np.tril(Rm, -1) == 0.0) and np.all(np.isfinite(Rm))
    G = R1.T @ R1 + R2.T @ R2
    assert np.linalg.norm(Rm.T @ Rm - G) <= 1e-12 * np.linalg.norm(G)


@pytest.mark.parametrize("ncols,friction", [(100, 0), (140, 0), (300, 0), (0, 1)])
def test_tsqr_widths_on_many_workers(ncols, friction):
    """Level 0 on many workgroups + tree for the factor widths the robots do not reach by themselves (column subsets of
    WALK-MAN: 7 / 9 / 19 tiles; all columns with friction: 36 tiles): R^T R == the fused Gram restricted to the columns."""
    from flobaroid_amd._lib import Engine

    t = load_topo("walkman_apriori")
    eng = Engine(t, floating=True, friction=bool(friction))
    rng = np.random.default_rng(500 + ncols)
    S = 6000
    st = random_states(t, S, rng, 1)
    st["sign"] = np.tanh(st["dq"] / 0.02)
    rhs = rng.standard_normal((S * eng.rows, 1))
    G = eng.gram(st, rhs=rhs)
    if ncols:
        cols = np.sort(rng.choice(eng.cols, size=ncols, replace=False)).astype(np.int32)
        R = eng.tsqr(st, rhs=rhs, cols=cols)
        idx = np.r_[cols, eng.cols]
        Gs = G[np.ix_(idx, idx)]
    else:
        R = eng.tsqr(st, rhs=rhs)
        Gs = G
    assert np.all(np.tril(R, -1) == 0.0)
    assert np.linalg.norm(R.T @ R - Gs) <= 1e-11 * np.linalg.norm(Gs)


@pytest.mark.parametrize("tag", ["gwA", "gwB"])
def test_dopt_sensitivities_match_the_reference_worker(tag):
    """excitation.dopt_sensitivities (fbr_fd_scores) against the sens_q / sens_dq / sens_ddq that the reference's own
    _dopt_gradient_worker_func (excitation/analyticalGradient.py:46-185) returned on the same samples and weights
    (tools/make_fixtures.py; its iDynTree calls answered by the oracle).  The worker evaluates a floating base at the
    identity pose with zero twist and zero base acceleration; finite differences with eps = 1e-6 amplify rounding by
    1/eps, hence the tolerance relative to |score| / eps."""
    import json

    from common import GOLDEN
    from flobaroid_amd import excitation as exc
    from flobaroid_amd._lib import Engine

    z = np.load(os.path.join(GOLDEN, "ref_compute_regressors.npz"), allow_pickle=True)
    meta = json.loads(str(z[tag + "_meta"]))
    t = load_topo(meta["robot"])
    eng = Engine(t, floating=bool(meta["floating"]))
    S = meta["S"]
    st = {"q": z[tag + "_q"], "dq": z[tag + "_dq"], "ddq": z[tag + "_ddq"]}
    if meta["floating"]:
        st.update(base_vel=np.zeros((S, 6)), base_acc=np.zeros((S, 6)), rpy=np.zeros((S, 3)))
    W = z[tag + "_W"]
    sq, sdq, sddq = exc.dopt_sensitivities(eng, st, W, meta["eps"], W_visc=z[tag + "_W_visc"], reference_state_carryover=True)
    scale = np.abs(eng.fd_scores(st, W, meta["eps"])).max() / meta["eps"]
    for got, name in ((sq, "sens_q"), (sdq, "sens_dq"), (sddq, "sens_ddq")):
        want = z["%s_%s" % (tag, name)]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-10 * scale, name
    # without the reference's state carry-over the acceleration sensitivities differ by exactly sens_dq_inertial[:, n-1]
    _, sdq0, sddq0 = exc.dopt_sensitivities(eng, st, W, meta["eps"])
    assert np.abs((sddq - sddq0) - sdq0[:, -1:]).max() <= 1e-9 * scale


@pytest.mark.parametrize("S", [0, 1, 2, 5])
def test_tiny_batches(S):
    """Edge sizes: empty, one sample, fewer samples than workers -- every reduction entry point."""
    cfg = CONFIGS[7]
    t, eng, om = _engine_oracle(cfg)
    st, rng = _states(t, cfg, S, 61)
    rhs = rng.standard_normal((S * om.rows, 1))
    Pa = om.P + 1
    if S == 0:
        assert np.all(eng.gram(st, rhs=rhs) == 0.0) and np.all(eng.tsqr(st, rhs=rhs) == 0.0)
        assert eng.regressor(st).shape == (0, om.P) and eng.fd_scores(st, np.zeros((0, om.P)), 1e-6).shape == (0, 1 + 3 * t.num_dofs)
        return
    A = _aug(om, st, rhs)
    Go = A.T @ A
    G = eng.gram(st, rhs=rhs)
    assert G.shape == (Pa, Pa) and np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go)
    R = eng.tsqr(st, rhs=rhs)
    assert np.all(np.tril(R, -1) == 0.0) and np.linalg.norm(R.T @ R - Go) <= 1e-11 * np.linalg.norm(Go)
    Gg = eng.gram_grouped(st, S, rhs=rhs)
    for g in range(S):
        Ag = A[g * om.rows:(g + 1) * om.rows]
        assert np.linalg.norm(Gg[g] - Ag.T @ Ag) <= 1e-11 * np.linalg.norm(Ag.T @ Ag)


def test_tsqr_badly_scaled_and_tiny_columns():
    """Columns 12 orders of magnitude apart and a column in the denormal range: no NaN / Inf, R^T R == G column-wise."""
    from flobaroid_amd._lib import Engine

    eng = Engine(load_topo("threeLinks"), floating=True)
    rng = np.random.default_rng(71)
    n = 40
    sc = 10.0 ** rng.uniform(-6, 6, n)
    sc[7] = 1e-170          # squares to the denormal range
    R1 = np.triu(rng.standard_normal((n, n))) * sc[None, :]
    R2 = np.triu(rng.standard_normal((n, n))) * sc[None, :]
    Rm = eng.tsqr_merge(R1, R2)
    assert np.all(np.isfinite(Rm))
    G = R1.T @ R1 + R2.T @ R2
    D = 1.0 / np.sqrt(np.maximum(np.diag(G), 1e-300))
    keep = np.arange(n) != 7
    E = (Rm.T @ Rm - G) * D[:, None] * D[None, :]
    assert np.abs(E[np.ix_(keep, keep)]).max() <= 1e-12


@pytest.mark.parametrize("seed,L,branch,floating,fric,sym", [
    (11, 14, 0.5, 1, 1, 0), (12, 30, 0.3, 1, 1, 1), (13, 44, 0.7, 0, 1, 0), (14, 55, 0.0, 1, 0, 1), (15, 60, 0.4, 0, 1, 1),
])
def test_random_trees_all_entry_points(seed, L, branch, floating, fric, sym):
    """Random kinematic trees (nothing tuned to the bundled robots): regressor, fused Gram (both kernel shapes), TSQR and
    prediction against the oracle."""
    from common import random_topology
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(seed)
    t = random_topology(rng, L, p_fixed=0.0 if branch == 0.0 else 0.25, branchiness=branch)
    if t.num_dofs + (6 if floating else 0) > 60:
        pytest.skip("more than 60 regressor rows")
    om = OracleModel(t, floating=bool(floating), fric=bool(fric), fric_sym=bool(sym))
    S = 300
    st = random_states(t, S, rng, floating)
    st["sign"] = np.tanh(st["dq"] / 0.02)
    Yo = om.regressor(st, st["sign"])
    rhs = rng.standard_normal((Yo.shape[0], 2))
    A = np.hstack([Yo, rhs])
    Go = A.T @ A
    for shape in ("two", "one"):
        eng = Engine(t, floating=bool(floating), friction=bool(fric), friction_symmetric=bool(sym), options={"gram_shape": 2 if shape == "two" else 1})
        G = eng.gram(st, rhs=rhs)
        assert np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go), shape
        if shape == "two":
            Y = eng.regressor(st)
            assert np.abs(Y - Yo).max() <= 1e-11 * np.abs(Yo).max()
            R = eng.tsqr(st, rhs=rhs)
            assert np.linalg.norm(R.T @ R - Go) <= 1e-10 * np.linalg.norm(Go)
            x = rng.standard_normal(om.P)
            tau = eng.predict(st, x)
            assert np.abs(tau.reshape(-1) - Yo @ x).max() <= 1e-10 * np.abs(Yo @ x).max()
        eng.close()


@pytest.mark.parametrize("name,fl,fric,S", [("kuka_lwr4", 0, 1, 50_000), ("walkman_left_arm", 1, 1, 500_000), ("walkman_apriori", 1, 0, 1_000_000)])
def test_baseline_sizes_through_size_independent_properties(name, fl, fric, S):
    """BASELINE.json configs 2-4 at their full sample counts (the CPU oracle cannot get there inside a test): the fused Gram
    must satisfy, for random x,  x^T G_YY x = |Y x|^2  and  x^T G_Y,tau = (Y x) . tau  with Y x from the streaming prediction
    kernel (a different kernel, itself pinned on the oracle above), additivity over a split of the samples (the accumulate
    path) and symmetry."""
    from flobaroid_amd._lib import Engine

    t = load_topo(name)
    eng = Engine(t, floating=bool(fl), friction=bool(fric))
    rng = np.random.default_rng(2024)
    st = random_states(t, S, rng, fl, use_limits=True)
    st["sign"] = np.tanh(st["dq"] / 0.02)
    tau = rng.standard_normal((S * eng.rows, 1))
    P = eng.cols
    G = eng.gram(st, rhs=tau)
    assert G.shape == (P + 1, P + 1) and np.all(np.isfinite(G))
    assert np.abs(G - G.T).max() <= 1e-13 * np.abs(G).max()
    assert abs(G[P, P] - float(tau[:, 0] @ tau[:, 0])) <= 1e-11 * G[P, P]
    for _ in range(3):
        x = rng.standard_normal(P)
        yx = eng.predict(st, x).reshape(-1)
        assert abs(x @ G[:P, :P] @ x - yx @ yx) <= 1e-10 * (yx @ yx)
        assert abs(x @ G[:P, P] - yx @ tau[:, 0]) <= 1e-10 * np.linalg.norm(yx) * np.linalg.norm(tau)
    h = S // 3
    G1 = eng.gram({k: v[:h] for k, v in st.items()}, rhs=tau[: h * eng.rows])
    G2 = eng.gram({k: v[h:] for k, v in st.items()}, rhs=tau[h * eng.rows:], out=G1.copy(), accumulate=True)
    assert np.linalg.norm(G2 - G) <= 1e-12 * np.linalg.norm(G)
    # the Householder TSQR factor of the same stream (config 5's route to the SDP inputs): R^T R = G, R upper triangular
    R = eng.tsqr(st, rhs=tau)
    assert np.all(np.tril(R, -1) == 0.0) and np.all(np.isfinite(R))
    assert np.linalg.norm(R.T @ R - G) <= 1e-11 * np.linalg.norm(G)
    eng.close()


@pytest.mark.parametrize("cfg", [CONFIGS[3], CONFIGS[7]], ids=cfg_id)
def test_gram_submit_pipelines_calls_and_matches_the_blocking_path(cfg):
    """fbr_gram_submit / fbr_wait: passes enqueued back to back (two in flight, the producer of one beside the last Gram launches of
    the one before) give bit for bit the Grams of the blocking calls; a blocking entry point in between waits for them; waiting for
    an old ticket or twice is harmless; host inputs are refused."""
    import torch

    from flobaroid_amd._lib import FbrError

    t, eng, om = _engine_oracle(cfg)
    S = 9000 if t.num_links < 10 else 2600   # several chunks each
    sets = []
    for i in range(5):
        st, rng = _states(t, cfg, S + 17 * i, 100 + i)
        rhs = rng.standard_normal((st["q"].shape[0] * om.rows, 1))
        sets.append(({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}, torch.from_numpy(rhs).cuda()))
    want = [eng.gram(st, rhs=rhs).clone() for st, rhs in sets]
    Pa = om.P + 1
    outs = [torch.full((Pa, Pa), float("nan"), dtype=torch.float64, device="cuda") for _ in sets]
    tickets = [eng.gram_submit(st, outs[i], rhs=rhs) for i, (st, rhs) in enumerate(sets[:3])]
    assert tickets == sorted(tickets) and len(set(tickets)) == 3
    eng.wait(tickets[0])
    assert torch.equal(outs[0], want[0])
    Y = eng.regressor({k: v[:5] for k, v in sets[0][0].items()})  # blocking call: runs after every submission
    assert torch.equal(outs[1], want[1]) and torch.equal(outs[2], want[2]) and Y.shape == (5 * om.rows, om.P)
    t3 = eng.gram_submit(sets[3][0], outs[3], rhs=sets[3][1])
    t4 = eng.gram_submit(sets[4][0], outs[4], rhs=sets[4][1])
    eng.wait()
    eng.wait(t3)
    eng.wait(t4)
    assert torch.equal(outs[3], want[3]) and torch.equal(outs[4], want[4])
    # accumulate across submissions: the sum of two passes
    acc = torch.zeros((Pa, Pa), dtype=torch.float64, device="cuda")
    eng.gram_submit(sets[0][0], acc, rhs=sets[0][1])
    eng.gram_submit(sets[1][0], acc, rhs=sets[1][1], accumulate=True)
    eng.wait()
    ref = (want[0] + want[1]).cpu().numpy()
    assert np.linalg.norm(acc.cpu().numpy() - ref) <= 1e-13 * np.linalg.norm(ref)
    with pytest.raises((FbrError, ValueError)):   # pageable host inputs are refused
        eng.gram_submit({k: v.cpu().numpy() for k, v in sets[0][0].items()}, outs[0], rhs=sets[0][1].cpu().numpy())
    # pinned host inputs: staged chunk by chunk on the copy stream, also across the two submissions in flight
    pinned = [({k: v.cpu().pin_memory() for k, v in st.items()}, rhs.cpu().pin_memory()) for st, rhs in sets[:3]]
    outs_h = [torch.full((Pa, Pa), float("nan"), dtype=torch.float64, device="cuda") for _ in pinned]
    for i, (st, rhs) in enumerate(pinned):
        eng.gram_submit(st, outs_h[i], rhs=rhs)
    eng.wait()
    # (pinned inputs are staged chunk by chunk and take the per-sample-image pass; device-resident ones the sample-contiguous pass of
    # option gram_lane: two summation orders, so bit for bit against a blocking call on the same pinned inputs, to rounding against want)
    for i in range(3):
        blocking = eng.gram(pinned[i][0], rhs=pinned[i][1])
        assert np.array_equal(outs_h[i].cpu().numpy(), blocking.cpu().numpy() if torch.is_tensor(blocking) else np.asarray(blocking))
        assert float(torch.linalg.norm(outs_h[i] - want[i])) <= 1e-13 * float(torch.linalg.norm(want[i]))
    A = np.hstack([om.regressor({k: v.cpu().numpy() for k, v in sets[2][0].items()}, sets[2][0]["sign"].cpu().numpy()), sets[2][1].cpu().numpy()])
    assert np.linalg.norm(outs[2].cpu().numpy() - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A)


@pytest.mark.parametrize("name,fl", [("kuka_lwr4", 0), ("walkman_apriori", 1)])
def test_any_link_and_dof_serialisation(name, fl):
    """include/fbr.h: links and DOFs may be serialised in ANY order (parents need not precede children).  A shuffled serialisation of
    the same robot gives the same regressor / Gram / TSQR factor under the row and column permutation, and matches the oracle run on
    the shuffled topology itself."""
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    t = load_topo(name)
    rng = np.random.default_rng(17)
    lperm = rng.permutation(t.num_links)
    dperm = rng.permutation(t.num_dofs)
    t2 = t.reordered_links([t.link_names[i] for i in lperm]).reordered_dofs([t.dof_names[i] for i in dperm])
    assert any(p > l for l, p in enumerate(t2.parent))  # some child is numbered before its parent
    S = 300
    st = random_states(t, S, rng, fl, use_limits=True)
    st2 = {k: (v[:, dperm] if v.shape[1] == t.num_dofs else v) for k, v in st.items()}
    e1, e2 = Engine(t, floating=bool(fl), friction=True), Engine(t2, floating=bool(fl), friction=True)
    sg, sg2 = np.tanh(st["dq"] / 0.02), np.tanh(st2["dq"] / 0.02)
    st["sign"], st2["sign"] = sg, sg2
    fb = 6 * fl
    n, L = t.num_dofs, t.num_links
    rows = np.concatenate([np.arange(fb), fb + dperm])                       # row i of the shuffled model = row rows[i] of the original
    cols = np.concatenate([np.concatenate([10 * l + np.arange(10) for l in lperm])] + [10 * L + n * k + dperm for k in range(3)])
    Y1 = e1.regressor(st).reshape(S, fb + n, -1)
    Y2 = e2.regressor(st2).reshape(S, fb + n, -1)
    assert np.abs(Y2 - Y1[:, rows][:, :, cols]).max() <= 1e-11 * np.abs(Y1).max()
    Yo = OracleModel(t2, floating=bool(fl), fric=True, fric_sym=True).regressor(st2, sg2).reshape(S, fb + n, -1)
    assert np.abs(Y2 - Yo).max() <= 1e-11 * np.abs(Yo).max()
    G1, G2 = e1.gram(st), e2.gram(st2)
    assert np.linalg.norm(G2 - G1[np.ix_(cols, cols)]) <= 1e-11 * np.linalg.norm(G1)
    R2 = e2.tsqr(st2)
    assert np.linalg.norm(R2.T @ R2 - G2) <= 1e-11 * np.linalg.norm(G2)
    e1.close()
    e2.close()


def test_gram_of_a_robot_beyond_60_rows_per_sample_comes_from_the_tsqr_factor():
    """More than 60 regressor rows per sample (54 DOF on a floating base) is outside the fused Gram's tile program (15 MFMA k-steps):
    ``fbr_gram_accumulate`` then forms G = R^T R from the Householder factor of the same rows, so every caller keeps working for any URDF
    the reference loads (model.py:116-168); the asynchronous and the grouped form name the limit instead."""
    import torch

    from common import random_topology
    from flobaroid_amd._lib import Engine, FbrError
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(77)
    t = random_topology(rng, 62, p_fixed=0.0, branchiness=0.4)
    assert t.num_dofs + 6 > 60
    eng = Engine(t, floating=True)
    om = OracleModel(t, floating=True)
    S = 300
    st = random_states(t, S, rng, 1)
    Y = om.regressor(st)
    tau = rng.standard_normal((Y.shape[0], 2))
    w = rng.random(Y.shape[0]) + 0.5
    A = np.hstack([Y, tau]) * w[:, None]
    Go = A.T @ A
    G = eng.gram(st, rhs=tau, w=w)
    assert np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go)
    G2 = eng.gram(st, rhs=tau, w=w, out=G.copy(), accumulate=True)
    assert np.linalg.norm(G2 - 2 * Go) <= 1e-11 * np.linalg.norm(Go)
    assert np.abs(eng.regressor(st) - Y).max() <= 1e-11 * np.abs(Y).max()
    dst = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    with pytest.raises(FbrError, match="60 rows"):
        eng.gram_submit(dst, torch.zeros((eng.cols, eng.cols), dtype=torch.float64, device="cuda"))
    with pytest.raises(FbrError, match="60 rows"):
        eng.gram_grouped(st, 3)
    eng.close()


@pytest.mark.parametrize("name", ["threeLinks", "kuka_lwr4", "walkman_left_arm", "walkman_apriori"])
def test_hip_against_idyntree_outputs(name):
    """The HIP path against iDynTree's own outputs (tests/golden/idyntree_<robot>.npz, written by tools/pin_idyntree.py where the
    reference's environment exists; skipped while the files are not committed): regressor, inverse dynamics and J^T w on the seeded
    states, fixed and floating base -- the same comparison tests/test_oracle.py makes for the oracle, so that a green run here does not
    rest on the oracle at all."""
    from flobaroid_amd._lib import Engine
    from test_oracle import idyntree_fixture

    fx = idyntree_fixture(name)
    t = load_topo(name)
    assert list(fx["link_names"]) == list(t.link_names) and list(fx["dof_names"]) == list(t.dof_names)
    S = fx["fb1_q"].shape[0]
    for fl in (0, 1):
        st = {k: fx[f"fb{fl}_{k}"] for k in ("q", "dq", "ddq")}
        if fl:
            st.update({k: fx[f"fb1_{k}"] for k in ("base_vel", "base_acc", "rpy")})
        eng = Engine(t, floating=bool(fl))
        Yi = fx[f"fb{fl}_Y"][:, (0 if fl else 6):, :].reshape(S * eng.rows, eng.cols)
        ti = fx[f"fb{fl}_tau"][:, (0 if fl else 6):]
        assert np.abs(eng.regressor(st) - Yi).max() <= 1e-9 * np.abs(Yi).max()
        assert np.abs(eng.inverse_dynamics(st, t.x_std()) - ti).max() <= 1e-9 * np.abs(ti).max()
        if fl and bool(fx["fb1_have_frame"]):
            w = np.random.default_rng(7).standard_normal((S, 6))
            ci = np.einsum("sij,si->sj", fx["fb1_J"], w)
            assert np.abs(eng.contact_torques(st, str(fx["frame"]), w) - ci).max() <= 1e-9 * np.abs(ci).max()
        eng.close()


@pytest.mark.parametrize("seed,L,floating,fric", [(21, 9, 1, 0), (22, 16, 0, 1), (23, 24, 1, 1), (24, 33, 1, 0), (25, 12, 0, 0)])
def test_prismatic_joints_all_entry_points(seed, L, floating, fric):
    """Random trees in which half of the movable joints are PRISMATIC (fbr_topology.joint_type; iDynTree's loader takes any URDF,
    model.py:60-67): every per-sample entry point and both reductions against the oracle, whose prismatic arithmetic is pinned on two
    independent formulations and a power balance (tests/test_oracle.py)."""
    from common import random_topology
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(seed)
    t = random_topology(rng, L, p_fixed=0.25, branchiness=0.4, p_prismatic=0.5)
    assert any(j == 2 for j in t.joint_type)
    if t.num_dofs + (6 if floating else 0) > 60:
        pytest.skip("more than 60 regressor rows")
    om = OracleModel(t, floating=bool(floating), fric=bool(fric))
    eng = Engine(t, floating=bool(floating), friction=bool(fric))
    S = 350
    st = random_states(t, S, rng, floating)
    st["sign"] = np.tanh(st["dq"] / 0.02)
    Yo = om.regressor(st, st["sign"])
    assert np.abs(eng.regressor(st) - Yo).max() <= 1e-11 * np.abs(Yo).max()
    x = np.concatenate([t.x_std(), rng.random(om.P - 10 * t.num_links + 4 * t.num_dofs)])  # (+ friction slots)
    to = om.inverse_dynamics(st, x, st["sign"])
    assert np.abs(eng.inverse_dynamics(st, x) - to).max() <= 1e-11 * np.abs(to).max()
    xi = rng.standard_normal(om.P)
    assert np.abs(eng.predict(st, xi).reshape(-1) - Yo @ xi).max() <= 1e-10 * np.abs(Yo @ xi).max()
    rhs = rng.standard_normal((Yo.shape[0], 1))
    w = 0.5 + rng.random(Yo.shape[0])
    A = np.hstack([Yo, rhs]) * w[:, None]
    Go = A.T @ A
    assert np.linalg.norm(eng.gram(st, rhs=rhs, w=w) - Go) <= 1e-11 * np.linalg.norm(Go)
    R = eng.tsqr(st, rhs=rhs, w=w)
    assert np.all(np.tril(R, -1) == 0) and np.linalg.norm(R.T @ R - Go) <= 1e-10 * np.linalg.norm(Go)
    link = t.num_links - 1
    wr = rng.standard_normal((S, 6))
    co = om.contact_torques(st, t.link_names[link], wr)
    assert np.abs(eng.contact_torques(st, t.link_names[link], wr) - co).max() <= 1e-11 * np.abs(co).max()
    # what the reductions run on: 4 columns for a link behind a prismatic joint, 7 behind a revolute one, 10 for the base link
    info = eng.link_merge_info()
    nfix = sum(1 for l in range(1, t.num_links) if t.dof_index[l] < 0)
    npris = sum(1 for j in t.joint_type if j == 2)
    nrev = t.num_links - 1 - nfix - npris
    if eng.get_option("link_merge"):  # (the "allcols" mode of this module switches the reductions off)
        assert info["reduced_cols"] == 10 + 7 * nrev + 4 * npris + (om.P - 10 * t.num_links)
    else:
        assert info["reduced_cols"] == info["cols"]
    eng.close()


def test_rhs_moments_of_a_tile_without_rows_repeat():
    """The robot of tests/test_emul.py::test_rhs_moments_of_a_tile_without_rows (fixed base, a link welded to it: the first tile has no
    image rows): Y^T tau from the packer's moments in all three kernel-shape settings, 15 repetitions each -- before round 5 two
    reduction workgroups raced on one entry of G here and a column's product with tau was lost now and then."""
    from common import random_topology
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(22)
    t = random_topology(rng, 16, p_fixed=0.25, branchiness=0.4)
    om = OracleModel(t, floating=False)
    S = 350
    st = random_states(t, S, rng, False)
    Yo = om.regressor(st)
    rhs = rng.standard_normal((Yo.shape[0], 1))
    A = np.hstack([Yo, rhs])
    Go = A.T @ A
    for shape in (0, 1, 2):
        eng = Engine(t, floating=False, options={"gram_shape": shape})
        for rep in range(15):
            G = eng.gram(st, rhs=rhs)
            assert np.linalg.norm(G - Go) <= 1e-11 * np.linalg.norm(Go), (shape, rep)
            assert np.array_equal(G, G.T)
        eng.close()
