"""Link merging and regrouping (fbr.h fbr_model_link_merge_info): the reductions run on 7 columns per moving body (10 for the base link)
and expand -- against the unmerged path (option link_merge = 0), the merged-only path (regroup = 0) and the oracle, on robots with chains of fixed links, every base / friction / gravity-only mode, weights, R_in,
accumulation, host and device memory, blocking calls and submissions."""
import numpy as np
import pytest

from common import random_states, random_topology

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reductions_whatever_the_size(engine_options):
    """This module is about the reduced paths: they are taken at the few hundred samples the oracle can check (the library's own
    threshold is looked at in test_small_batches_skip_the_reductions, which sets the option itself)."""
    with engine_options(reduce_min_work=0):
        yield


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("seed,L,floating,fric,grav", [(1, 14, 1, 0, 0), (2, 22, 0, 1, 0), (3, 30, 1, 1, 0), (4, 18, 0, 0, 1), (5, 40, 1, 0, 0),
                                                       (6, 9, 1, 1, 1)])
def test_merged_equals_unmerged_and_oracle(seed, L, floating, fric, grav):
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(100 + seed)
    t = random_topology(rng, L, p_fixed=0.5, branchiness=0.4)  # half of the joints fixed: fixed chains, fixed leaves, moving links on fixed ones
    if t.num_dofs + (6 if floating else 0) > 60 or t.num_dofs == 0:
        pytest.skip("row count outside the fused kernels")
    om = OracleModel(t, floating=bool(floating), fric=bool(fric), fric_sym=True, grav_only=bool(grav))
    S = 700
    st = random_states(t, S, rng, floating)
    st["sign"] = np.tanh(st["dq"] / 0.02)
    Yo = om.regressor(st, st["sign"])
    k = 2
    rhs = rng.standard_normal((Yo.shape[0], k))
    w = 0.5 + rng.random(Yo.shape[0])
    eng = Engine(t, floating=bool(floating), friction=bool(fric), friction_symmetric=True, gravity_only=bool(grav))
    info = eng.link_merge_info()
    nfixed = sum(1 for l in range(1, t.num_links) if t.dof_index[l] < 0)
    if grav:  # (m, h) of a fixed link are not combinations of the (m, h) columns of its body alone: nothing merged
        assert info["moving_links"] == t.num_links and info["reduced_cols"] == info["cols"]
    else:
        # 7 columns per link behind a joint, 10 for the base link: the reductions run on a full-rank set of columns
        nrev = t.num_links - 1 - nfixed
        assert info["moving_links"] == t.num_links - nfixed and info["reduced_cols"] == 10 + 7 * nrev + (info["cols"] - 10 * t.num_links)
    for wt in (None, w):
        A = np.hstack([Yo, rhs]) * (1.0 if wt is None else wt[:, None])
        Go = A.T @ A
        G = eng.gram(st, rhs=rhs, w=wt)
        R = eng.tsqr(st, rhs=rhs, w=wt)
        eng.set_option("link_merge", 0)
        assert eng.link_merge_info()["reduced_cols"] == info["cols"]
        Gp = eng.gram(st, rhs=rhs, w=wt)
        Rp = eng.tsqr(st, rhs=rhs, w=wt)
        eng.set_option("link_merge", 1)
        if not grav:  # fixed links merged, revolute links not regrouped: the middle model, which also takes the short factorisations
            eng.set_option("regroup", 0)
            assert eng.link_merge_info()["reduced_cols"] == 10 * (t.num_links - nfixed) + (info["cols"] - 10 * t.num_links)
            Gm = eng.gram(st, rhs=rhs, w=wt)
            Rm = eng.tsqr(st, rhs=rhs, w=wt)
            eng.set_option("regroup", 1)
            assert _rel(Gm, Go) <= 1e-12 and _rel(Rm.T @ Rm, Go) <= 1e-12
            # the regrouped model's factorisation by row groups at this size, with both writers
            eng.set_option("tsqr_group_min_samples", 1)
            for writer in (8, 16, 32):  # (32: rows staged in the LDS, streamed out in 16-byte pieces)
                eng.set_option("tsqr_writer", writer)
                Rg = eng.tsqr(st, rhs=rhs, w=wt)
                assert _rel(Rg.T @ Rg, Go) <= 1e-12 and np.all(np.tril(Rg, -1) == 0), writer
            eng.set_option("tsqr_writer", 0)
            eng.set_option("tsqr_group_min_samples", 24000)
        assert _rel(G, Go) <= 1e-12 and _rel(Gp, Go) <= 1e-12 and _rel(G, Gp) <= 1e-13
        assert np.array_equal(G, G.T)
        assert np.all(np.tril(R, -1) == 0)
        assert _rel(R.T @ R, Go) <= 1e-12 and _rel(Rp.T @ Rp, Go) <= 1e-12
    # accumulation and a streamed factor: two halves == the whole
    h = S // 2
    a = {kk: v[:h] for kk, v in st.items()}
    b = {kk: v[h:] for kk, v in st.items()}
    r = om.rows
    A = np.hstack([Yo, rhs])
    Go = A.T @ A
    G2 = eng.gram(a, rhs=rhs[: h * r])
    G2 = eng.gram(b, rhs=rhs[h * r:], out=G2, accumulate=True)
    assert _rel(G2, Go) <= 1e-12
    R2 = eng.tsqr(b, rhs=rhs[h * r:], R_in=eng.tsqr(a, rhs=rhs[: h * r]))
    assert _rel(R2.T @ R2, Go) <= 1e-12 and np.all(np.tril(R2, -1) == 0)
    eng.close()


def test_walkman_merged_submissions_are_bitwise_the_blocking_result():
    """WALK-MAN (18 of 48 links fixed): device-resident blocking calls and two-in-flight submissions of the merged path return the
    same bits; against the unmerged path to rounding."""
    import os
    import torch
    from common import load_topo
    from flobaroid_amd._lib import Engine

    t = load_topo("walkman_apriori")
    eng = Engine(t, floating=True)
    info = eng.link_merge_info()
    assert (info["links"], info["moving_links"], info["cols"], info["reduced_cols"]) == (48, 30, 480, 213)  # 10 + 29 x 7: the rank of the regressor
    eng.use_torch_stream()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    S = 6000
    sts = [{kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in random_states(t, S, rng, True).items()} for _ in range(3)]
    rhs = [torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev) for _ in range(3)]
    Gb = [eng.gram(s, rhs=r_) for s, r_ in zip(sts, rhs)]
    Rb = [eng.tsqr(s, rhs=r_) for s, r_ in zip(sts, rhs)]
    Go = [torch.zeros_like(Gb[0]) for _ in range(3)]
    Ro = [torch.zeros_like(Rb[0]) for _ in range(3)]
    tickets = []
    for i in range(3):  # mixed kinds, two in flight
        tickets.append(eng.gram_submit(sts[i], Go[i], rhs=rhs[i]))
        if len(tickets) > 1:
            eng.wait(tickets[-2])
        tickets.append(eng.tsqr_submit(sts[i], Ro[i], rhs=rhs[i]))
        eng.wait(tickets[-2])
    eng.wait(tickets[-1])
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(Go[i], Gb[i]) and torch.equal(Ro[i], Rb[i])
    eng.set_option("link_merge", 0)
    Gp = eng.gram(sts[0], rhs=rhs[0])
    Rp = eng.tsqr(sts[0], rhs=rhs[0])
    gn = float(torch.linalg.norm(Gp))
    assert float(torch.linalg.norm(Gb[0] - Gp)) <= 1e-13 * gn
    assert float(torch.linalg.norm(Rb[0].T @ Rb[0] - Gp)) <= 1e-12 * gn and float(torch.linalg.norm(Rp.T @ Rp - Gp)) <= 1e-12 * gn
    eng.close()


def test_walkman_tile_program_of_the_reductions():
    """What the fused pass of WALK-MAN runs: 213 columns in 15 tiles, tau's products from the packer (no dense tile), the 120
    tile pairs turned into full row segments -- ONE part; with the switches off, the programs the round started from."""
    import os
    from common import load_topo
    from flobaroid_amd._lib import Engine

    t = load_topo("walkman_apriori")

    def info(options, k=1, num_samples=-1):
        eng = Engine(t, floating=True, options=options)
        out = eng.gram_program_info(k, num_samples)
        eng.close()
        return out

    assert info({}) == {"tiles": 15, "pairs": 120, "mfma_per_sample": 272, "parts": 1}
    assert info({"gram_orient": 0})["parts"] == 2
    assert info({"gram_rhs_tile": 1}) == {"tiles": 16, "pairs": 136, "mfma_per_sample": 336, "parts": 2}
    assert info({}, k=16)["tiles"] == 16  # many rhs columns keep their dense tile
    assert info({"regroup": 0, "gram_rhs_tile": 1})["mfma_per_sample"] == 537
    assert info({"link_merge": 0, "gram_rhs_tile": 1}) == {"tiles": 33, "pairs": 561, "mfma_per_sample": 1235, "parts": 5}
    # the info calls describe the program a batch of THAT size executes: below the pay-off threshold (7.8 k samples) all 480 columns
    assert info({"gram_rhs_tile": 1, "reduce_min_work": 1e9}, num_samples=5000) == {"tiles": 33, "pairs": 561, "mfma_per_sample": 1235, "parts": 5}
    assert info({"reduce_min_work": 1e9}, num_samples=8000) == info({})


def test_small_batches_skip_the_reductions():
    """A short batch of a small robot runs over all its columns (the reduced pass costs a second model's launches and two expansion
    kernels); the result is the same either way, and a long batch takes the reductions."""
    import time
    from common import load_topo
    from flobaroid_amd._lib import Engine

    t = load_topo("kuka_lwr4")
    rng = np.random.default_rng(3)
    eng = Engine(t, floating=False)
    st = random_states(t, 3000, rng, False)
    rhs = rng.standard_normal((3000 * eng.rows, 1))
    eng.set_option("reduce_min_work", 0)
    G_forced = eng.gram(st, rhs=rhs)
    assert eng.link_merge_info(3000)["reduced_cols"] < eng.cols
    eng.set_option("reduce_min_work", 1e9)
    assert eng.link_merge_info(3000)["reduced_cols"] == eng.cols and eng.link_merge_info(10**6)["reduced_cols"] < eng.cols
    G_default = eng.gram(st, rhs=rhs)
    eng.set_option("link_merge", 0)
    G_plain = eng.gram(st, rhs=rhs)
    assert np.array_equal(G_default, G_plain)          # 3000 samples x 21 merged-away columns: not worth a second pass
    assert not np.array_equal(G_forced, G_plain) and _rel(G_forced, G_plain) <= 1e-13
    eng.close()


def test_column_subset_factor_through_the_regrouped_model():
    """fbr_tsqr_cols of a subset about as wide as the regrouped column set (WALK-MAN's 213 base columns) takes the reduced factorisation
    and expands with the subset's columns of E: the same R^T R as the direct factorisation of the subset and as the oracle, the same
    sign-normalised R (the subset has full column rank), with row weights, a streamed R_in and as a submission; a narrow subset and a
    base-wrench-only row mask keep the direct path."""
    import scipy.linalg as sla
    import torch
    from common import load_topo
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    t = load_topo("walkman_apriori")
    rng = np.random.default_rng(12)
    S = 900
    st = random_states(t, S, rng, True, use_limits=True)
    om = OracleModel(t, floating=True)
    Y = om.regressor(st)
    rhs = rng.standard_normal((Y.shape[0], 1))
    w = 0.5 + rng.random(Y.shape[0])
    eng = Engine(t, floating=True, options={"tsqr_group_min_samples": 1})
    ic = np.sort(sla.qr(eng.gram(random_states(t, 3000, rng, True, use_limits=True)), pivoting=True, mode="r")[1][:213]).astype(np.int32)
    norm = lambda R: R * np.where(np.diag(R) < 0, -1.0, 1.0)[:, None]
    for wt in (None, w):
        A = np.hstack([Y[:, ic], rhs]) * (1.0 if wt is None else wt[:, None])
        Go = A.T @ A
        wi_red = eng.tsqr_work_info(S, k=1, cols=ic)
        R = eng.tsqr(st, rhs=rhs, w=wt, cols=ic)
        eng.set_option("link_merge", 0)
        wi_dir = eng.tsqr_work_info(S, k=1, cols=ic)
        Rd = eng.tsqr(st, rhs=rhs, w=wt, cols=ic)
        eng.set_option("link_merge", 1)
        assert wi_red != wi_dir                                    # two different programs ...
        assert np.all(np.tril(R, -1) == 0) and _rel(R.T @ R, Go) <= 1e-11 and _rel(Rd.T @ Rd, Go) <= 1e-11
        assert _rel(norm(R), norm(Rd)) <= 1e-8                     # ... the same factor
        h = S // 2
        first = {k: v[:h] for k, v in st.items()}
        second = {k: v[h:] for k, v in st.items()}
        wa, wb = (None, None) if wt is None else (wt[: h * om.rows], wt[h * om.rows:])
        Rs = eng.tsqr(second, rhs=rhs[h * om.rows:], w=wb, cols=ic, R_in=eng.tsqr(first, rhs=rhs[: h * om.rows], w=wa, cols=ic))
        assert _rel(Rs.T @ Rs, Go) <= 1e-11
    # submission with device tensors == the blocking call, bit for bit
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    drhs = torch.from_numpy(rhs).cuda()
    Rb = eng.tsqr(dev, rhs=drhs, cols=ic)
    Ro = torch.zeros_like(Rb)
    eng.wait(eng.tsqr_submit(dev, Ro, rhs=drhs, cols=ic))
    assert torch.equal(Ro, Rb)
    # a subset NARROWER than the regrouped column set (180 of 214 augmented columns: the expanded rows R_red E[:, cols] are more than a working
    # factor of the subset holds -- they are folded as rows, not as a merge partner; a randomised sweep found the overflow) and a WIDER,
    # rank-deficient one (300 columns)
    for sub in (ic[:180], np.sort(rng.choice(om.P, 300, replace=False)).astype(np.int32)):
        A = np.hstack([Y[:, sub], rhs])
        assert eng.tsqr_work_info(S, k=1, cols=sub) != (eng.set_option("link_merge", 0), eng.tsqr_work_info(S, k=1, cols=sub), eng.set_option("link_merge", 1))[1]
        Rn = eng.tsqr(st, rhs=rhs, cols=sub)
        assert np.all(np.tril(Rn, -1) == 0) and _rel(Rn.T @ Rn, A.T @ A) <= 1e-11, len(sub)
    # narrow subset / masked joint rows: the direct path (same answers)
    few = ic[:60]
    A = np.hstack([Y[:, few], rhs])
    assert _rel((lambda R: R.T @ R)(eng.tsqr(st, rhs=rhs, cols=few)), A.T @ A) <= 1e-11
    mask = np.zeros((S, om.rows))
    mask[:, :6] = 1.0
    A = np.hstack([Y[:, ic], rhs]) * mask.reshape(-1)[:, None]
    assert _rel((lambda R: R.T @ R)(eng.tsqr(st, rhs=rhs, w=mask.reshape(-1), cols=ic)), A.T @ A) <= 1e-11
    eng.close()
