"""fbr_model_set_option / fbr_model_get_option (include/fbr.h): the switches and thresholds of a model handle go through the C-ABI, take
effect on the next call, and never through the environment."""
import numpy as np
import pytest

from common import load_topo, random_states

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def test_options_round_trip_and_unknown_keys():
    from flobaroid_amd._lib import Engine, FbrError

    eng = Engine(load_topo("kuka_lwr4"))
    opts = eng.options()
    assert opts["link_merge"] == 1 and opts["reduce_min_work"] == 1e9 and opts["tsqr_group_min_samples"] == 24000 and len(opts) >= 15
    eng.set_option("reduce_min_work", 0)
    assert eng.get_option("reduce_min_work") == 0
    with pytest.raises(FbrError, match="unknown option"):
        eng.set_option("no_such_switch", 1)
    with pytest.raises(FbrError, match="unknown option"):
        eng.get_option("FBR_NO_LINK_MERGE")
    eng2 = Engine(load_topo("kuka_lwr4"), options={"gram_shape": 1, "link_merge": 0})
    assert eng2.get_option("gram_shape") == 1 and eng2.get_option("link_merge") == 0 and eng.get_option("link_merge") == 1  # per handle
    eng.close()
    eng2.close()


def test_the_environment_is_ignored(monkeypatch):
    """Round 4's switches were environment variables; a process that still sets them gets the library's defaults."""
    from flobaroid_amd._lib import Engine

    t = load_topo("walkman_apriori")
    for name in ("FBR_NO_LINK_MERGE", "FBR_REDUCE_ALWAYS", "FBR_GRAM_SHAPE", "FBR_CHUNK_SAMPLES", "FBR_TSQR_NO_GROUPS"):
        monkeypatch.setenv(name, "1")
    eng = Engine(t, floating=True)
    assert eng.link_merge_info()["reduced_cols"] == 213 and eng.link_merge_info(500)["reduced_cols"] == 480
    assert eng.gram_program_info(1) == {"tiles": 15, "pairs": 120, "mfma_per_sample": 272, "parts": 1}
    eng.close()


def test_program_options_rebuild_the_tile_program_between_calls():
    """gram_shape / gram_rhs_tile / gram_orient change the cached tile programs: the next call runs the new one and gives the same Gram."""
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    t = load_topo("walkman_left_arm")
    rng = np.random.default_rng(5)
    S = 400
    st = random_states(t, S, rng, True)
    eng = Engine(t, floating=True)
    om = OracleModel(t, floating=True)
    rhs = rng.standard_normal((S * eng.rows, 1))
    A = np.hstack([om.regressor(st), rhs])
    Go = A.T @ A
    seen = []
    for key, val in (("gram_shape", 0), ("gram_shape", 1), ("gram_shape", 2), ("gram_rhs_tile", 1), ("gram_orient", 0), ("reduce_min_work", 0), ("regroup", 0)):
        eng.set_option(key, val)
        G = eng.gram(st, rhs=rhs)
        assert _rel(G, Go) <= 1e-11, (key, val)
        seen.append(eng.gram_program_info(1, S))
    assert seen[3]["mfma_per_sample"] > seen[2]["mfma_per_sample"]  # the dense rhs tile costs MFMAs the packer's moments do not
    eng.close()


def test_an_engine_destroyed_after_a_reduced_pass_leaves_the_next_one_working():
    """The reduced models run on their parent's stream; they are released before the parent destroys that stream (round-4 advisor
    finding: a use-after-destroy whose sticky HIP error a later handle of the same thread would have reported as its own)."""
    from flobaroid_amd._lib import Engine

    t = load_topo("walkman_apriori")
    rng = np.random.default_rng(9)
    st = random_states(t, 300, rng, True)
    ref = None
    for i in range(3):
        eng = Engine(t, floating=True, options={"reduce_min_work": 0, "tsqr_group_min_samples": 1})
        G = eng.gram(st)
        R = eng.tsqr(st)
        assert _rel(R.T @ R, G) <= 1e-11
        ref = G if ref is None else ref
        assert np.array_equal(G, ref)
        eng.close()          # destroys the handle with both reduced models used
    eng = Engine(load_topo("kuka_lwr4"))
    assert np.isfinite(eng.gram(random_states(load_topo("kuka_lwr4"), 100, rng, False))).all()
    eng.close()


@pytest.mark.parametrize("name,floating", [("walkman_apriori", True), ("walkman_left_arm", True), ("kuka_lwr4", False)])
@pytest.mark.parametrize("mode", ["direct", "reduced"])
def test_gram_lane_and_force_tiles_are_switches_not_results(name, floating, mode):
    """Round 6's Gram pass over sample-contiguous images (options gram_lane, gram_force_tiles; fbr.h fbr_gram_lane_info): on or off, with
    and without the force tiles, the Gram is the oracle's -- with row weights, without a rhs column, with a sample count that fills no whole
    block, accumulated over two calls, with a base-wrench-only row mask and as a submission; the info call says which pass a batch takes
    and what it executes."""
    import torch

    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    t = load_topo(name)
    om = OracleModel(t, floating=floating)
    rng = np.random.default_rng(61)
    S = 64 * 9 + 37
    st = random_states(t, S, rng, floating, use_limits=True)
    Y = om.regressor(st)
    tau = rng.standard_normal((Y.shape[0], 1))
    w = 0.5 + rng.random(Y.shape[0])
    A = np.hstack([Y, tau])
    base = {"reduce_min_work": 0 if mode == "reduced" else 1e30}
    got = {}
    for key, opts in (("lane", {}), ("no_force_tiles", {"gram_force_tiles": 0}), ("images", {"gram_lane": 0})):
        eng = Engine(t, floating=floating, options=dict(base, **opts))
        info = eng.gram_lane_info(1, S)
        # (all 480 columns of WALK-MAN need a tile program in two parts: the pass over per-sample images whatever the switch says)
        inside = not (name == "walkman_apriori" and mode == "direct")
        assert info["active"] == (key != "images" and inside)
        if key == "lane" and inside:
            assert (info["force_tiles"] >= 1) == floating and info["stages"] >= 1 and info["block_image_bytes"] == 8192 * info["tile_rows"]
            assert info["balanced_pair_levels"] <= info["busiest_wave_pair_levels"] <= info["balanced_pair_levels"] + info["levels"]
            lane_mfma = info["mfma_per_block"]
        if key == "no_force_tiles" and inside:
            assert info["force_tiles"] == 0 and (info["mfma_per_block"] > lane_mfma) == floating
        G = eng.gram(st, rhs=tau)
        assert _rel(G, A.T @ A) <= 1e-12 and np.array_equal(G, G.T) and np.array_equal(G, eng.gram(st, rhs=tau))
        Gw = eng.gram(st, rhs=tau, w=w)
        assert _rel(Gw, (A * w[:, None]).T @ (A * w[:, None])) <= 1e-12
        G0 = eng.gram(st)
        assert _rel(G0, Y.T @ Y) <= 1e-12
        h = 64 * 4 + 5
        first = {k: v[:h] for k, v in st.items()}
        second = {k: v[h:] for k, v in st.items()}
        G2 = eng.gram(second, rhs=tau[h * om.rows:], out=eng.gram(first, rhs=tau[: h * om.rows]), accumulate=True)
        assert _rel(G2, A.T @ A) <= 1e-12
        if floating:   # base-wrench-only identification: every joint row switched off
            wb = np.zeros((S, om.rows))
            wb[:, :6] = 1.0 + rng.random((S, 6))
            wb = wb.reshape(-1)
            Gb = eng.gram(st, rhs=tau, w=wb)
            assert _rel(Gb, (A * wb[:, None]).T @ (A * wb[:, None])) <= 1e-12
        # one Gram per group of samples (fbr_gram_grouped): groups that fill no whole block, with row weights
        ngr = 7
        Sg = S // ngr
        stg = {k: v[: ngr * Sg] for k, v in st.items()}
        wg = w[: ngr * Sg * om.rows]
        Gg = eng.gram_grouped(stg, ngr, w=wg)
        for gi in range(ngr):
            Yg = Y[gi * Sg * om.rows:(gi + 1) * Sg * om.rows] * wg[gi * Sg * om.rows:(gi + 1) * Sg * om.rows, None]
            assert _rel(Gg[gi], Yg.T @ Yg) <= 1e-12
        got[key + "_grouped"] = Gg
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
        out = torch.zeros((om.P + 1, om.P + 1), dtype=torch.float64, device="cuda")
        eng.wait(eng.gram_submit(dev, out, rhs=torch.from_numpy(tau).cuda()))
        assert np.array_equal(out.cpu().numpy(), G)
        got[key] = G
        eng.close()
    assert _rel(got["lane"], got["images"]) <= 1e-13 and _rel(got["no_force_tiles"], got["images"]) <= 1e-13
    assert _rel(got["lane_grouped"], got["images_grouped"]) <= 1e-13


@pytest.mark.parametrize("name", ["walkman_apriori", "walkman_left_arm"])
def test_tsqr_force_group_is_a_switch_not_a_result(name):
    """The force rows of a floating base as a row group of their own (option tsqr_force_group): the same sign-normalised factor with and
    without it, fewer MFMAs with it -- also for a column subset and with row weights."""
    import scipy.linalg as sla

    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    t = load_topo(name)
    om = OracleModel(t, floating=True)
    rng = np.random.default_rng(62)
    S = 1200
    st = random_states(t, S, rng, True, use_limits=True)
    Y = om.regressor(st)
    tau = rng.standard_normal((Y.shape[0], 1))
    w = 0.5 + rng.random(Y.shape[0])
    Go = np.hstack([Y, tau]).T @ np.hstack([Y, tau])
    rank = int(np.linalg.matrix_rank(Go[: om.P, : om.P]))
    cols = np.sort(sla.qr(Go[: om.P, : om.P], pivoting=True, mode="r")[1][: rank - 3]).astype(np.int32)
    norm = lambda R: R * np.where(np.diag(R) < 0, -1.0, 1.0)[:, None]
    res = {}
    for fg in (1, 0):
        eng = Engine(t, floating=True, options={"tsqr_group_min_samples": 1, "tsqr_force_group": fg})
        R = eng.tsqr(st, rhs=tau)
        assert np.all(np.tril(R, -1) == 0) and _rel(R.T @ R, Go) <= 1e-11
        Aw = np.hstack([Y[:, cols], tau]) * w[:, None]
        Rc = eng.tsqr(st, rhs=tau, w=w, cols=cols)
        assert _rel(Rc.T @ Rc, Aw.T @ Aw) <= 1e-11
        res[fg] = (norm(Rc), eng.tsqr_work_info(1000000, k=1)["flop"])
        eng.close()
    assert _rel(res[1][0], res[0][0]) <= 1e-9 and res[1][1] < res[0][1]
