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
