"""CPU emulation of the HIP kernels' data flow (same headers as the kernels) against the oracle (no GPU)."""
import numpy as np
import pytest

from common import CONFIGS, cfg_id, load_topo, random_states, random_topology
from emul_lib import Emul
from oracle.oracle import OracleModel


def _setup(cfg, S, seed):
    name, fl, fr, sym, grav, strb = cfg
    t = load_topo(name)
    om = OracleModel(t, floating=fl, fric=fr, fric_sym=sym, grav_only=grav, stribeck=strb)
    em = Emul(t, floating=fl, fric=fr, fric_sym=sym, grav_only=grav, stribeck=strb)
    rng = np.random.default_rng(seed)
    st = random_states(t, S, rng, fl)
    sign = np.tanh(st["dq"] / 0.02)
    return t, om, em, st, sign, rng


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
def test_device_math_matches_oracle(cfg):
    t, om, em, st, sign, rng = _setup(cfg, 25, 1)
    assert (em.rows, em.cols) == (om.rows, om.P)
    Y = om.regressor(st, sign)
    Ye = em.regressor(st, sign)
    assert np.abs(Y - Ye).max() <= 1e-13 * np.abs(Y).max()
    x = np.concatenate([t.x_std(), rng.random(om.P + 4 * t.num_dofs)])
    tau = om.inverse_dynamics(st, x, sign, 0.9 * st["dq"])
    te = em.inverse_dynamics(st, x, sign, 0.9 * st["dq"])
    assert np.abs(tau - te).max() <= 1e-13 * np.abs(tau).max()
    xi = rng.standard_normal(om.P)
    tp = em.inverse_dynamics(st, xi, sign, mode=1)
    assert np.abs(tp.reshape(-1) - Y @ xi).max() <= 1e-12 * np.abs(Y @ xi).max()


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
@pytest.mark.parametrize("k", [0, 3])
@pytest.mark.parametrize("shape", [0, 1, 2], ids=["auto", "one_per_cu", "two_per_cu"])
def test_tile_program_reproduces_gram(cfg, k, shape, request):
    """Both compiled kernel shapes (FbrGramConfig) and the product's choice between them."""
    import emul_lib
    emul_lib.lib().emul_set_gram_shape(shape)
    request.addfinalizer(lambda: emul_lib.lib().emul_set_gram_shape(0))
    t, om, em, st, sign, rng = _setup(cfg, 12, 2)
    Y = om.regressor(st, sign)
    rhs = rng.standard_normal((Y.shape[0], k)) if k else None
    w = rng.random(Y.shape[0]) + 0.5
    A = (Y if rhs is None else np.hstack([Y, rhs])) * w[:, None]
    G = em.gram(st, rhs, sign, w)
    assert np.linalg.norm(G - A.T @ A) <= 1e-13 * np.linalg.norm(A.T @ A)
    info = em.program_info(k)
    dense = ((om.P + k + 15) // 16) * ((om.P + k + 15) // 16 + 1) // 2 * ((om.rows + 3) // 4)
    # the chain packing must never cost much more than the dense tiling (friction and rhs columns have tiles of their own:
    # a few MFMAs of absolute slack for the two-joint toy model)
    assert info["mfma"] <= dense * 1.35 + 6
    assert 2 * info["part_image_max"] * 8 <= 150 * 1024  # two DMA buffers of the largest part fit the LDS


def test_walkman_program_is_sparse():
    t = load_topo("walkman_apriori")
    em = Emul(t, floating=True)
    info = em.program_info(1)
    assert info["mfma"] < 0.4 * 4185  # vs 31*32/2 tiles x 9 k-steps of the dense tiling
    assert info["T"] * 8 * 18 >= info["npairs"]  # one-per-CU shape: 18 accumulators per wave
    assert info["T"] <= 8


@pytest.mark.parametrize("fric,sym", [(False, True), (True, True), (True, False)])
def test_shape_chooser_rule(fric, sym):
    """fbr_gram_build_best: the two-workgroups-per-CU shape only when the model fits a single part."""
    import emul_lib
    t = load_topo("walkman_apriori")
    em = Emul(t, floating=True, fric=fric, fric_sym=sym)
    try:
        auto = em.program_info(1)
        emul_lib.lib().emul_set_gram_shape(2)
        two = em.program_info(1)
        emul_lib.lib().emul_set_gram_shape(1)
        one = em.program_info(1)
    finally:
        emul_lib.lib().emul_set_gram_shape(0)
    assert one["T"] < two["T"] and one["mfma"] == two["mfma"]
    assert auto["T"] == (two["T"] if two["T"] <= 1 else one["T"])
    if fric:  # friction columns cost few MFMAs: one packed row each, no products with links of other branches
        assert auto["mfma"] < 1.1 * 1529


@pytest.mark.parametrize("seed,L,branch,floating,fric,sym", [
    (1, 12, 0.5, 1, 1, 1), (2, 25, 0.3, 1, 0, 1), (3, 25, 0.8, 0, 1, 0), (4, 40, 0.15, 1, 1, 0), (5, 40, 0.6, 0, 0, 1),
    (6, 55, 0.0, 1, 1, 1), (7, 60, 0.4, 0, 1, 0),
])
@pytest.mark.parametrize("shape", [1, 2], ids=["one_per_cu", "two_per_cu"])
def test_random_trees(seed, L, branch, floating, fric, sym, shape, request):
    """Tile program on random trees (deep chains, bushy trees, fixed joints, friction): emulated Gram == oracle Gram."""
    import emul_lib
    rng = np.random.default_rng(seed)
    t = random_topology(rng, L, p_fixed=0.0 if seed == 6 else 0.25, branchiness=branch)
    if t.num_dofs + (6 if floating else 0) > 60:
        pytest.skip("more than 60 regressor rows")
    emul_lib.lib().emul_set_gram_shape(shape)
    request.addfinalizer(lambda: emul_lib.lib().emul_set_gram_shape(0))
    om = OracleModel(t, floating=bool(floating), fric=bool(fric), fric_sym=bool(sym))
    em = Emul(t, floating=bool(floating), fric=bool(fric), fric_sym=bool(sym))
    S = 6
    st = random_states(t, S, rng, floating)
    sign = np.tanh(st["dq"] / 0.02)
    Y = om.regressor(st, sign)
    rhs = rng.standard_normal((Y.shape[0], 2))
    A = np.hstack([Y, rhs])
    G = em.gram(st, rhs, sign, None)
    assert np.linalg.norm(G - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
@pytest.mark.parametrize("k", [1, 2])
def test_rhs_moments_from_the_packer_reproduce_gram(cfg, k, request):
    """Few rhs columns (the reference's one: tau) get no dense tile: Y^T rhs and rhs^T rhs are accumulated where the image is packed
    (fbr_gram_rhs_moments -- what fbr_gram_accumulate runs for k <= 2 and at most 255 columns), with row weights, odd sample counts
    (the last sample has no partner) and every kernel shape's tile program built WITHOUT rhs tiles."""
    import emul_lib
    emul_lib.lib().emul_set_rhs_moments(1)
    request.addfinalizer(lambda: emul_lib.lib().emul_set_rhs_moments(0))
    t, om, em, st, sign, rng = _setup(cfg, 11, 5)
    Y = om.regressor(st, sign)
    rhs = rng.standard_normal((Y.shape[0], k))
    w = rng.random(Y.shape[0]) + 0.5
    A = np.hstack([Y, rhs]) * w[:, None]
    G = em.gram(st, rhs, sign, w)
    assert np.linalg.norm(G - A.T @ A) <= 1e-13 * np.linalg.norm(A.T @ A)
    assert np.array_equal(G, G.T)
    if om.P <= 255:
        emul_lib.lib().emul_set_rhs_moments(0)
        more = em.program_info(k)
        emul_lib.lib().emul_set_rhs_moments(1)
        less = em.program_info(k)
        assert less["mfma"] < more["mfma"]  # the dense tile's MFMAs are gone


@pytest.mark.parametrize("case", ["walkman_apriori", "kuka_lwr4", "random1", "random2", "random3", "prismatic4", "prismatic5", "prismatic6"])
@pytest.mark.parametrize("which", [0, 1], ids=["merged", "regrouped"])
def test_column_reductions_expand_to_the_gram_of_all_columns(case, which, request):
    """The library's column reductions without a GPU: the reduced robot and the expansion matrix E come from the very code
    fbr_model_create runs (csrc/fbr_reduce.h); the reduced model's tile program -- column masks, tau's products from the packer, pairs
    turned for full row segments -- is emulated lane by lane; and E^T G_red E must be the oracle's Gram of ALL columns."""
    import emul_lib
    from oracle.oracle import OracleModel
    from common import random_topology

    rng = np.random.default_rng(77)
    if case.startswith("random") or case.startswith("prismatic"):
        seed = int(case[-1])
        rng = np.random.default_rng(300 + seed)
        # (prismatic*: half of the movable joints slide -- their links keep m and h, the six inertia entries ride on the parent)
        t = random_topology(rng, 8 + 6 * (seed % 4), p_fixed=0.45, branchiness=0.5, p_prismatic=0.5 if case.startswith("prismatic") else 0.0)
        floating, fric = seed != 2, seed in (3, 5)
    else:
        t = load_topo(case)
        floating, fric = case == "walkman_apriori", case == "kuka_lwr4"
    if t.num_dofs == 0:
        pytest.skip("no joints")
    om = OracleModel(t, floating=floating, fric=fric, fric_sym=True)
    em = emul_lib.Emul(t, floating=floating, fric=fric, fric_sym=True)
    red = em.reduction(which)
    nfixed = sum(1 for l in range(1, t.num_links) if t.dof_index[l] < 0)
    if red is None:
        assert which == 0 and nfixed == 0
        return
    rem, E = red
    nf = om.P - 10 * t.num_links  # friction columns
    assert rem.num_links == t.num_links - nfixed
    npris = sum(1 for j in t.joint_type if j == 2)
    assert rem.cols == (10 + 7 * (rem.num_links - 1 - npris) + 4 * npris if which else 10 * rem.num_links) + nf
    if which and floating:  # the regrouped columns are a BASIS: as many as the regressor's numerical rank (with a floating base)
        Yb = om.regressor(random_states(t, 200, np.random.default_rng(1), floating), None if not fric else np.zeros((200, t.num_dofs)))
        assert np.linalg.matrix_rank(Yb[:, : 10 * t.num_links]) == rem.cols - nf
    S = 7  # (odd: the last sample has no partner for its base rows)
    st = random_states(t, S, rng, floating)
    sign = np.tanh(st["dq"] / 0.02)
    Y = om.regressor(st, sign)
    k = 1
    rhs = rng.standard_normal((Y.shape[0], k))
    w = rng.random(Y.shape[0]) + 0.5
    A = np.hstack([Y, rhs]) * w[:, None]
    # the regressor of the reduced robot spans the full one: Y = Y_red E, column by column
    Yr = rem.regressor(st, sign)
    assert np.abs(Yr @ E - Y).max() <= 1e-12 * max(np.abs(Y).max(), 1.0)
    emul_lib.lib().emul_set_rhs_moments(1)
    request.addfinalizer(lambda: emul_lib.lib().emul_set_rhs_moments(0))
    Gr = rem.gram(st, rhs, sign, w)
    Ea = np.zeros((rem.cols + k, om.P + k))
    Ea[: rem.cols, : om.P] = E
    Ea[rem.cols:, om.P:] = np.eye(k)
    G = Ea.T @ Gr @ Ea
    assert np.linalg.norm(G - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)


def test_rhs_moments_of_a_tile_without_rows(request):
    """A fixed base whose first 16 columns (the base link and a link welded to it) have no regressor rows at all: their tile occupies no
    image space and shares its offset with the next tile.  The packer's rhs moments are attributed through the item's own column (not
    through the image offset): every column has exactly one reduction workgroup on the device -- round 5 found two of them racing on one
    G entry for such robots (a lost update the sequential emulation cannot see; it checks the ownership instead)."""
    import emul_lib

    rng = np.random.default_rng(22)
    t = random_topology(rng, 16, p_fixed=0.25, branchiness=0.4)
    assert t.dof_index[1] < 0 and t.parent[1] == 0   # link 1 is welded to the fixed base: columns 0..19 never move
    om = OracleModel(t, floating=False)
    em = emul_lib.Emul(t, floating=False)
    S = 9
    st = random_states(t, S, rng, False)
    Y = om.regressor(st)
    assert not Y[:, :16].any()
    rhs = rng.standard_normal((Y.shape[0], 1))
    A = np.hstack([Y, rhs])
    emul_lib.lib().emul_set_rhs_moments(1)
    request.addfinalizer(lambda: emul_lib.lib().emul_set_rhs_moments(0))
    for shape in (0, 1, 2):
        emul_lib.lib().emul_set_gram_shape(shape)
        request.addfinalizer(lambda: emul_lib.lib().emul_set_gram_shape(0))
        G = em.gram(st, rhs)
        assert np.linalg.norm(G - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A), shape


@pytest.mark.parametrize("cfg", CONFIGS, ids=cfg_id)
def test_fused_kinematics_and_torques_program(cfg):
    """csrc/fbr_kinid.h on the CPU: the device kernel's own step program (slots of the branch-point records, levels of the joint stack,
    flush lists) and its own lane body against the oracle -- every torque row written exactly once, both parameter layouts."""
    t, om, em, st, sign, rng = _setup(cfg, 25, 3)
    x = np.concatenate([t.x_std(), rng.random(om.P + 4 * t.num_dofs)])
    tau = om.inverse_dynamics(st, x, sign, 0.9 * st["dq"])
    tf, (nsteps, maxlvl, nslots) = em.fused_inverse_dynamics(st, x, sign, 0.9 * st["dq"])
    assert nsteps == t.num_links and maxlvl >= 1
    assert np.abs(tau - tf).max() <= 1e-13 * np.abs(tau).max()
    Y = om.regressor(st, sign)
    xi = rng.standard_normal(om.P)
    tp, _ = em.fused_inverse_dynamics(st, xi, sign, mode=1)
    assert np.abs(tp.reshape(-1) - Y @ xi).max() <= 1e-12 * np.abs(Y @ xi).max()


def test_fused_program_of_walkman_needs_two_slots():
    """The per-wave scratch of the fused kernel: WALK-MAN's branch points (waist: legs + torso, torso: arms + neck) nest two deep."""
    t = load_topo("walkman_apriori")
    em = Emul(t, floating=True)
    st = random_states(t, 2, np.random.default_rng(0), True)
    _, (nsteps, maxlvl, nslots) = em.fused_inverse_dynamics(st, t.x_std())
    assert (nsteps, maxlvl) == (48, 10) and nslots <= 4
    red, _E = em.reduction(0)   # the link-merged robot fbr_predict runs on
    _, (ns2, ml2, sl2) = red.fused_inverse_dynamics(st, np.zeros(10 * red.num_links))
    assert (ns2, ml2) == (30, 10) and sl2 <= 3


@pytest.mark.parametrize("seed", range(12))
def test_fused_program_on_random_trees(seed):
    """Random trees with fixed / revolute / prismatic joints, bushy and chain-like, fixed and floating base, friction layouts."""
    rng = np.random.default_rng(500 + seed)
    L = int(rng.integers(2, 26))
    t = random_topology(rng, L, p_fixed=float(rng.choice([0.0, 0.2, 0.5])), branchiness=float(rng.choice([0.0, 0.3, 0.7, 1.0])),
                        p_prismatic=float(rng.choice([0.0, 0.3])))
    fl, fr, sym = bool(seed & 1), bool(seed & 2), bool(seed & 4)
    strb = 0.1 if (fr and seed % 3 == 0) else 0.0
    om = OracleModel(t, floating=fl, fric=fr, fric_sym=sym, stribeck=strb)
    em = Emul(t, floating=fl, fric=fr, fric_sym=sym, stribeck=strb)
    st = random_states(t, 7, rng, fl)
    sign = np.tanh(st["dq"] / 0.02) if t.num_dofs else np.zeros((7, 0))
    x = np.concatenate([t.x_std(), rng.random(om.P + 4 * t.num_dofs)])
    tau = om.inverse_dynamics(st, x, sign, 0.9 * st["dq"])
    tf, info = em.fused_inverse_dynamics(st, x, sign, 0.9 * st["dq"])
    assert np.abs(tau - tf).max() <= 1e-12 * max(np.abs(tau).max(), 1e-300), info
    Y = om.regressor(st, sign)
    xi = rng.standard_normal(om.P)
    tp, _ = em.fused_inverse_dynamics(st, xi, sign, mode=1)
    assert np.abs(tp.reshape(-1) - Y @ xi).max() <= 1e-11 * max(np.abs(Y @ xi).max(), 1e-300)


@pytest.mark.parametrize("seed", range(6))
def test_fused_contact_and_fd_programs_on_random_trees(seed):
    """The other two users of the fused lane body (csrc/fbr_kinid.h): J^T w of a contact wrench (mode 2: only the frame's link carries a
    wrench) and the finite-difference scores (one lane per perturbed evaluation, weights against the unit wrenches)."""
    rng = np.random.default_rng(900 + seed)
    L = int(rng.integers(3, 20))
    t = random_topology(rng, L, p_fixed=0.2, branchiness=float(rng.choice([0.2, 0.6, 1.0])), p_prismatic=0.2 * (seed & 1))
    fl, fr = bool(seed & 1), bool(seed & 2)
    om = OracleModel(t, floating=fl, fric=fr)
    em = Emul(t, floating=fl, fric=fr)
    S = 5
    st = random_states(t, S, rng, fl)
    sign = np.tanh(st["dq"] / 0.02)
    # contact: a frame on a random link
    flink = int(rng.integers(0, L))
    fp = rng.standard_normal(3) * 0.2
    t.frames["probe"] = {"link": flink, "R": np.eye(3), "p": fp}
    w = rng.standard_normal((S, 6))
    ref = om.contact_torques(st, "probe", w)
    got, _ = em.fused_inverse_dynamics(st, w, sign, mode=2, flink=flink, fp=fp)
    assert np.abs(got - ref.reshape(S, -1)).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)
    # finite-difference scores
    n, eps = t.num_dofs, 1e-6
    W = rng.standard_normal((S * om.rows, om.P))
    sc = em.fused_fd_scores(st, W, eps, sign)
    Wb = W.reshape(S, om.rows, om.P)
    refs = np.empty_like(sc)
    refs[:, 0] = np.einsum("src,src->s", Wb, om.regressor(st, sign).reshape(S, om.rows, om.P))
    for kind, key in enumerate(("q", "dq", "ddq")):
        for d in range(n):
            sp = {k: v.copy() for k, v in st.items()}
            sp[key][:, d] += eps
            refs[:, 1 + kind * n + d] = np.einsum("src,src->s", Wb, om.regressor(sp, sign).reshape(S, om.rows, om.P))
    assert np.abs(sc - refs).max() <= 1e-11 * np.abs(refs).max()


@pytest.mark.parametrize("seed", range(8))
def test_tree_parts_of_the_lane_writer(seed):
    """fbr_kinid_build_parts (csrc/fbr_kinid.h): the tree cut into programs for the waves of one workgroup -- every part closed under
    parents, every link owned once; the parts' torques (each link's wrench added by its owner) sum to the robot's."""
    rng = np.random.default_rng(1300 + seed)
    if seed == 0:
        t, fl = load_topo("walkman_apriori"), True
    else:
        t = random_topology(rng, int(rng.integers(2, 25)), p_fixed=0.2, branchiness=float(rng.choice([0.0, 0.4, 1.0])), p_prismatic=0.2)   # (joint paths of at most 24: FBR_KINID_MAXD)
        fl = bool(seed & 1)
    om = OracleModel(t, floating=fl)
    em = Emul(t, floating=fl)
    st = random_states(t, 5, rng, fl)
    x = t.x_std()
    tau = om.inverse_dynamics(st, x)
    for nparts in (1, 2, 4):
        tp, steps = em.fused_parts_inverse_dynamics(st, x, nparts)
        assert len(steps) == min(nparts, t.num_links)
        assert np.abs(tp - tau).max() <= 1e-12 * max(np.abs(tau).max(), 1e-300), (nparts, steps)
    if seed == 0:   # WALK-MAN: four parts walk 48 links + a short trunk each
        assert sum(steps) <= 48 + 4 * 6, steps


@pytest.mark.parametrize("case", ["walkman_apriori", "walkman_left_arm", "kuka_lwr4", "random1", "random2", "random3", "random4", "prismatic5", "prismatic6"])
@pytest.mark.parametrize("which", [-1, 0, 1], ids=["all_columns", "merged", "regrouped"])
def test_gram_over_sample_contiguous_images(case, which):
    """csrc/fbr_gram64.h without a GPU: the producer's destination words (one per part, link and parameter; level stride; the column
    swizzle inside a 32-sample run; every image position written by exactly one lane; the force rows of the inertia columns left as
    structural zeros), the consumer's stage tables (pieces, slabs, the pairs of a wave below their common depth) and the per-lane rhs
    moments, on the full model and on the reduced robots the library runs for large batches -- against the oracle's [Y | tau] Gram, with
    row weights, a block that is not full (70 samples) and k = 0 / 1."""
    import emul_lib

    rng = np.random.default_rng(91)
    if case.startswith("random") or case.startswith("prismatic"):
        seed = int(case[-1])
        rng = np.random.default_rng(500 + seed)
        t = random_topology(rng, 7 + 5 * (seed % 4), p_fixed=0.35, branchiness=0.5, p_prismatic=0.5 if case.startswith("prismatic") else 0.0)
        floating = seed != 2
    else:
        t = load_topo(case)
        floating = case != "kuka_lwr4"
    if t.num_dofs == 0:
        pytest.skip("no joints")
    om = OracleModel(t, floating=floating)
    em = emul_lib.Emul(t, floating=floating)
    E = np.eye(om.P)
    if which >= 0:
        red = em.reduction(which)
        if red is None:
            pytest.skip("nothing to reduce")
        em, E = red
    S = 70
    st = random_states(t, S, rng, floating)
    Y = om.regressor(st, None)
    tau = rng.standard_normal((Y.shape[0], 1))
    w = rng.random(Y.shape[0]) + 0.5
    got = em.gram64(st, tau, w)
    if got is None:
        # (WALK-MAN's 480 columns, and the 300 of its moving bodies, need a tile program in two parts; the regrouped robot the library
        # runs on large batches is inside the pass)
        assert case.startswith("random") or case.startswith("prismatic") or (case == "walkman_apriori" and which < 1)
        pytest.skip("model outside the sample-contiguous pass")
    Gr, stats = got
    # without the force tiles (option gram_force_tiles = 0): the same Gram, more MFMAs when the base floats
    emul_lib.lib().emul_set_force_tiles(0)
    try:
        Gn, stats_n = em.gram64(st, tau, w)
    finally:
        emul_lib.lib().emul_set_force_tiles(1)
    assert np.linalg.norm(Gn - Gr) <= 1e-13 * np.linalg.norm(Gr) and stats_n["force_tiles"] == 0
    if floating and stats["force_tiles"]:   # (0: the extra tile pairs did not fit the accumulator slots)
        assert stats["mfma_per_block"] < stats_n["mfma_per_block"]
    elif floating:
        assert not case.startswith("walkman")
    else:
        assert stats["force_tiles"] == 0 and stats["mfma_per_block"] == stats_n["mfma_per_block"]
    # sixteen waves of ten accumulators (option gram_lane_waves = 16): the same Gram through the other slot layout of the reduction
    emul_lib.lib().emul_set_wide16(1)
    try:
        Gw16, _ = em.gram64(st, tau, w)
    finally:
        emul_lib.lib().emul_set_wide16(0)
    assert np.linalg.norm(Gw16 - Gr) <= 1e-13 * np.linalg.norm(Gr)
    Ea = np.zeros((em.cols + 1, om.P + 1))
    Ea[: em.cols, : om.P] = E
    Ea[-1, -1] = 1.0
    A = np.hstack([Y, tau]) * w[:, None]
    assert np.linalg.norm(Ea.T @ Gr @ Ea - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)
    assert np.array_equal(Gr, Gr.T)
    G0, stats0 = em.gram64(st)
    assert np.linalg.norm(E.T @ G0 @ E - Y.T @ Y) <= 1e-12 * np.linalg.norm(Y.T @ Y)
    assert stats0["mfma_per_block"] == stats["mfma_per_block"] and 1 <= stats["parts"] <= 4
    assert stats["balanced_pair_levels"] <= stats["busiest_wave_pair_levels"] <= stats["balanced_pair_levels"] + stats["levels"] and 1 <= stats["stages"] <= stats["levels"]


@pytest.mark.parametrize("case,floating,sym", [("kuka_lwr4", False, True), ("kuka_lwr4", False, False), ("walkman_left_arm", True, True),
                                               ("random2", False, True), ("random3", True, False), ("random4", True, True)])
@pytest.mark.parametrize("which", [-1, 1], ids=["all_columns", "regrouped"])
def test_gram_over_sample_contiguous_images_with_friction_columns(case, floating, sym, which):
    """Friction columns in the pass of csrc/fbr_gram64.h (the reference's identifyFrictionSimultaneously option sets: kuka_lwr4.yaml,
    walkman_left_arm.yaml): a friction tile holds the levels of its own columns' joints only, its pairs run where both tiles have rows, the
    part that owns a joint's link writes the joint's friction values (one per column, on the joint's row) and their rhs moments -- against
    the oracle's [Y | tau] Gram, with the sign series, row weights and a block that is not full."""
    import emul_lib

    rng = np.random.default_rng(93)
    if case.startswith("random"):
        seed = int(case[-1])
        rng = np.random.default_rng(700 + seed)
        t = random_topology(rng, 6 + 4 * (seed % 3), p_fixed=0.3, branchiness=0.5)
    else:
        t = load_topo(case)
    if t.num_dofs == 0:
        pytest.skip("no joints")
    om = OracleModel(t, floating=floating, fric=True, fric_sym=sym)
    em = emul_lib.Emul(t, floating=floating, fric=True, fric_sym=sym)
    E = np.eye(om.P)
    if which >= 0:
        red = em.reduction(which)
        if red is None:
            pytest.skip("nothing to reduce")
        em, E = red
    S = 70
    st = random_states(t, S, rng, floating)
    sign = np.tanh(st["dq"] / 0.02)
    Y = om.regressor(st, sign)
    tau = rng.standard_normal((Y.shape[0], 1))
    w = rng.random(Y.shape[0]) + 0.5
    got = em.gram64(st, tau, w, sign)
    if got is None:
        assert case.startswith("random")  # (the shipped robots with friction columns are inside the pass)
        pytest.skip("model outside the sample-contiguous pass")
    Gr, stats = got
    Ea = np.zeros((em.cols + 1, om.P + 1))
    Ea[: em.cols, : om.P] = E
    Ea[-1, -1] = 1.0
    A = np.hstack([Y, tau]) * w[:, None]
    assert np.linalg.norm(Ea.T @ Gr @ Ea - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)
    assert np.array_equal(Gr, Gr.T)
    G0, _ = em.gram64(st, sign=sign)
    assert np.linalg.norm(E.T @ G0 @ E - Y.T @ Y) <= 1e-12 * np.linalg.norm(Y.T @ Y)
