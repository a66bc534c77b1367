"""N1 end to end on the device: candidate trajectories from their Fourier coefficients (fbr_fourier_states), pinned on outputs of
the reference's own generator (tests/golden/ref_trajectories.npz, tools/make_fixtures.py reference_trajectories: excitation/
trajectoryGenerator.py run unmodified -- pure NumPy, no iDynTree on this path), then straight into the grouped Gram / D-optimality."""
import os

import numpy as np
import pytest

from common import GOLDEN, load_topo

pytestmark = pytest.mark.gpu


def _fixture():
    return dict(np.load(os.path.join(GOLDEN, "ref_trajectories.npz"), allow_pickle=False))


def _candidates(fx, c, mode):
    from flobaroid_amd import excitation as exc

    n = int(fx["num_dofs"])
    nf = fx[f"c{c}_nf"]
    a = [fx[f"c{c}_a"][j, : nf[j]] for j in range(n)]
    b = [fx[f"c{c}_b"][j, : nf[j]] for j in range(n)]
    q0 = fx[f"c{c}_q0"]
    lim = [tuple(l) for l in fx["joint_limits"]] if mode == "bounded" else None
    use_deg = mode == "classic_deg"
    return exc.fourier_coefficients(a, b, np.rad2deg(q0) if use_deg else q0, nf, float(fx[f"c{c}_wf"]), joint_limits=lim, use_deg=use_deg)


@pytest.mark.parametrize("mode", ["classic", "classic_deg", "bounded"])
def test_fourier_states_match_the_reference_generator(mode):
    from flobaroid_amd import excitation as exc
    from flobaroid_amd._lib import Engine

    fx = _fixture()
    eng = Engine(load_topo("kuka_lwr4"))   # 7 DOF: only the joint count matters to the generator
    freq = float(fx["freq"])
    for c in range(int(fx["num_cases"])):
        cand = _candidates(fx, c, mode)
        tag = f"c{c}_{mode}_"
        T = fx[tag + "positions"].shape[0]
        if mode == "bounded":  # the generator objects' own centre / range (BoundedOscillationGenerator.__init__)
            assert np.allclose(cand["q_offset"], fx[tag + "q_center"], rtol=0, atol=1e-15) and np.allclose(cand["q_range"], fx[tag + "q_range"], rtol=0, atol=1e-15)
        for device in (True, False):
            st = exc.candidate_states(eng, [cand], T, freq, device=device, use_deg_vectorised_quirk=(mode == "classic_deg"))
            for key, ref in (("q", "positions"), ("dq", "velocities"), ("ddq", "accelerations")):
                got = st[key].cpu().numpy() if device else st[key]
                want = fx[tag + ref]
                assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0), (c, mode, key)
        if mode == "classic_deg":  # without the quirk: the per-sample path of the reference (getAngle / getVelocity / getAcceleration)
            st = exc.candidate_states(eng, [cand], T, freq, device=False)
            assert np.abs(st["q"] - fx[tag + "persample_positions"]).max() <= 1e-12 * np.abs(fx[tag + "persample_positions"]).max()
    eng.close()


def test_candidates_from_coefficients_to_dopt_without_leaving_the_device():
    """Four candidates in one batch: states generated on the device == the reference generator's, their grouped Gram == the Gram of the
    same samples handed over from the host, and the D-optimality of the reference's expression from both."""
    import scipy.linalg as sla

    from flobaroid_amd import excitation as exc
    from flobaroid_amd._lib import Engine

    fx = _fixture()
    t = load_topo("kuka_lwr4")
    eng = Engine(t)
    freq = float(fx["freq"])
    C = int(fx["num_cases"])
    T = min(fx[f"c{c}_classic_positions"].shape[0] for c in range(C))
    cands = [_candidates(fx, c, "classic") for c in range(C)]
    st = exc.candidate_states(eng, cands, T, freq)
    host = {k: np.concatenate([fx[f"c{c}_classic_{r}"][:T] for c in range(C)]) for k, r in (("q", "positions"), ("dq", "velocities"), ("ddq", "accelerations"))}
    for k in host:
        assert np.abs(st[k].cpu().numpy() - host[k]).max() <= 1e-12 * max(np.abs(host[k]).max(), 1.0)
    Gd = eng.gram_grouped(st, C).cpu().numpy()
    Gh = eng.gram_grouped(host, C)
    assert np.linalg.norm(Gd - Gh) <= 1e-11 * np.linalg.norm(Gh)
    rng = np.random.default_rng(1)
    from common import random_states

    ic = np.sort(sla.qr(eng.gram(random_states(t, 2000, rng, False, use_limits=True)), pivoting=True, mode="r")[1][:43])
    d_dev = exc.candidate_dopt_from_coefficients(eng, cands, T, freq, ic)
    d_host = exc.candidate_dopt(eng, host, C, ic)
    assert np.allclose(d_dev, d_host, rtol=1e-9, atol=1e-9) and np.all(np.isfinite(d_dev))
    eng.close()
