"""The floating-base estimator chain the way the reference's own ``Identification`` drives it, on the HIP engine.

tests/golden/ref_identification_fb.npz holds what identifier.py (unchanged, run in the build container on the work-alike Model / Data with
the CPU stand-in engine: tests/test_dropin_identifier.py, tools/make_fixtures.py) computes for
  * threeLinks, floating base, configs/threeLinks.yaml (data-driven pivoted QR, OLS, 2 000 samples), and
  * WALK-MAN with the configs/walkman_full.yaml option set: useBaseWrenchForBaseParams (identifier.py:888-892, :617-681),
    useTrajectoryWeighting over two measurement files of different noise, contact wrenches on both foot FT frames, post-identification
    friction (:979-1099).
Here the same files go through ``Model`` / ``Data`` + ``estimation.*`` on the GPU (tests/fb_chain.py: weighted-row TSQR, streaming
prediction): identical base-parameter index sets, xBase / xStd / tauEstimated / base_error / friction within 1e-6."""
import pytest

import fb_chain

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reduction_mode")]  # three modes of the column reductions: conftest.py


@pytest.mark.parametrize("name", ["threelinks", "walkman"])
def test_floating_base_chain_matches_the_references_identification(name, tmp_path):
    from flobaroid_amd.data import Data
    from flobaroid_amd.model import Model

    z = fb_chain.load_fixture()
    opt, files, topo, tpath = fb_chain.write_inputs(z, name, tmp_path)
    out = fb_chain.run_chain(opt, files, topo, tpath, Model, Data)
    fb_chain.compare(out, z, name, 1e-6)
