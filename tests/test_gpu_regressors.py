"""The reference's own regressor test (tests/test_regressors.py:15-126) run through the C-ABI on the GPU.

Same model (threeLinks, floating base), same number of states (100), same state distributions (lines 50-66), same contact
(frame `contact_ft`, wrench [0, 0, 10, 0, 0, 0], line 109) and the same pass criterion (line 126:
||(Y x + J^T f) - (ID + J^T f)||_F <= 0.01 over all samples).  Where the reference calls iDynTree once per sample
(`inverseDynamicsInertialParametersRegressor`, `inverseDynamics`, `getFrameFreeFloatingJacobian`), one batched call each of
`fbr_regressor`, `fbr_inverse_dynamics`, `fbr_contact_torques` runs here.  The kernels are then held to their own bar (1e-10),
and the same check runs on the other bundled robots.
"""
import numpy as np
import numpy.linalg as la
import pytest

from common import load_topo, random_states

pytestmark = pytest.mark.gpu


def _run(robot, contact_frame, num_samples, seed):
    from flobaroid_amd._lib import Engine

    topo = load_topo(robot)
    eng = Engine(topo, floating=True)
    n_dofs = topo.num_dofs
    num_model_params = topo.num_links * 10
    dim = n_dofs + 6
    assert (eng.rows, eng.cols) == (dim, num_model_params)
    xStdModel = topo.x_std()

    rng = np.random.default_rng(seed)
    st = random_states(topo, num_samples, rng, floating=1)  # the distributions of test_regressors.py:50-66
    regressor_stack = eng.regressor(st)
    assert regressor_stack.shape == (dim * num_samples, num_model_params)
    idyn_torques = eng.inverse_dynamics(st, xStdModel).reshape(-1)

    contact = np.tile(np.array([0, 0, 10, 0, 0, 0], dtype=float), (num_samples, 1))
    contactForceSum = eng.contact_torques(st, contact_frame, contact).reshape(-1)
    assert la.norm(contactForceSum) > 0.0

    regressor_torques = np.dot(regressor_stack, xStdModel) + contactForceSum
    idyn_torques = idyn_torques + contactForceSum
    error = np.reshape(regressor_torques - idyn_torques, (num_samples, dim))
    return la.norm(error), la.norm(idyn_torques)


def test_regressors():
    error_norm, scale = _run("threeLinks", "contact_ft", 100, 0)
    assert error_norm <= 0.01               # the reference's criterion
    assert error_norm <= 1e-10 * scale       # ours


@pytest.mark.parametrize("robot,frame", [("kuka_lwr4", None), ("walkman_left_arm", None), ("walkman_apriori", "l_leg_ft")])
def test_regressors_other_robots(robot, frame):
    topo = load_topo(robot)
    if frame is None or frame not in topo.frames:
        frame = topo.link_names[-1]
    error_norm, scale = _run(robot, frame, 100, 1)
    assert error_norm <= 0.01
    assert error_norm <= 1e-10 * scale
