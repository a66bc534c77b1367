"""Close the one open pin of the oracle: outputs of iDynTree itself.

The per-sample arithmetic of the reference's hot path is iDynTree 15.0.0 (pyproject.toml:12,42), which is not installable in the build
container, so ``oracle/fbr_oracle.c`` is pinned on known answers and properties only.  On ANY machine that has the reference's
environment (``import idyntree`` works -- e.g. ``uv sync`` in a FloBaRoID checkout) this script runs the three iDynTree entry points
of the path exactly as the reference calls them

    KinDynComputations.inverseDynamicsInertialParametersRegressor   identification/model.py:425-446
    KinDynComputations.inverseDynamics                              identification/model.py:268-296
    KinDynComputations.getFrameFreeFloatingJacobian                 identification/model.py:535-549, tests/test_regressors.py:108-113

on the seeded states of ``tests/common.random_states`` for the four bundled robots (fixed and floating base, one contact frame each) and
writes ``tests/golden/idyntree_<robot>.npz`` -- inputs, outputs, iDynTree's own link / joint serialisation and a-priori parameter
vector.  ``tests/test_oracle.py::test_oracle_against_idyntree_outputs`` (CPU) and ``tests/test_gpu_parity.py::
test_hip_against_idyntree_outputs`` (GPU) load the files when present and skip otherwise; commit the four files and row (c) of SURVEY 8
turns from "parity unpinned" into a pinned oracle.

    python tools/pin_idyntree.py --model-dir /path/to/FloBaRoID/model            # write the fixtures
    python tools/pin_idyntree.py --model-dir ... --check                           # and compare with oracle/ right away

Without iDynTree it explains what is missing and exits with status 3 (so that a CI job can tell "not pinned" from "mismatch" = 1).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ROBOTS = {  # fixture name -> (URDF in the reference's model/ directory, contact frame of the Jacobian check)
    "threeLinks": ("threeLinks.urdf", "contact_ft"),
    "kuka_lwr4": ("kuka_lwr4.urdf", "lwr_7_link"),
    "walkman_left_arm": ("walkman_left_arm.urdf", "l_arm_ft"),
    "walkman_apriori": ("walkman_apriori.urdf", "l_sole"),
}
SAMPLES = 40
SEED = 20260929


def _import_idyntree():
    try:
        import idyntree.bindings as iDynTree  # the module name of iDynTree >= 9 wheels
        return iDynTree
    except ImportError:
        try:
            import iDynTree  # older SWIG installs (the reference's own import: identification/model.py:13)
            return iDynTree
        except ImportError:
            return None


def _vec(iDynTree, values):
    v = iDynTree.VectorDynSize(len(values))
    for i, x in enumerate(values):
        v.setVal(i, float(x))
    return v


def run_robot(iDynTree, urdf: str, frame: str, floating: bool, st: dict) -> dict:
    """The reference's call sequence on every sample of ``st`` (model.py:425-446, 268-296, 535-549)."""
    loader = iDynTree.ModelLoader()
    if not loader.loadModelFromFile(urdf):
        raise RuntimeError(f"iDynTree could not load {urdf}")
    model = loader.model()
    kd = iDynTree.KinDynComputations()
    assert kd.loadRobotModel(model)
    n, L = model.getNrOfDOFs(), model.getNrOfLinks()
    grav = iDynTree.Vector3()
    for i, g in enumerate((0.0, 0.0, -9.81)):
        grav.setVal(i, g)
    x_std = iDynTree.VectorDynSize(10 * L)
    model.getInertialParameters(x_std)
    S = st["q"].shape[0]
    Y = np.zeros((S, 6 + n, 10 * L))
    tau = np.zeros((S, 6 + n))
    J = np.zeros((S, 6, 6 + n))
    have_frame = model.getFrameIndex(frame) >= 0
    for s in range(S):
        q, dq, ddq = _vec(iDynTree, st["q"][s]), _vec(iDynTree, st["dq"][s]), _vec(iDynTree, st["ddq"][s])
        base_acc = iDynTree.Vector6()
        if floating:
            rpy = st["rpy"][s]
            world_T_base = iDynTree.Transform(iDynTree.Rotation.RPY(rpy[0], rpy[1], rpy[2]), iDynTree.Position.Zero()).inverse()
            kd.setRobotState(world_T_base, q, iDynTree.Twist.FromPython([float(v) for v in st["base_vel"][s]]), dq, grav)
            for i in range(6):
                base_acc.setVal(i, float(st["base_acc"][s][i]))
        else:
            kd.setRobotState(q, dq, grav)
        reg = iDynTree.MatrixDynSize()
        assert kd.inverseDynamicsInertialParametersRegressor(base_acc, ddq, reg)
        Y[s] = reg.toNumPy()
        ext = iDynTree.LinkWrenches(model)
        gen = iDynTree.FreeFloatingGeneralizedTorques(model)
        assert kd.inverseDynamics(base_acc, ddq, ext, gen)
        tau[s, :6] = gen.baseWrench().toNumPy()
        tau[s, 6:] = gen.jointTorques().toNumPy()
        if have_frame:
            jac = iDynTree.MatrixDynSize(6, 6 + n)
            assert kd.getFrameFreeFloatingJacobian(frame, jac)
            J[s] = jac.toNumPy()
    return {
        "Y": Y, "tau": tau, "J": J, "have_frame": np.array(have_frame),
        "x_std": x_std.toNumPy(),
        "link_names": np.array([model.getLinkName(i) for i in range(L)]),
        "joint_names": np.array([model.getJointName(i) for i in range(model.getNrOfJoints())]),
        "dof_names": np.array([model.getJointName(i) for i in range(model.getNrOfJoints()) if model.getJoint(i).getNrOfDOFs() > 0]),
    }


def make_states(topo, floating: bool):
    from common import random_states

    rng = np.random.default_rng(SEED)
    return random_states(topo, SAMPLES, rng, floating, use_limits=False)


def compare_with_oracle(name: str, fx: dict, verbose: bool = True) -> float:
    """Worst relative deviation of oracle/ from the iDynTree outputs of one fixture (what the tests assert <= 1e-9)."""
    from common import load_topo
    from oracle.oracle import OracleModel

    topo = load_topo(name)
    worst = 0.0
    # serialisation: iDynTree's link / DOF order must be the one the topology was serialised in
    assert list(fx["link_names"]) == list(topo.link_names), (name, "link order differs from iDynTree's", list(fx["link_names"]), topo.link_names)
    assert list(fx["dof_names"]) == list(topo.dof_names), (name, "DOF order differs from iDynTree's")
    d = float(np.abs(fx["x_std"] - topo.x_std()[: fx["x_std"].size]).max() / max(np.abs(fx["x_std"]).max(), 1e-300))
    worst = max(worst, d)
    for fl in (0, 1):
        st = {k[len(f"fb{fl}_"):]: fx[k] for k in fx if k.startswith(f"fb{fl}_") and k.split("_", 1)[1] in ("q", "dq", "ddq", "base_vel", "base_acc", "rpy")}
        om = OracleModel(topo, floating=bool(fl))
        Y = om.regressor(st).reshape(SAMPLES, om.rows, om.P)
        Yi = fx[f"fb{fl}_Y"][:, (0 if fl else 6):, :]
        tau = om.inverse_dynamics(st, topo.x_std())
        ti = fx[f"fb{fl}_tau"][:, (0 if fl else 6):]
        dY = float(np.abs(Y - Yi).max() / np.abs(Yi).max())
        dt = float(np.abs(tau - ti).max() / np.abs(ti).max())
        worst = max(worst, dY, dt)
        if verbose:
            print(f"  {name} floating={fl}: regressor {dY:.2e}  inverse dynamics {dt:.2e}", end="")
        if fl and bool(fx["fb1_have_frame"]):
            frame = ROBOTS[name][1]
            w = np.random.default_rng(SEED + 1).standard_normal((SAMPLES, 6))
            ct = om.contact_torques(st, frame, w)
            ci = np.einsum("sij,si->sj", fx["fb1_J"], w)
            dc = float(np.abs(ct - ci).max() / np.abs(ci).max())
            worst = max(worst, dc)
            if verbose:
                print(f"  J^T w ({frame}) {dc:.2e}", end="")
        if verbose:
            print()
    return worst


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model-dir", default=os.environ.get("FLOBAROID_MODEL_DIR", "/root/reference/model"), help="the reference's model/ directory (URDFs)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--check", action="store_true", help="compare the written fixtures with oracle/ (needs gcc)")
    args = ap.parse_args()
    iDynTree = _import_idyntree()
    if iDynTree is None:
        print("pin_idyntree: neither `idyntree.bindings` nor `iDynTree` can be imported here.  Run this script in the reference's environment "
              "(FloBaRoID: `uv sync`, idyntree==15.0.0) with --model-dir pointing at its model/ directory; it writes tests/golden/idyntree_*.npz, "
              "which the CPU and GPU test suites pick up on their next run.", file=sys.stderr)
        return 3
    from common import load_topo

    worst = 0.0
    for name, (urdf, frame) in ROBOTS.items():
        path = os.path.join(args.model_dir, urdf)
        if not os.path.exists(path):
            print(f"pin_idyntree: {path} not found (--model-dir)", file=sys.stderr)
            return 2
        topo = load_topo(name)
        fx = {"idyntree_version": np.array(getattr(iDynTree, "__version__", "unknown")), "seed": np.array(SEED), "frame": np.array(frame)}
        for fl in (0, 1):
            st = make_states(topo, bool(fl))
            out = run_robot(iDynTree, path, frame, bool(fl), st)
            for k, v in st.items():
                fx[f"fb{fl}_{k}"] = v
            for k in ("Y", "tau", "J", "have_frame"):
                fx[f"fb{fl}_{k}"] = out[k]
            for k in ("x_std", "link_names", "joint_names", "dof_names"):
                fx[k] = out[k]
        dst = os.path.join(args.out, f"idyntree_{name}.npz")
        np.savez_compressed(dst, **fx)
        print("wrote", dst, f"({os.path.getsize(dst) / 1024:.0f} KiB)")
        if args.check:
            worst = max(worst, compare_with_oracle(name, dict(np.load(dst, allow_pickle=False))))
    if args.check:
        print(f"worst relative deviation oracle vs iDynTree: {worst:.2e} (bar 1e-9)")
        return 0 if worst <= 1e-9 else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
