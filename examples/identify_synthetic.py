#!/usr/bin/env python3
"""End-to-end identification on synthetic data with the flobaroid_amd drop-ins (the flow of the reference's
examples/identify_*.sh: Data -> Model.computeRegressors -> base-parameter OLS -> standard parameters -> URDF copy).

    python examples/identify_synthetic.py [--robot kuka_lwr4|walkman_left_arm|walkman_apriori] [--samples 20000] [--urdf in.urdf --out out.urdf]

The measurements are simulated with the GPU inverse dynamics of a perturbed ("real") parameter vector plus noise; the
identification then runs exactly as Identification.estimateParameters would on the reference's Model/Data attributes,
except that the tall matrices are never formed: the estimators work from the fused Gram / TSQR factors."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flobaroid_amd import estimation as est  # noqa: E402
from flobaroid_amd.data import Data  # noqa: E402
from flobaroid_amd.model import Model  # noqa: E402
from flobaroid_amd.topology import replace_params_in_urdf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robot", default="kuka_lwr4")
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--noise", type=float, default=0.05)
    ap.add_argument("--urdf", help="URDF to copy with the identified parameters written in (optional)")
    ap.add_argument("--out", default="identified.urdf")
    args = ap.parse_args()

    floating = args.robot.startswith("walkman")
    opt = dict(floatingBase=int(floating), identifyFrictionSimultaneously=0, identifySymmetricVelFriction=1, identifyGravityParamsOnly=0,
               simulateTorques=0, useAPriori=1, useStructuralRegressor=1, skipSamples=0, startOffset=0, verbose=0, showTiming=1,
               filterRegressor=0, estimateWith="std", randomSamples=5000, minTol=1e-4, selectBlocksFromMeasurements=0,
               materializeLimitBytes=0)  # never materialise YStd: everything below works from the small reductions
    path = os.path.join(ROOT, "flobaroid_amd", "robots", args.robot + ".topology.json")
    t0 = time.perf_counter()
    model = Model(opt, path)
    print(f"{args.robot}: {model.num_links} links, {model.num_dofs} DOF, {model.num_identified_params} std params, "
          f"{model.num_base_params} base params (structural QR on the GPU Gram: {time.perf_counter() - t0:.2f} s)")

    # "real" robot: a-priori parameters perturbed by 10 %
    rng = np.random.default_rng(42)
    topo = model.topology
    S, n = args.samples, model.num_dofs
    lo = np.array([topo.limits[j]["lower"] for j in topo.dof_names])
    hi = np.array([topo.limits[j]["upper"] for j in topo.dof_names])
    vm = np.array([topo.limits[j]["velocity"] for j in topo.dof_names])
    samples = {"positions": lo + (hi - lo) * rng.random((S, n)), "velocities": (rng.random((S, n)) - 0.5) * 2 * vm,
               "accelerations": (rng.random((S, n)) - 0.5) * 2 * np.pi, "times": np.arange(S) / 200.0, "frequency": np.array(200.0)}
    if floating:
        samples.update(base_velocity=np.pi * rng.random((S, 6)), base_acceleration=np.pi * rng.random((S, 6)), base_rpy=0.1 * rng.random((S, 3)))
    x_real = model.xStdModel * (1.0 + 0.1 * rng.standard_normal(model.xStdModel.shape))
    tau = model.simulateDynamicsBatch(samples, np.arange(S), xStdModel=x_real)
    samples["torques"] = tau + rng.normal(0, args.noise, tau.shape)

    data = Data(opt)
    data.init_from_data(samples)
    t0 = time.perf_counter()
    model.computeRegressors(data)
    print(f"computeRegressors on {S} samples ({model.N_OUT} x {model.num_identified_params} block each): {time.perf_counter() - t0:.3f} s")

    # base-parameter OLS + standard parameters, from the augmented TSQR factor of [YStd | tau]
    t0 = time.perf_counter()
    st = model._states_from_samples(data.samples, np.arange(S))
    R_aug = model.engine.tsqr(st, rhs=model.tau.reshape(-1, 1))
    P = model.num_identified_params
    xBase, Rb, sv = est.identify_base_parameters(R_aug, model.independent_cols, P, S * model.N_OUT, add_contacts=False)
    xStd = est.find_std_from_base(model.K, xBase) + model.xStdModel[model.identified_params]
    rho = est.residual_sq_from_R(Rb, model.num_base_params, xBase)
    p_sigma = est.std_dev_for_params(Rb, model.num_base_params, xBase, rho, S * model.N_OUT)
    print(f"TSQR + OLS + std-dev: {time.perf_counter() - t0:.3f} s; cond(YBase) = {sv[0] / sv[-1]:.1f}")
    xBase_real = model.K @ (x_real[model.identified_params] - model.xStdModel[model.identified_params])
    print(f"base-parameter error vs the simulated robot: {np.linalg.norm(xBase - xBase_real) / np.linalg.norm(xBase_real) * 100:.2f} %  "
          f"(median relative std-dev {np.median(p_sigma) * 100:.2f} %)")
    print(f"torque residual: {np.sqrt(rho / (S * model.N_OUT)):.4f} (noise sigma {args.noise})")
    if args.urdf:
        replace_params_in_urdf(args.urdf, args.out, topo, xStd)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
