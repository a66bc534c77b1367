"""The floating-base estimator chain of the reference's ``Identification.estimateParameters`` / ``estimateRegressorTorques``
(identifier.py:857-977, 127-204; base-wrench rows and trajectory weighting :617-681; post-identification friction :979-1099) written
with this repository's ``Model`` / ``Data`` / ``estimation`` on whatever engine the ``Model`` carries -- the HIP engine in the ``-m gpu``
test, the CPU stand-in in the build container's cross-check.  Every reduction goes through the engine (TSQR of the weighted rows,
streaming prediction); nothing here materialises YBase.  Compared against tests/golden/ref_identification_fb.npz, the outputs of the
reference's own class on the same files (tools/make_fixtures.py: reference_identification_fb)."""
import json
import os

import numpy as np

from common import GOLDEN, load_topo


def load_fixture():
    return np.load(os.path.join(GOLDEN, "ref_identification_fb.npz"), allow_pickle=False)


def write_inputs(z, name, tmp_path):
    """measurement files + option set + topology file of a scenario from the fixture"""
    meta = json.loads(str(z[name + "_meta"]))
    files = []
    for i in range(meta["files"]):
        pre = "%s_in%d_" % (name, i)
        arrs = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and not k.startswith(pre + "contacts_")}
        contacts = {k[len(pre + "contacts_"):]: z[k] for k in z.files if k.startswith(pre + "contacts_")}
        if contacts:
            arrs["contacts"] = np.array(contacts)
        path = os.path.join(str(tmp_path), "%s_%d.npz" % (name, i))
        np.savez(path, **arrs)
        files.append(path)
    topo = load_topo(meta["robot"])
    tpath = os.path.join(str(tmp_path), meta["robot"] + ".topology.json")
    topo.save_json(tpath)
    return dict(meta["opt"]), files, topo, tpath


def run_chain(opt, files, topo, tpath, model_cls, data_cls):
    """xBase, xStd, tauEstimated, base_error (+ the post-identified friction) as Identification computes them"""
    import numpy.linalg as la

    from flobaroid_amd import estimation as est, helpers as fh

    np.random.seed(3)
    model = model_cls(opt, tpath)
    data = data_cls(opt)
    data.init_from_files([files])
    model.computeRegressors(data)
    eng, st = model.engine, model._states
    S, rows, P = data.num_used_samples, model.N_OUT, model.num_identified_params
    ic = np.asarray(model.independent_cols)
    skip = int(opt.get("skipSamples", 0))
    target = model.tau if opt["useAPriori"] else model.torques_stack
    if opt["floatingBase"] and opt.get("useBaseWrenchForBaseParams", False):     # identifier.py:888-892
        w = est.base_wrench_row_mask(S, rows)
        rhs = np.stack((target, model.contactForcesSum), axis=1)
        fbnd = list(getattr(data, "file_boundaries", [0]))
        if opt.get("useTrajectoryWeighting", 0) and len(fbnd) > 2:               # identifier.py:654-679
            R0 = eng.tsqr(st, rhs=rhs, w=w)
            x_pre = est.identify_base_parameters(R0, ic, P, 6 * S, add_contacts=False)[0]
            xf = np.zeros(P)
            xf[ic] = x_pre
            resid = target.reshape(S, rows)[:, :6] - np.asarray(eng.predict(st, xf)).reshape(S, rows)[:, :6]
            w2 = np.zeros((S, rows))
            w2[:, :6] = est.trajectory_row_weights(resid, fbnd, S, skip)
            w = w2.reshape(-1)
        R = eng.tsqr(st, rhs=rhs, w=w)
        xBase = est.identify_base_parameters(R, ic, P, 6 * S, add_contacts=bool(opt["addContacts"]))[0]
    else:
        R = eng.tsqr(st, rhs=np.stack((model.tau, model.contactForcesSum), axis=1))
        xBase = est.identify_base_parameters(R, ic, P, S * rows, add_contacts=bool(opt["addContacts"]))[0]
    xStd = est.find_std_from_base(model.K, xBase)                                # identifier.py:328-341
    if opt["useAPriori"]:
        xStd = xStd + model.xStdModel[model.identified_params]
    out = dict(xBase=xBase, xStd=xStd, independent_cols=ic, num_base_params=model.num_base_params)
    tauEst = np.asarray(eng.predict(st, xStd)).reshape(-1)                       # np.dot(YStd, xStd), identifier.py:135-141
    fb = 6 if opt["floatingBase"] else 0
    vel = data.samples["velocities"][: S * (skip + 1) : skip + 1]
    fric = None
    if opt.get("postIdentifyFriction", False) and (opt["floatingBase"] or opt.get("identifyFrictionSimultaneously", False)):
        ni = model.num_model_params
        xin = np.zeros(P)
        xin[:ni] = xStd[:ni]
        resid2 = (model.torques_stack - np.asarray(eng.predict(st, xin)).reshape(-1)).reshape(S, rows)
        prior = np.array([topo.friction[nm]["f_velocity"] for nm in model.jointNames])
        fric = est.post_identify_friction(resid2, vel, fh.getFrictionSignVelocities(data.samples, opt)[: S * (skip + 1) : skip + 1],
                                          fh.getFrictionSignSeries(data.samples, opt)[: S * (skip + 1) : skip + 1], fb,
                                          deadzone=float(opt.get("frictionVelocityDeadZone", 0.0)), lambda_fv=float(opt.get("frictionFvRegularization", 0.0)),
                                          alpha_fv=float(opt.get("frictionFvRegularizationRelative", 0.0)), fv_apriori=prior)
        for k in ("Fc", "Fv", "off"):
            out["postid_" + k] = fric[k]
    if opt["addContacts"]:
        tauEst = tauEst + model.contactForcesSum
    if not opt.get("identifyFrictionSimultaneously", False) and fric is not None:   # identifier.py:172-200
        sgn = fh.getFrictionSignSeries(data.samples, opt)[: S * (skip + 1) : skip + 1]
        t2 = tauEst.reshape(S, rows)
        t2[:, fb:] += fric["Fc"][None, :] * sgn + fric["Fv"][None, :] * vel + fric["off"][None, :]
        tauEst = t2.reshape(-1)
    out["tauEstimated"] = tauEst.reshape(S, rows)
    out["base_error"] = np.mean(la.norm(model.tauMeasured - out["tauEstimated"], axis=1))
    return out


def compare(out, z, name, tol):
    import numpy.linalg as la

    assert int(out["num_base_params"]) == int(z[name + "_out_num_base_params"])
    assert np.array_equal(np.asarray(out["independent_cols"]), z[name + "_out_independent_cols"])       # identical index sets
    errs = {}
    for k in ("xBase", "xStd", "tauEstimated", "postid_Fc", "postid_Fv", "postid_off"):
        if name + "_out_" + k not in z.files:
            continue
        ref = z[name + "_out_" + k]
        errs[k] = la.norm(np.asarray(out[k]) - ref) / max(la.norm(ref), 1e-300)
    errs["base_error"] = abs(float(out["base_error"]) - float(z[name + "_out_base_error"])) / float(z[name + "_out_base_error"])
    print(name, {k: "%.1e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= tol, (name, k, v)
    return errs
