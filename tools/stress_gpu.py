"""Randomised GPU stress: random kinematic trees x option combinations x both Gram kernel shapes, fused Gram and TSQR against the
oracle (gpurun -- python tools/stress_gpu.py).  Prints one line per case and the number of failures.  run_case(seed) is also what
tests/test_gpu_determinism.py runs, one seed per parametrised case."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def run_case(seed):
    """[(description, relative Gram error, relative TSQR error, program info)] for both kernel shapes of one random case (empty: the draw is skipped)."""
    from common import random_topology, random_states
    from flobaroid_amd._lib import Engine
    from oracle.oracle import OracleModel

    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(2, 58)); branch = float(rng.random()); fl = int(rng.integers(0, 2)); fr = int(rng.integers(0, 2)); sym = int(rng.integers(0, 2))
    grav = int(rng.random() < 0.15); strb = 0.05 if rng.random() < 0.3 else 0.0
    t = random_topology(rng, L, p_fixed=float(rng.random()) * 0.4, branchiness=branch)
    if t.num_dofs == 0 or t.num_dofs + 6 * fl > 60:
        return []
    k = int(rng.integers(0, 17))
    om = OracleModel(t, floating=bool(fl), fric=bool(fr), fric_sym=bool(sym), grav_only=bool(grav), stribeck=strb)
    S = int(rng.integers(1, 400)) if seed % 3 else int(rng.integers(1500, 4000))  # (the long ones reach the depth-ordered TSQR columns)
    st = random_states(t, S, rng, fl)
    if grav: st["dq"][:] = 0; st["ddq"][:] = 0
    st["sign"] = np.tanh(st["dq"] / 0.02)
    Y = om.regressor(st, st["sign"])
    rhs = rng.standard_normal((Y.shape[0], k)) if k else None
    w = rng.random(Y.shape[0]) + 0.5 if rng.random() < 0.5 else None
    A = Y if rhs is None else np.hstack([Y, rhs])
    if w is not None: A = A * w[:, None]
    Go = A.T @ A
    out = []
    # the variants are options of the model handle (the library reads no environment): both compiled shapes of the Gram kernel, the
    # column reductions forced (the random cases are small: the reduced paths are what is stressed), the row-group TSQR from 1 sample on
    for shape, code in (("two", 2), ("one", 1)):
        eng = Engine(t, floating=bool(fl), friction=bool(fr), friction_symmetric=bool(sym), gravity_only=bool(grav), stribeck_velocity=strb,
                     options={"gram_shape": code, "reduce_min_work": 0, "tsqr_group_min_samples": 1})
        assert eng.get_option("gram_shape") == code and eng.get_option("reduce_min_work") == 0
        G = eng.gram(st, rhs=rhs, w=w)
        err = np.linalg.norm(G - Go) / max(np.linalg.norm(Go), 1e-300)
        info = eng.gram_program_info(k, S)
        R = eng.tsqr(st, rhs=rhs, w=w) if shape == "two" and eng.cols + k <= 768 else None  # (the TSQR kernels' column limit)
        e2 = np.linalg.norm(R.T @ R - Go) / max(np.linalg.norm(Go), 1e-300) if R is not None else 0
        out.append((f"seed {seed} L={L} n={t.num_dofs} fl={fl} fr={fr}{'s' if sym else 'a'} g={grav} st={strb} k={k} S={S} w={w is not None} {shape}: "
                    f"parts {info['parts']} gram {err:.1e} tsqr {e2:.1e}", err, e2, info))
        eng.close()
    return out


if __name__ == "__main__":
    bad = 0
    first = int(os.environ.get("FBR_STRESS_FIRST", 0))
    for seed in range(first, first + int(os.environ.get("FBR_STRESS_SEEDS", 40))):
        for desc, err, e2, _info in run_case(seed):
            flag = "" if err < 1e-11 and e2 < 1e-9 else "  <-- BAD"
            bad += bool(flag)
            print(desc + flag, flush=True)
    print("BAD", bad)
