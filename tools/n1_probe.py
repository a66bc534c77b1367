"""Timing of the SURVEY 8(f) N1 pieces (trajectory-optimiser inner loop) on one GPU: grouped Gram vs one call per
candidate, and the finite-difference score sweep.  python tools/n1_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology

dev = torch.device("cuda", 0)
out = {}
for robot, fl in [("kuka_lwr4", False), ("walkman_apriori", True)]:
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", robot + ".topology.json"))
    eng = Engine(topo, floating=fl)
    ng, Sg = 64, 2000
    st_np, rng = synth_states(topo, ng * Sg, 3, fl)
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

    t_grouped = timed(lambda: eng.gram_grouped(st, ng))
    subs = [{k: v[g * Sg:(g + 1) * Sg].contiguous() for k, v in st.items()} for g in range(ng)]
    t_single = timed(lambda: [eng.gram(s) for s in subs], reps=2)
    S = 2000
    sub = subs[0]
    W = torch.randn((S * eng.rows, eng.cols), dtype=torch.float64, device=dev)
    t_fd = timed(lambda: eng.fd_scores(sub, W, 1e-6))
    n = topo.num_dofs
    out[robot] = {"candidates": ng, "samples_per_candidate": Sg, "gram_grouped_ms": t_grouped * 1e3, "one_call_per_candidate_ms": t_single * 1e3,
                  "fd_scores_samples": S, "fd_regressor_evaluations": S * (1 + 3 * n), "fd_scores_ms": t_fd * 1e3,
                  "fd_evaluations_per_s": S * (1 + 3 * n) / t_fd}
    eng.close()
print(json.dumps(out, indent=1))
