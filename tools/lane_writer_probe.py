#!/usr/bin/env python3
"""fbr_tsqr with the one-lane-per-sample regressor writer (option tsqr_lane_writer: kinematics fused, column-major chunks) against the
kinematics kernel + workgroup-per-sample writers, WALK-MAN 1 M samples and one rank's shard (125 k): time, kernel split, R^T R vs Gram."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states  # noqa: E402
from flobaroid_amd._lib import Engine  # noqa: E402
from flobaroid_amd.topology import Topology  # noqa: E402


def timed(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


dev = torch.device("cuda", 0)
out = {}
cases = [("walkman_apriori", True, 1_000_000), ("walkman_apriori", True, 125_000), ("walkman_left_arm", True, 500_000), ("kuka_lwr4", False, 500_000)]
for robot, floating, S in cases:
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", robot + ".topology.json"))
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, floating)[0].items()}
    r = {}
    for lane in (0, 1, 0, 1):
        eng = Engine(topo, floating=floating, options={"tsqr_lane_writer": lane})
        eng.use_torch_stream()
        rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        w = torch.rand((S * eng.rows,), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(6)) + 0.5
        R = eng.tsqr(st, rhs=rhs)
        G = eng.gram(st, rhs=rhs)
        err = float(torch.linalg.norm(R.T @ R - G) / torch.linalg.norm(G))
        Rw = eng.tsqr(st, rhs=rhs, w=w)
        Gw = eng.gram(st, rhs=rhs, w=w)
        errw = float(torch.linalg.norm(Rw.T @ Rw - Gw) / torch.linalg.norm(Gw))
        eng.profile_enable(True)
        eng.profile_get()
        t = timed(lambda: eng.tsqr(st, rhs=rhs))
        pr = eng.profile_get()
        r.setdefault(f"lane{lane}", []).append({"ms": t * 1e3, "relerr_RtR_vs_gram": err, "relerr_weighted": errw,
                                                "kernel_ms": {k: round(v[0] / 8, 3) for k, v in pr.items() if v[1]}})
        eng.close()
    out[f"{robot}_{S}"] = r
    print(robot, S, json.dumps(r), flush=True)
