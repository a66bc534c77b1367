#!/usr/bin/env python3
"""The fused Gram over sample-contiguous images (option gram_lane, csrc/fbr_gram64.h) against the per-sample-image pass: time per 1 M
WALK-MAN samples (blocking and two submissions in flight), kernel split, agreement of the two Grams."""
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

dev = torch.device("cuda", 0)
for robot, floating, S in [("walkman_apriori", True, 1_000_000), ("walkman_apriori", True, 125_000), ("walkman_left_arm", True, 500_000), ("kuka_lwr4", False, 500_000)]:
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", robot + ".topology.json"))
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, floating)[0].items()}
    res = {}
    Gs = {}
    for lane in (0, 1, 0, 1):
        eng = Engine(topo, floating=floating, options={"gram_lane": lane})
        eng.use_torch_stream()
        if lane:
            res["lane_info"] = eng.gram_lane_info(1, S)
            res["program_info"] = eng.gram_program_info(1, S)
        rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        G = eng.gram(st, rhs=rhs)
        Gs[lane] = G.clone()
        for _ in range(3):
            eng.gram(st, rhs=rhs, out=G)
        torch.cuda.synchronize()
        eng.profile_enable(True)
        eng.profile_get()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.gram(st, rhs=rhs, out=G)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        pr = eng.profile_get()
        eng.profile_enable(False)
        outs = [torch.zeros_like(G), torch.zeros_like(G)]
        eng.wait(eng.gram_submit(st, outs[0], rhs=rhs))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = None
        for i in range(20):
            tk = eng.gram_submit(st, outs[i & 1], rhs=rhs)
            if pend is not None:
                eng.wait(pend)
            pend = tk
        eng.wait(pend)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / 20
        res.setdefault(f"lane{lane}", []).append({"blocking_ms": dt * 1e3, "pipelined_ms": dtp * 1e3, "kernel_ms": {k: round(v[0] / 10, 3) for k, v in pr.items() if v[1]},
                                                  "repeat_bitwise": bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], G))})
        eng.close()
    res["rel_diff_lane_vs_images"] = float(torch.linalg.norm(Gs[1] - Gs[0]) / torch.linalg.norm(Gs[0]))
    print(robot, S, json.dumps(res), flush=True)
