#!/usr/bin/env python3
"""96-row wave-private level-0 folds (option tsqr_narrow_tall, one wave per SIMD) against the 48-row shape (two waves per SIMD):
left arm 500 k samples (BASELINE configs[2]), KUKA 500 k, and the WALK-MAN 1 M call whose arm / leg groups take the narrow kernels."""
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
for robot, floating, S in [("walkman_left_arm", True, 500_000), ("kuka_lwr4", False, 500_000), ("walkman_apriori", True, 1_000_000), ("walkman_apriori", True, 125_000)]:
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", robot + ".topology.json"))
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, floating)[0].items()}
    r = {}
    Rs = {}
    for tall in (0, 1, 0, 1):
        eng = Engine(topo, floating=floating, options={"tsqr_narrow_tall": tall})
        eng.use_torch_stream()
        rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        R = eng.tsqr(st, rhs=rhs)
        G = eng.gram(st, rhs=rhs)
        err = float(torch.linalg.norm(R.T @ R - G) / torch.linalg.norm(G))
        eng.profile_enable(True)
        eng.profile_get()
        t = timed(lambda: eng.tsqr(st, rhs=rhs))
        pr = eng.profile_get()
        r.setdefault(f"tall{tall}", []).append({"ms": t * 1e3, "relerr_RtR_vs_gram": err, "kernel_ms": {k: round(v[0] / 8, 3) for k, v in pr.items() if v[1]}})
        Rs[tall] = R.clone()
        eng.close()
    r["max_abs_diff_R_between_shapes_rel"] = float((Rs[0].abs() - Rs[1].abs()).abs().max() / Rs[0].abs().max())
    out[f"{robot}_{S}"] = r
print(json.dumps(out, indent=1))
