#!/usr/bin/env python3
"""fbr_predict / fbr_inverse_dynamics_batch: the fused kernel (option fused_id = 1, csrc/fbr_kinid.h) against the two-kernel path
(kinematics records through HBM + one wave per sample), device-resident inputs, same results required."""
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
import _opts  # noqa: F401,E402
from flobaroid_amd.topology import Topology  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for robot, floating, S in [("walkman_apriori", True, 1_000_000), ("walkman_left_arm", True, 1_000_000), ("kuka_lwr4", False, 1_000_000)]:
        topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", robot + ".topology.json"))
        st_np, _ = synth_states(topo, S, 1, floating)
        st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
        r = {}
        res = {}
        for fused in (0, 1):
            for merge in (1, 0):
                eng = Engine(topo, floating=floating, options={"fused_id": fused, "link_merge": merge})
                eng.use_torch_stream()
                x = np.random.default_rng(3).standard_normal(eng.cols)
                tau = torch.empty((S, eng.rows), dtype=torch.float64, device=dev)
                t = timed(lambda: eng.predict(st, x, out=tau))
                tau2 = torch.empty((S, eng.rows), dtype=torch.float64, device=dev)
                t2 = timed(lambda: eng.inverse_dynamics(st, topo.x_std(), out=tau2))
                r[f"fused{fused}_merge{merge}"] = {"predict_ms": t * 1e3, "inverse_dynamics_ms": t2 * 1e3}
                res[(fused, merge)] = (tau.clone(), tau2.clone())
                del eng
        ref = res[(0, 0)]
        for key, (a, b) in res.items():
            r[f"fused{key[0]}_merge{key[1]}"]["rel_diff_vs_two_kernel_unmerged"] = [
                float((a - ref[0]).abs().max() / ref[0].abs().max()), float((b - ref[1]).abs().max() / ref[1].abs().max())]
        out[robot] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
