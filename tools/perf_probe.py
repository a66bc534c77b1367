#!/usr/bin/env python3
"""Per-kernel timings of every C-ABI entry point on one GPU (device-resident inputs), for tuning."""
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
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    which = sys.argv[1:] or ["regressor", "gram", "tsqr", "id"]
    dev = torch.device("cuda", 0)
    out = {}
    for robot, floating, S_reg, S_gram, S_tsqr in [("walkman_apriori", True, 100_000, 500_000, 150_000),
                                                   ("walkman_left_arm", True, 500_000, 500_000, 200_000),
                                                   ("kuka_lwr4", False, 500_000, 500_000, 200_000)]:
        topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", robot + ".topology.json"))
        eng = Engine(topo, floating=floating)
        eng.use_torch_stream()
        eng.profile_enable(True)
        Smax = max(S_reg, S_gram, S_tsqr)
        st_np, _ = synth_states(topo, Smax, 1, floating)
        st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
        sub = lambda S: {k: v[:S].contiguous() for k, v in st.items()}
        rows, P = eng.rows, eng.cols
        r = {"rows": rows, "cols": P}
        if "regressor" in which:
            s = sub(S_reg)
            Y = torch.empty((S_reg * rows, P), dtype=torch.float64, device=dev)
            eng.profile_get()
            t = timed(lambda: eng.regressor(s, out=Y))
            pr = eng.profile_get()
            kms = pr["regressor"][0] / max(pr["regressor"][1], 1) * (pr["regressor"][1] / 4)
            byts = 8.0 * rows * P * S_reg
            r["regressor"] = {"S": S_reg, "wall_ms": t * 1e3, "kernel_ms": pr["regressor"][0] / 4, "kin_ms": pr["kin"][0] / 4,
                              "GBps_kernel": byts / (pr["regressor"][0] / 4 * 1e-3) / 1e9, "samples_per_s": S_reg / t}
            del Y
        if "gram" in which:
            s = sub(S_gram)
            rhs = torch.randn((S_gram * rows, 1), dtype=torch.float64, device=dev)
            eng.profile_get()
            t = timed(lambda: eng.gram(s, rhs=rhs))
            pr = eng.profile_get()
            r["gram"] = {"S": S_gram, "wall_ms": t * 1e3, "kernel_ms": pr["gram"][0] / 4, "samples_per_s": S_gram / t,
                         "info": eng.gram_program_info(1)}
        if "tsqr" in which:
            s = sub(S_tsqr)
            rhs = torch.randn((S_tsqr * rows, 2), dtype=torch.float64, device=dev)
            eng.profile_get()
            t = timed(lambda: eng.tsqr(s, rhs=rhs), reps=2)
            pr = eng.profile_get()
            Pa = P + 2
            flop = 2.0 * S_tsqr * rows * Pa * Pa
            r["tsqr"] = {"S": S_tsqr, "wall_ms": t * 1e3, "tsqr_ms": pr["tsqr"][0] / 3, "samples_per_s": S_tsqr / t,
                         "GFps": flop / t / 1e9}
        if "id" in which:
            s = sub(S_gram)
            x = topo.x_std()
            eng.profile_get()
            t = timed(lambda: eng.inverse_dynamics(s, x))
            pr = eng.profile_get()
            r["id"] = {"S": S_gram, "wall_ms": t * 1e3, "kernel_ms": pr["id"][0] / 4, "samples_per_s": S_gram / t}
        out[robot] = r
        eng.close()
        del st
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
