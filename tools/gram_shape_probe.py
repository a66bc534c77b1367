"""Fused-Gram rate of both compiled kernel shapes (FbrGramConfig) for every bundled robot and friction layout.

    gpurun -- 'python tools/gram_shape_probe.py > gpurun_out/gram_shapes.txt'

The chooser (fbr_gram_build_best, fbr_program.h) takes the two-workgroups-per-CU shape unless it splits the model into more
than FBR_MAX_PARTS_TWO_PER_CU parts; this probe is the measurement behind that rule."""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import synth_states  # noqa: E402
from flobaroid_amd._lib import Engine  # noqa: E402
from flobaroid_amd.topology import Topology  # noqa: E402

dev = torch.device("cuda", 0)
S = int(os.environ.get("PROBE_SAMPLES", 200000))
for path in sorted(glob.glob(os.path.join(ROOT, "flobaroid_amd/robots/*.topology.json"))):
    topo = Topology.load(path)
    name = os.path.basename(path).split(".")[0]
    for floating in ([True, False] if "walkman" in name else [False]):
        st_np, _ = synth_states(topo, S, 1, floating)
        st_np["sign"] = np.tanh(st_np["dq"] / 0.02)
        st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
        for fr, sym in [(0, 1), (1, 1), (1, 0)]:
            line = f"{name:24s} floating={int(floating)} friction={fr} sym={sym}:"
            for shape, code in (("two", 2), ("one", 1), ("", 0)):   # engine option "gram_shape"
                eng = Engine(topo, floating=floating, friction=bool(fr), friction_symmetric=bool(sym), options={"gram_shape": code})
                rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
                eng.gram(st, rhs=rhs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    eng.gram(st, rhs=rhs)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / 3
                info = eng.gram_program_info(1)
                line += f"  {shape or 'auto'}: T={info['parts']:3d} {S / t / 1e6:6.2f} M/s"
                eng.close()
            print(f"{line}   (cols {eng.cols})", flush=True)
