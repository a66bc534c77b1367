"""Non-negative least-squares fit of the Gram part cost model (FbrGramConfig c0, cload, cmfma, cimg in csrc/fbr_program.h) to the per-part
cycles printed by tools/gram_timing_probe.py (build container: needs the kernel emulation library for the part statistics).

    python tools/fit_gram_cost.py gpurun_out/gram_timing_*.txt
"""
import ctypes
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import emul_lib  # noqa: E402
from common import load_topo  # noqa: E402
from emul_lib import Emul  # noqa: E402

topo = load_topo("walkman_apriori")
data = {"two": [], "one": []}
for path in sys.argv[1:]:
    cur, parts = None, {}
    runs = []
    for line in open(path):
        m = re.search(r"=== floating=(\d) friction=(\d) sym=(\d) shape=(\w+)", line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4))
            parts = {}
            runs.append((cur, parts))
        m = re.search(r"part (\d+) .*barrier=(\d+) dma_issue=(\d+) mfma=(\d+) \|", line)
        if m and cur:
            parts[int(m.group(1))] = sum(int(x) for x in m.groups()[1:])  # last launch wins
    for (fl, fr, sym, shape), parts in runs:
        em = Emul(topo, floating=bool(fl), fric=bool(fr), fric_sym=bool(sym))
        emul_lib.lib().emul_set_gram_shape({"two": 2, "one": 1}.get(shape, 0))
        out = (ctypes.c_int * 600)()
        T = emul_lib.lib().emul_part_stats(ctypes.byref(em.t), 1, out, 200)
        info = em.program_info(1)
        emul_lib.lib().emul_set_gram_shape(0)
        if T != len(parts):
            print(f"skip {path} {fl, fr, sym, shape}: {T} parts here, {len(parts)} in the log")
            continue
        key = shape if shape in data else ("two" if info["T"] == T and T > 0 and out[2] <= 4608 else "one")
        for p in range(T):
            data[key].append((out[3 * p], out[3 * p + 1], out[3 * p + 2], parts[p]))
for shape, d in data.items():
    if not d:
        continue
    d = np.array(d, float)
    # the fixed cost per sample and part (barrier + DMA issue) is not identifiable from full parts alone: kept at the value
    # fitted when the partition still produced parts of all sizes
    c0 = {"two": 1171.0, "one": 1873.0}[shape]
    A = np.column_stack([d[:, 0], d[:, 1], d[:, 2]])
    from scipy.optimize import nnls
    x, _ = nnls(A, d[:, 3] - c0)  # non-negative coefficients: the partitioner extrapolates with them
    x = np.concatenate([[c0], x])
    r = (c0 + A @ x[1:] - d[:, 3]) / d[:, 3]
    print(f"{shape}: {len(d)} parts  c0={x[0]:.0f} cload={x[1]:.1f} cmfma={x[2]:.1f} cimg={x[3]:.2f}  rms {np.sqrt(np.mean(r ** 2)):.3f} max {np.abs(r).max():.3f}")
