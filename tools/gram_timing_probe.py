"""Per-part phase cycles of the streaming Gram kernel (engine option "gram_timing": the diagnostic instantiation), WALK-MAN floating base.

    gpurun -- 'python tools/gram_timing_probe.py [fixed] [friction] [asym] > gpurun_out/gram_timing.txt 2>&1'

FBR_OPT_GRAM_SHAPE=1|2 (tools/_opts.py -> engine option "gram_shape") forces a kernel shape.  tools/fit_gram_cost.py fits the part cost model (FbrGramConfig) to this output."""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from bench import synth_states  # noqa: E402
from flobaroid_amd._lib import Engine  # noqa: E402
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology  # noqa: E402

fric = "friction" in sys.argv
sym = "asym" not in sys.argv
floating = "fixed" not in sys.argv
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=floating, friction=fric, friction_symmetric=sym)
S = 60000
st_np, _ = synth_states(topo, S, 1, floating)
st_np["sign"] = np.tanh(st_np["dq"] / 0.02)
dev = torch.device("cuda", 0)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
eng.gram(st, rhs=rhs)
torch.cuda.synchronize()
print(f"=== floating={int(floating)} friction={int(fric)} sym={int(sym)} shape={eng.get_option('gram_shape'):.0f} {eng.gram_program_info(1)}", file=sys.stderr, flush=True)
eng.set_option("gram_timing", 1)
eng.gram(st, rhs=rhs)
torch.cuda.synchronize()
