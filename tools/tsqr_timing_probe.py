import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
S = 60000
st_np, _ = synth_states(topo, S, 1, True)
dev = torch.device("cuda", 0)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
rhs = torch.randn((S * eng.rows, 2), dtype=torch.float64, device=dev)
eng.tsqr(st, rhs=rhs)
eng.set_option("tsqr_timing", 1)
eng.tsqr(st, rhs=rhs)
