import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
S = 80000
st_np, _ = synth_states(topo, S, 1, True)
dev = torch.device("cuda", 0)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
eng.gram(st, rhs=rhs)
os.environ["FBR_GRAM_TIMING"] = "1"
eng.gram(st, rhs=rhs)
