"""One Householder-TSQR call on WALK-MAN (150 k samples, 1 rhs column) and one on the left arm (500 k) -- the workload of the TSQR PMC passes."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
dev = torch.device("cuda", 0)
for robot, S in (("walkman_apriori", 150000), ("walkman_left_arm", 500000)):
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots", robot + ".topology.json"))
    eng = Engine(topo, floating=True)
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
    rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    eng.tsqr(st, rhs=rhs)
    print(robot, S, eng.tsqr_work_info(S, k=1))
    eng.close()
