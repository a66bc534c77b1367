"""Six WALK-MAN TSQR submissions, two in flight (fbr_tsqr_submit): python tools/tsqr_pipe_probe.py [S]   (under rocprofv3 --kernel-trace for the timeline)"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
outs = [torch.zeros((eng.cols + 1, eng.cols + 1), dtype=torch.float64, device=dev) for _ in range(2)]
eng.wait(eng.tsqr_submit(st, outs[0], rhs=rhs))
torch.cuda.synchronize()
t0 = time.perf_counter()
pend = None
for i in range(6):
    tk = eng.tsqr_submit(st, outs[i & 1], rhs=rhs)
    if pend is not None:
        eng.wait(pend)
    pend = tk
eng.wait(pend)
torch.cuda.synchronize()
print(f"S={S}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms per submission")
