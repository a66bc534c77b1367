"""TSQR of the base regressor [Y[:, independent columns] | tau] of S WALK-MAN samples (config 5's call, one pass): wall time.
python tools/tsqr_cols_probe.py [S] [reps]   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time, numpy as np, torch, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
G = eng.gram({k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, 10000, 7, True)[0].items()}).cpu().numpy()
Rq, piv = sla.qr(G, pivoting=True, mode="r")
rank = int(np.sum(np.abs(np.diag(Rq)) > 0.005))
cols = np.sort(piv[:rank]).astype(np.int32)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
R = torch.zeros((rank + 1, rank + 1), dtype=torch.float64, device=dev)
eng.tsqr(st, rhs=rhs, cols=cols, out=R)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.tsqr(st, rhs=rhs, cols=cols, out=R)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"S={S} base columns {rank}: tsqr {dt*1e3:.2f} ms  {S/dt/1e6:.3f} M samples/s", eng.tsqr_work_info(S, k=1, cols=cols))
