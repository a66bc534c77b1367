"""Householder TSQR of S WALK-MAN samples (device resident, 1 rhs column): wall time of the call and its work counters.
python tools/tsqr_probe.py [S] [reps]   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
R = torch.zeros((eng.cols + 1, eng.cols + 1), dtype=torch.float64, device=dev)
eng.tsqr(st, rhs=rhs, out=R)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.tsqr(st, rhs=rhs, out=R)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
wi = eng.tsqr_work_info(S, k=1)
print(f"S={S} tsqr {dt*1e3:.2f} ms  {S/dt/1e6:.3f} M samples/s  executed {wi['flop']/dt/1e12:.2f} TFLOP/s  dense model {2.0*S*eng.rows*(eng.cols+1)**2/dt/1e12:.1f} TFLOP/s", wi)
