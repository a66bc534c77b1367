"""Config 3: WALK-MAN left arm (floating, 13 x 90 block), 500 k samples: TSQR wall time and per-class device times."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_left_arm.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
R = eng.tsqr(st, rhs=rhs)
torch.cuda.synchronize()
eng.profile_enable(True); eng.profile_get()
t0 = time.perf_counter()
for _ in range(5):
    R = eng.tsqr(st, rhs=rhs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
pr = eng.profile_get(); eng.profile_enable(False)
wi = eng.tsqr_work_info(S, k=1)
print(f"left arm S={S}: {dt*1e3:.2f} ms  executed {wi['flop']/dt/1e12:.2f} TF ({wi['flop']/dt/1e12/78.6:.3f})", {k: round(v[0] / 5, 3) for k, v in pr.items() if v[1]}, wi)
