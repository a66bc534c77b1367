"""A/B of libfbr builds (tools/_build/*.so via FBR_LIB_PATH): WALK-MAN TSQR of all columns (config 4 shape) and of the base columns
(config 5 shape), wall time + per-kernel-class device times.  python tools/ab_lib.py [S]"""
import os, sys, time, numpy as np, torch, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
G = eng.gram({k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, 10000, 7, True)[0].items()}).cpu().numpy()
Rq, piv = sla.qr(G, pivoting=True, mode="r")
cols = np.sort(piv[:213]).astype(np.int32)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
for name, kw in (("all 481 columns", {}), ("214 base columns", {"cols": cols})):
    R0 = eng.tsqr(st, rhs=rhs, **kw)
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_get()
    t0 = time.perf_counter()
    for _ in range(3):
        R = eng.tsqr(st, rhs=rhs, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    pr = eng.profile_get(); eng.profile_enable(False)
    print(f"{name}: S={S} {dt*1e3:.2f} ms  {S/dt/1e6:.2f} M samples/s |", {k: round(v[0] / 3, 2) for k, v in pr.items() if v[1]}, "| checksum", float(torch.linalg.norm(R)))
