"""Fused pass of one rank's shard (125 k WALK-MAN samples) for the chunk count given by the engine option "min_chunks"
(FBR_OPT_MIN_CHUNKS=.. through tools/_opts.py)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _opts  # noqa: F401,E402
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
for _ in range(100): G = eng.gram(st, rhs=rhs)
torch.cuda.synchronize()
outs = [torch.zeros_like(G), torch.zeros_like(G)]
best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); pend = None
    for i in range(40):
        tk = eng.gram_submit(st, outs[i & 1], rhs=rhs)
        if pend is not None: eng.wait(pend)
        pend = tk
    eng.wait(pend); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40)
print(f"min_chunks={eng.get_option('min_chunks')} S={S}: {best*1e3:.3f} ms per pass = {S/best/1e6:.1f} M samples/s", flush=True)
