"""Fused Gram rate of one batch of S WALK-MAN samples (device resident) -- run once per FBR_OPT_MIN_CHUNKS / FBR_OPT_CHUNK_SAMPLES value
(engine options "min_chunks" / "chunk_samples" through tools/_opts.py; the library reads no environment)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _opts  # noqa: F401,E402  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
dev = torch.device("cuda", 0)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
G = torch.zeros((eng.cols + 1, eng.cols + 1), dtype=torch.float64, device=dev)
for _ in range(3):
    eng.gram(st, rhs=rhs, out=G)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    eng.gram(st, rhs=rhs, out=G)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"S={S} min_chunks={eng.get_option('min_chunks')} chunk_samples={eng.get_option('chunk_samples')} ms={dt*1e3:.3f} Msamples/s={S/dt/1e6:.3f}")
