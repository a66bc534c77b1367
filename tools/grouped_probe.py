"""fbr_gram_grouped (one Gram per candidate trajectory, N1) through the column reductions against the unreduced grouped pass."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
for ng, per in ((64, 2000), (16, 8000), (256, 512)):
    S = ng * per
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
    res = {}
    for mode in ("reduced", "plain", "reduced"):
        eng.set_option("reduce_grouped_min_samples", 1e18 if mode == "plain" else 512)   # (groups below the threshold run over all columns)
        for _ in range(5): G = eng.gram_grouped(st, ng)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): G = eng.gram_grouped(st, ng)
        torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 10, G.clone())
    e = float(torch.linalg.norm(res["reduced"][1] - res["plain"][1]) / torch.linalg.norm(res["plain"][1]))
    print(f"{ng} groups x {per} samples: reduced {res['reduced'][0]*1e3:.2f} ms, all columns {res['plain'][0]*1e3:.2f} ms, relerr {e:.1e}", flush=True)
