import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rh = {k: (torch.randn((S * eng.rows, k), dtype=torch.float64, device=dev) if k else None) for k in (0, 1, 3)}
w = torch.rand((S * eng.rows,), dtype=torch.float64, device=dev) + 0.5
def run(tag, k, wt, plain=False, rows=False):
    if plain: os.environ["FBR_NO_LINK_MERGE"] = "1"
    if rows: os.environ["FBR_LINK_MERGE_ROWS"] = "1"
    R = eng.tsqr(st, rhs=rh[k], w=wt)
    os.environ.pop("FBR_NO_LINK_MERGE", None); os.environ.pop("FBR_LINK_MERGE_ROWS", None)
    os.environ["FBR_NO_LINK_MERGE"] = "1"
    G = eng.gram(st, rhs=rh[k], w=wt)
    os.environ.pop("FBR_NO_LINK_MERGE", None)
    e = float(torch.linalg.norm(R.T @ R - G) / torch.linalg.norm(G))
    print(f"{tag:30s} k={k} w={wt is not None} plain={plain} rows={rows}: {e:.2e}", flush=True)
seq = sys.argv[2] if len(sys.argv) > 2 else "A"
if seq == "A":
    run("first", 1, None); run("again", 1, None); run("weights", 1, w); run("k0", 0, None); run("k1 again", 1, None); run("k3", 3, None)
    run("rows k1", 1, None, rows=True); run("rows k0", 0, None, rows=True); run("rows k1 w", 1, w, rows=True)
elif seq == "B":
    run("first", 1, None); run("plain", 1, None, plain=True); run("after plain", 1, None); run("again", 1, None)
    run("rows", 1, None, rows=True)
