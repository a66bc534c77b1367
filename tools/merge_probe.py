"""fbr_tsqr_merge of two WALK-MAN-sized triangles (one node of the rank tree across GPUs): time and agreement of the two merge paths."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
for n in (481, 214, 92):
    g = torch.Generator(device="cuda").manual_seed(n)
    Ra = torch.triu(torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g))
    Rb = torch.triu(torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g))
    res = {}
    for var in ("pipelined", "one_wg"):
        eng.set_option("tsqr_tree_one_wg", 1 if var == "one_wg" else 0)
        R = eng.tsqr_merge(Ra, Rb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            R = eng.tsqr_merge(Ra, Rb)
        torch.cuda.synchronize()
        res[var] = ((time.perf_counter() - t0) / 20 * 1e3, R.clone())
    G = Ra.T @ Ra + Rb.T @ Rb
    Rp = res["pipelined"][1]
    print(f"n={n}: pipelined {res['pipelined'][0]:.3f} ms, one workgroup {res['one_wg'][0]:.3f} ms, bitwise equal {bool(torch.equal(Rp, res['one_wg'][1]))}, "
          f"||R^T R - G||/||G|| = {float(torch.linalg.norm(Rp.T @ Rp - G) / torch.linalg.norm(G)):.1e}")
