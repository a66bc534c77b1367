"""Link merging (fbr_api.hip build_reduction) against the unmerged path on the same inputs: Gram and TSQR factor, errors and times.
python tools/linkmerge_probe.py [S]"""
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
for k in (1, 0, 3):
    rhs = torch.randn((S * eng.rows, k), dtype=torch.float64, device=dev) if k else None
    w = torch.rand((S * eng.rows,), dtype=torch.float64, device=dev) + 0.5
    for wt in (None, w):
        res = {}
        for mode in ("merged", "plain") + (("nore",) if os.environ.get("PROBE_NOREGROUP") else ()):
            eng.set_option("link_merge", 0 if mode == "plain" else 1)   # options of the handle (the library reads no environment)
            eng.set_option("regroup", 0 if mode == "nore" else 1)
            G = eng.gram(st, rhs=rhs, w=wt)
            R = eng.tsqr(st, rhs=rhs, w=wt)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); G2 = eng.gram(st, rhs=rhs, w=wt); torch.cuda.synchronize(); tg = time.perf_counter() - t0
            t0 = time.perf_counter(); R2 = eng.tsqr(st, rhs=rhs, w=wt); torch.cuda.synchronize(); tr = time.perf_counter() - t0
            res[mode] = (G, R, tg, tr, bool(torch.equal(G, G2)), bool(torch.equal(R, R2)))
        Gm, Rm, tgm, trm, sg, sr = res["merged"]
        Gp, Rp, tgp, trp, _, _ = res["plain"]
        gn = float(torch.linalg.norm(Gp))
        eg = float(torch.linalg.norm(Gm - Gp)) / gn
        asym = float(torch.linalg.norm(Gm - Gm.T)) / gn
        er = float(torch.linalg.norm(Rm.T @ Rm - Gp)) / gn
        erp = float(torch.linalg.norm(Rp.T @ Rp - Gp)) / gn
        low = float(torch.linalg.norm(torch.tril(Rm, -1)))
        if "nore" in res:
            print(f"   merged only: gram {res['nore'][2]*1e3:.2f} ms tsqr {res['nore'][3]*1e3:.2f} ms")
        print(f"k={k} weights={wt is not None} S={S}: gram {tgm*1e3:.2f} vs {tgp*1e3:.2f} ms, relerr {eg:.2e} asym {asym:.1e} repeat {sg} | "
              f"tsqr {trm*1e3:.2f} vs {trp*1e3:.2f} ms, RtR-G merged {er:.2e} plain {erp:.2e} lower {low:.1e} repeat {sr}", flush=True)
eng.set_option("link_merge", 1); eng.set_option("regroup", 1)
