import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import synth_range, with_tau
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
topo = Topology.load("flobaroid_amd/robots/walkman_apriori.topology.json")
dev = torch.device("cuda", 0)
eng = Engine(topo, floating=True); eng.use_torch_stream()
st, rhs = with_tau(eng, topo, synth_range(topo, 1000000, 0, 1000000, dev))
P = eng.cols
Ghp = torch.zeros((P + 1, P + 1), dtype=torch.float64).pin_memory()
def run(stx, rhsx, label, back=False):
    Gd = [torch.zeros((P + 1, P + 1), dtype=torch.float64, device=dev) for _ in range(2)]
    for rep in range(2):
        eng.profile_enable(True); eng.profile_get()
        torch.cuda.synchronize(); t0 = time.perf_counter(); pend = None
        for i in range(6):
            tk = eng.gram_submit(stx, Gd[i & 1], rhs=rhsx)
            if pend is not None:
                eng.wait(pend)
                if back: Ghp.copy_(Gd[(i - 1) & 1])
            pend = tk
        eng.wait(pend); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 6
        pr = eng.profile_get(); eng.profile_enable(False)
    print(label, round(dt * 1e3, 2), "ms/step", {k: round(v[0] / 6, 2) for k, v in pr.items() if v[1]})
if len(sys.argv) > 1 and sys.argv[1] == "first":
    sub = {k: v[:200000].contiguous() for k, v in st.items()}
    eng.tsqr(sub, rhs=rhs[:200000 * eng.rows].contiguous())
run(st, rhs, "resident")
hst = {k: v.cpu().pin_memory() for k, v in st.items()}; hrhs = rhs.cpu().pin_memory()
run(hst, hrhs, "pinned  ")
run(hst, hrhs, "pinned + Gram back to the host every step", back=True)
if len(sys.argv) > 1:
    sub = {k: v[:200000].contiguous() for k, v in st.items()}
    eng.tsqr(sub, rhs=rhs[:200000 * eng.rows].contiguous())
    run(st, rhs, "resident after a tsqr call")
    run(hst, hrhs, "pinned after a tsqr call  ")
