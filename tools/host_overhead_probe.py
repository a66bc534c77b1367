import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
dev = torch.device("cuda", 0)
topo = Topology.load("/root/repo/flobaroid_amd/robots/walkman_apriori.topology.json")
eng = Engine(topo, floating=True); eng.use_torch_stream()
for S in (125000, 30000):
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
    rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
    R = eng.tsqr(st, rhs=rhs); G = eng.gram(st, rhs=rhs)
    outs = [torch.zeros_like(R), torch.zeros_like(R)]
    for name, sub in (("tsqr", lambda o: eng.tsqr_submit(st, o, rhs=rhs)), ("gram", lambda o: eng.gram_submit(st, o, rhs=rhs))):
        for _ in range(3): eng.wait(sub(outs[0]))
        torch.cuda.synchronize()
        t_sub = 0.0; t0 = time.perf_counter()
        n = 20
        for i in range(n):
            a = time.perf_counter(); tk = sub(outs[i & 1]); t_sub += time.perf_counter() - a
            eng.wait(tk)
        torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / n
        print(f"S={S} {name}: host enqueue {t_sub/n*1e3:.3f} ms of {tot*1e3:.3f} ms per call (submitted one at a time)", flush=True)
