import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
ROOT="/root/repo"
dev = torch.device("cuda", 0)
for name, fl, S in (("kuka_lwr4", False, 50000), ("walkman_apriori", True, 125000)):
    topo = Topology.load(os.path.join(ROOT, f"flobaroid_amd/robots/{name}.topology.json"))
    for env in ({"gram_rhs_tile": 1}, {}, {"gram_rhs_tile": 1}, {}, {"link_merge": 0, "gram_rhs_tile": 1}, {}):   # engine options of the variant
        eng = Engine(topo, floating=fl, options=env)
        eng.use_torch_stream()
        st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, fl)[0].items()}
        rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
        for _ in range(200): G = eng.gram(st, rhs=rhs)   # (clocks up)
        torch.cuda.synchronize()
        eng.profile_enable(True); eng.profile_get()
        t0 = time.perf_counter()
        for _ in range(20): G = eng.gram(st, rhs=rhs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        pr = eng.profile_get(); eng.profile_enable(False)
        print("   ", {k: (round(v[0] / 20, 3), v[1] // 20) for k, v in pr.items() if v[1]})
        outs = [torch.zeros_like(G), torch.zeros_like(G)]
        eng.wait(eng.gram_submit(st, outs[0], rhs=rhs)); torch.cuda.synchronize()
        t0 = time.perf_counter(); pend = None
        for i in range(20):
            tk = eng.gram_submit(st, outs[i & 1], rhs=rhs)
            if pend is not None: eng.wait(pend)
            pend = tk
        eng.wait(pend); torch.cuda.synchronize(); dtp = (time.perf_counter() - t0) / 20
        print(name, S, env, f"blocking {dt*1e3:.3f} ms  pipelined {dtp*1e3:.3f} ms = {S/dtp/1e6:.1f} M/s", eng.link_merge_info()["reduced_cols"], eng.gram_program_info(1), flush=True)
        eng.close()
