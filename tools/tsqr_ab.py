"""A/B of TSQR kernel-shape switches inside one process (the switches are options of the model handle, include/fbr.h):
python tools/tsqr_ab.py [S] [option=val,option2=val2 ...]   -- each argument after S is one variant (comma-separated settings, '-' = defaults).
Prints wall time, per-class device time and ||R^T R - G|| / ||G|| against the fused Gram of the same samples."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
import _opts  # noqa: F401  (FBR_OPT_<KEY>=value -> engine options)
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
variants = sys.argv[2:] or ["-"]
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
G = eng.gram(st, rhs=rhs)
gn = float(torch.linalg.norm(G))
Rfirst = None
for var in variants:
    sets = [] if var == "-" else [kv.split("=") for kv in var.split(",")]
    saved = {k: eng.get_option(k) for k, _ in sets}
    for k, v in sets:
        eng.set_option(k, float(v))
    R = eng.tsqr(st, rhs=rhs)
    err = float(torch.linalg.norm(R.T @ R - G)) / gn
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_get()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        R2 = eng.tsqr(st, rhs=rhs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pr = eng.profile_get(); eng.profile_enable(False)
    same = bool(torch.equal(R, R2))
    if Rfirst is None:
        Rfirst = R.clone()
    same_as_first = bool(torch.equal(R, Rfirst))
    # the same calls submitted two in flight (fbr_tsqr_submit): the next call's kinematics / first writer run beside the trees
    outs = [torch.zeros_like(R), torch.zeros_like(R)]
    n2 = 6
    eng.wait(eng.tsqr_submit(st, outs[0], rhs=rhs))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = None
    for i in range(n2):
        tk = eng.tsqr_submit(st, outs[i & 1], rhs=rhs)
        if pend is not None:
            eng.wait(pend)
        pend = tk
    eng.wait(pend)
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / n2
    same = same and bool(torch.equal(R, outs[0])) and bool(torch.equal(R, outs[1]))
    wi = eng.tsqr_work_info(S, k=1)
    print(f"pipelined {dtp*1e3:8.2f} ms ({wi['flop']/dtp/1e12/78.6:.3f}) |", end=" ")
    print(f"{var:40s} S={S} {dt*1e3:8.2f} ms  executed {wi['flop']/dt/1e12:6.2f} TF ({wi['flop']/dt/1e12/78.6:.3f})  relerr {err:.2e} repeat-bitwise {same} bitwise-equal-to-first-variant {same_as_first} |",
          {k: round(v[0] / reps, 2) for k, v in pr.items() if v[1]}, flush=True)
    for k, v in saved.items():
        eng.set_option(k, v)
