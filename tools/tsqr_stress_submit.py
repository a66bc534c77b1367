"""Stress of the asynchronous submissions (fbr_tsqr_submit / fbr_gram_submit, two in flight, mixed kinds) on WALK-MAN: every result must be
bit-identical to the blocking call's.  python tools/tsqr_stress_submit.py [S] [rounds]"""
import os, sys, time, numpy as np, torch, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
S = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
eng.use_torch_stream()
st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth_states(topo, S, 1, True)[0].items()}
rhs = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
sub = {k: v[:10000].contiguous() for k, v in st.items()}
_, piv = sla.qr(eng.gram(sub).cpu().numpy(), pivoting=True, mode="r")
cols = np.sort(piv[:213]).astype(np.int32)
Ra = eng.tsqr(st, rhs=rhs).clone()
Rc = eng.tsqr(st, rhs=rhs, cols=cols).clone()
G = eng.gram(st, rhs=rhs).clone()
kinds = [("tsqr", Ra, {}), ("cols", Rc, {"cols": cols}), ("gram", G, None)]
outs = {k: [torch.zeros_like(ref) for _ in range(2)] for k, ref, _ in kinds}
rng = np.random.default_rng(0)
pend = []
bad = 0
t0 = time.perf_counter()
n = 0
for r in range(rounds):
    for _ in range(6):
        k, ref, kw = kinds[int(rng.integers(0, 3))]
        buf = outs[k][n & 1]
        try:
            tk = eng.gram_submit(st, buf, rhs=rhs) if kw is None else eng.tsqr_submit(st, buf, rhs=rhs, **kw)
            pend.append((tk, k, ref, buf))
            if len(pend) > 1:
                t, kk, rf, bf = pend.pop(0)
                eng.wait(t)
                if not torch.equal(rf, bf):
                    bad += 1
                    print("MISMATCH", kk, t, flush=True)
        except Exception as e:
            print("ERROR at submission", n, k, repr(e)[:300], flush=True)
            sys.exit(2)
        n += 1
while pend:
    t, kk, rf, bf = pend.pop(0)
    eng.wait(t)
    bad += not torch.equal(rf, bf)
torch.cuda.synchronize()
print(f"{n} submissions, {bad} mismatches, {(time.perf_counter() - t0) / n * 1e3:.2f} ms each")
