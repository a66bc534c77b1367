import os, sys, numpy as np, torch, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import synth_range, with_tau, _np_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True); eng.use_torch_stream()
dev = torch.device("cuda", 0)
Rs = eng.gram(_np_states(topo, 10000, 99))
Q, RQ, PQ = sla.qr(Rs, pivoting=True, mode="economic")
r = int(np.count_nonzero(np.abs(np.diag(RQ)) > 0.005)); ic = np.sort(PQ[:r]).astype(np.int32); print("r", r)
S = 20000
st, rhs = with_tau(eng, topo, synth_range(topo, S, 0, S, dev))
stn = {k: v.cpu().numpy() for k, v in st.items()}
taun = rhs.cpu().numpy()
# full width
R = eng.tsqr(st, rhs=rhs).cpu().numpy(); print("full-width residual^2 (device)", R[480, 480] ** 2, "expected", 0.0025 * S * 35)
Rn = eng.tsqr(stn, rhs=taun); print("full-width residual^2 (host)", Rn[480, 480] ** 2)
Rb = eng.tsqr(st, rhs=rhs, cols=ic).cpu().numpy(); print("cols residual^2 (device)", Rb[r, r] ** 2)
Rbn = eng.tsqr(stn, rhs=taun, cols=ic); print("cols residual^2 (host)", Rbn[r, r] ** 2)
Y = eng.regressor(stn); t = taun.reshape(-1)
print("lstsq residual^2 on materialised YB", np.linalg.norm(Y[:, ic] @ np.linalg.lstsq(Y[:, ic], t, rcond=None)[0] - t) ** 2)
print("lstsq residual^2 on materialised Y", np.linalg.norm(Y @ np.linalg.lstsq(Y, t, rcond=None)[0] - t) ** 2)
tau2 = eng.inverse_dynamics(stn, topo.x_std()); print("ID dev vs host", np.abs(tau2.reshape(-1, 1) - (taun)).max())
print("Y x_std vs ID", np.abs(Y @ topo.x_std() - tau2.reshape(-1)).max())
for S in (125000, 1000000):
    st, rhs = with_tau(eng, topo, synth_range(topo, S, 0, S, dev))
    Rb = eng.tsqr(st, rhs=rhs, cols=ic).cpu().numpy(); print(S, "cols residual^2 (device)", Rb[r, r] ** 2, "expected", 0.0025 * S * 35)
    G = eng.gram(st, rhs=rhs).cpu().numpy()
    sel = np.concatenate((ic, [480])); Gb = G[np.ix_(sel, sel)]
    x = np.linalg.solve(Gb[:r, :r], Gb[:r, r]); print(S, "residual^2 from the Gram", Gb[r, r] - Gb[:r, r] @ x)
    tau_chk = eng.inverse_dynamics(st, topo.x_std()).reshape(-1, 1)
    print(S, "rhs - ID max", float((rhs - tau_chk).abs().max()))
    Yx = eng.predict(st, topo.x_std()).reshape(-1, 1); print(S, "predict(x_std) - ID max", float((Yx - tau_chk).abs().max()))
    h = S // 2
    R1 = eng.tsqr({k: v[:h].contiguous() for k, v in st.items()}, rhs=rhs[:h * 35].contiguous(), cols=ic)
    R2 = eng.tsqr({k: v[h:].contiguous() for k, v in st.items()}, rhs=rhs[h * 35:].contiguous(), cols=ic, R_in=R1).cpu().numpy()
    print(S, "2-pass streamed residual^2", R2[r, r] ** 2)
