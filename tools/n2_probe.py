"""Data.preprocess on the host (SciPy) and on the device: joint channels of a long recording."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flobaroid_amd._lib import Engine
from flobaroid_amd.data import Data
from flobaroid_amd.topology import Topology
eng = Engine(Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json")), floating=True)
rng = np.random.default_rng(0)
S, n, Fs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 29, 200.0
T = np.arange(S) / Fs
Q0 = np.sin(T[:, None] * (0.3 + 0.1 * np.arange(n))) + 0.01 * rng.standard_normal((S, n))
Tau0 = 5 * np.cos(T[:, None] * (0.2 + 0.05 * np.arange(n))) + 0.3 * rng.standard_normal((S, n))
opt = {"filterMedianSize": 5, "useDeg": 0, "num_dofs": n, "filterLowPass1": [8.0, 5], "filterLowPass2": [6.0, 5], "filterLowPass3": [3.0, 4],
       "waitForZeroAcc": 0, "zeroAccThresh": 0.1}
out = {}
for name, engine in (("host", None), ("device", eng), ("device", eng)):
    Q, V, Vdot, Tau = Q0.copy(), np.zeros((S, n)), np.zeros((S, n)), Tau0.copy()
    t0 = time.perf_counter()
    Data(opt).preprocess(Q, V, Vdot, Tau, T, Fs, engine=engine)
    out[name] = (time.perf_counter() - t0, Q, Tau)
print(f"S={S} n={n}: host SciPy {out['host'][0]:.2f} s, device (host arrays in / out) {out['device'][0]:.3f} s, max |dQ| {np.abs(out['host'][1]-out['device'][1]).max():.2e}")
import torch, scipy.signal as sig
b, a = sig.butter(5, 8.0 / (Fs / 2))
Xd = torch.from_numpy(Q0).cuda()
eng.filtfilt(b, a, Xd); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.filtfilt(b, a, Xd)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
t0 = time.perf_counter(); sig.filtfilt(b, a, Q0, axis=0); dth = time.perf_counter() - t0
print(f"filtfilt {S} x {n} resident in HBM: {dt*1e3:.2f} ms ({S*n*8*4/dt/1e9:.0f} GB/s of the 4 sweeps' input) vs scipy {dth:.2f} s")
