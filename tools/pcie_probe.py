"""PCIe-inclusive rate of the fused pass: host NumPy states / rhs in, host Gram out (FBR_HOST buffers through the C-ABI)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_states
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
topo = Topology.load(os.path.join(ROOT, "flobaroid_amd/robots/walkman_apriori.topology.json"))
eng = Engine(topo, floating=True)
S = 1_000_000
st, rng = synth_states(topo, S, 1, True)
rhs = rng.standard_normal((S * eng.rows, 1))
eng.gram(st, rhs=rhs)
t0 = time.perf_counter()
for _ in range(3):
    G = eng.gram(st, rhs=rhs)
dt = (time.perf_counter() - t0) / 3
host_bytes = sum(v.nbytes for v in st.values()) + rhs.nbytes
print(json.dumps({"samples": S, "seconds": dt, "samples_per_s": S / dt, "host_bytes_in": host_bytes, "H2D_GB_per_s_if_only_copy": host_bytes / dt / 1e9}))
