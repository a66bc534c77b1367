"""world_size-2 gloo tests of the multi-GPU plumbing (sample sharding, Gram all-reduce, TSQR tree)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from common import ROOT
from flobaroid_amd.dist import shard_range


def test_shard_range_partitions():
    for S in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(S, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == S
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = """
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from flobaroid_amd.dist import shard_range, allreduce_gram, tsqr_tree
from common import load_topo, random_states
from oracle.oracle import OracleModel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
t = load_topo("kuka_lwr4")
om = OracleModel(t)
S = 301
st = random_states(t, S, np.random.default_rng(5), 0, use_limits=True)   # same seed on every rank
Y = om.regressor(st)
tau = om.inverse_dynamics(st, t.x_std()).reshape(-1, 1)
A = np.hstack([Y, tau])
a, b = shard_range(S, rank, world)
Al = A[a * om.rows:b * om.rows]
# Gram path: local reduction (stands in for fbr_gram_accumulate on this rank's shard) + all-reduce
G = allreduce_gram(torch.from_numpy(Al.T @ Al))
assert np.linalg.norm(G.numpy() - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)
# TSQR path: local factor + binary tree with a Householder merge (stands in for fbr_tsqr_merge)
def merge(Ra, Rb):
    return torch.from_numpy(np.linalg.qr(np.vstack([Ra.numpy(), Rb.numpy()]), mode="r"))
R = tsqr_tree(torch.from_numpy(np.linalg.qr(Al, mode="r")), merge)
assert np.all(np.tril(R.numpy(), -1) == 0)
assert np.linalg.norm(R.numpy().T @ R.numpy() - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)
Rs = [torch.empty_like(R) for _ in range(world)]
dist.all_gather(Rs, R)
assert all(torch.equal(Rs[0], r) for r in Rs)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(_WORKER.format(root=ROOT)))
    port = _free_port()
    procs = []
    for world in (2, 3):
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + world))
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=240)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        procs = []
