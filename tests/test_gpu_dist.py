"""Multi-process paths on a real GPU, through libfbr (not NumPy stand-ins).

The GPU box has ONE device, and RCCL refuses two ranks on one device, so two processes share cuda:0 and talk over gloo: every
rank reduces its shard with ``fbr_gram_accumulate`` / ``fbr_tsqr`` on the GPU, the Gram is all-reduced, and the rank tree of
``flobaroid_amd.dist.tsqr_tree`` merges the per-rank factors with ``fbr_tsqr_merge`` ON THE GPU (factors travel as host
tensors: gloo has no CUDA send / recv).  The 8-GPU RCCL run itself is the driver's (bench.py --gpus N).
"""
import os
import subprocess
import sys
import textwrap

import pytest

from common import ROOT
from test_dist import _free_port

pytestmark = pytest.mark.gpu

_WORKER = """
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from flobaroid_amd.dist import shard_range, allreduce_gram, tsqr_tree, selfcheck
from flobaroid_amd._lib import Engine
from common import load_topo, random_states
from oracle.oracle import OracleModel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
for name, fl, S in (("walkman_apriori", 1, 700), ("kuka_lwr4", 0, 2001)):   # wide (8-wave) and narrow (wave-private) merge kernels
    t = load_topo(name)
    eng = Engine(t, floating=fl, device=0)
    om = OracleModel(t, floating=fl)
    st = random_states(t, S, np.random.default_rng(5), fl, use_limits=True)   # same seed on every rank
    tau = om.inverse_dynamics(st, t.x_std()).reshape(-1, 1)
    a, b = shard_range(S, rank, world)
    sl = {{k: v[a:b] for k, v in st.items()}}
    # Gram path: this rank's shard on the GPU, then the all-reduce
    G = allreduce_gram(torch.from_numpy(eng.gram(sl, rhs=tau[a * eng.rows:b * eng.rows])))
    G1 = eng.gram(st, rhs=tau)     # the whole batch on one GPU
    assert np.linalg.norm(G.numpy() - G1) <= 1e-12 * np.linalg.norm(G1)
    A = np.hstack([om.regressor(st), tau])
    assert np.linalg.norm(G1 - A.T @ A) <= 1e-11 * np.linalg.norm(A.T @ A)
    # TSQR path: per-rank factor on the GPU, rank tree with fbr_tsqr_merge on the GPU
    merges = []
    def merge(Ra, Rb):
        merges.append(1)
        return torch.from_numpy(eng.tsqr_merge(Ra.numpy(), Rb.numpy()))
    R = tsqr_tree(torch.from_numpy(eng.tsqr(sl, rhs=tau[a * eng.rows:b * eng.rows])), merge).numpy()
    assert np.all(np.tril(R, -1) == 0)
    assert np.linalg.norm(R.T @ R - G1) <= 1e-11 * np.linalg.norm(G1), (name, np.linalg.norm(R.T @ R - G1) / np.linalg.norm(G1))
    assert merges or rank != 0   # rank 0 is the root of the tree: it merged at least once
    Rs = [torch.empty(R.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Rs, torch.from_numpy(R))
    assert all(torch.equal(Rs[0], r) for r in Rs)
    # the self-check bench.py runs at the start of every N > 1 run, with fbr_tsqr_merge on the GPU as the merge
    chk = selfcheck(None, lambda Ra, Rb: torch.from_numpy(eng.tsqr_merge(Ra.numpy(), Rb.numpy())), timeout=120.0)
    assert chk["world"] == world and chk["tsqr_tree_relerr"] <= 1e-12, chk
    eng.close()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_shards_allreduce_and_rank_tree_through_libfbr(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(_WORKER.format(root=ROOT)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_bench_gpus_more_than_devices_fails_loudly():
    """bench.py --gpus N on a box with fewer devices must refuse (no silent one-rank run labelled n_gpus: N)."""
    import torch

    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"refusing" in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_bench_one_rank_under_torchrun_rccl():
    """The RCCL path of bench.py on the one available device: launched the way the driver launches N > 1."""
    import json

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--samples", "40000", "--no-cpu-baseline",
           "--sustain-seconds", "0", "--no-secondary"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    js = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(js) == 1 and js[0]["n_gpus"] == 1 and js[0]["value"] > 1e5 and 0 < js[0]["roofline"]["frac"] < 1


def test_bench_two_gpus_over_rccl(tmp_path):
    """The N > 1 path on real devices: ``bench.py --gpus 2`` starts two ranks, one per GPU, over RCCL -- sharded fused Gram +
    all-reduce, per-rank ``fbr_tsqr`` and the rank tree of ``dist.tsqr_tree`` on DEVICE-RESIDENT factors (RCCL send / recv / broadcast
    of CUDA tensors, ``fbr_tsqr_merge`` on the receiver; no host hop).  Skipped where fewer than two devices are visible (the GPU box
    of this build has one; an 8-GPU node runs it)."""
    import json

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    for n in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--samples", "200000",
                              "--no-cpu-baseline", "--sustain-seconds", "0"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines[n] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    two = lines[2]
    assert two["n_gpus"] == 2 and two["ranks_seen_by_process_group"] == 2 and two["config"]["backend"] == "nccl"
    assert two["config"]["engine"].startswith("flobaroid_amd._lib.Engine")
    # the same 200 k samples whatever the world size: the all-reduced Gram agrees with the one-GPU Gram
    for k in ("trace", "fro"):
        assert abs(two["gram_checksum"][k] / lines[1]["gram_checksum"][k] - 1) <= 1e-12
    assert two["tsqr"]["rank_tree_levels"] == 1 and two["tsqr"]["relerr_RtR_vs_allreduced_gram"] <= 1e-11
    assert two["weak_scaling"]["relerr_vs_world_x_sharded_gram"] <= 1e-11
