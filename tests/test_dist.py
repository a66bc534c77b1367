"""world_size-2 gloo tests of the multi-GPU plumbing (sample sharding, Gram all-reduce, TSQR tree)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

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
from flobaroid_amd.dist import shard_range, allreduce_gram, tsqr_tree, warm_p2p, _tree_edges, selfcheck, Watchdog
from common import load_topo, random_states
from oracle.oracle import OracleModel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
warm_p2p()   # one message over every edge of the rank tree + all-reduce + broadcast: must neither hang nor mismatch
chk = selfcheck()   # every exchange step on data each rank can check alone
assert chk["world"] == world and chk["tsqr_tree_relerr"] <= 1e-12
if os.environ.get("FBR_TEST_BAD_MERGE"):   # a merge that returns a wrong factor must be reported by name, not survive
    try:
        selfcheck(merge=lambda a, b: a)
        print("BAD MERGE NOT DETECTED"); sys.exit(5)
    except RuntimeError as e:
        assert "tsqr_tree" in str(e) and "self-check failed" in str(e), str(e)
if os.environ.get("FBR_TEST_BAD_EDGE"):   # a payload corrupted on a tree edge: the receiver names the edge, every other rank learns that a peer failed
    import flobaroid_amd.dist as D
    orig = D._edge_payload
    if rank == 1:   # (rank 1 sends to rank 0 on the first level; only the sender's copy is wrong)
        D._edge_payload = lambda r, s, n=4096: orig(r, s, n) + (torch.arange(n) == 7).to(torch.float64)
    try:
        selfcheck(timeout=60.0)
        print("BAD EDGE NOT DETECTED"); sys.exit(6)
    except RuntimeError as e:
        if rank == 0:
            assert "tree edge 1 -> 0 differs from what rank 1 sends in 1 of 4096" in str(e), str(e)
        else:
            assert "other rank(s) reported a failure in step 'send/recv over the tree edges'" in str(e), str(e)
    D._edge_payload = orig
# packed triangles (the default) and dense squares give the same factor; without the broadcast only the root has it
Rt = torch.triu(torch.randn((9, 9), dtype=torch.float64, generator=torch.Generator().manual_seed(3 + rank)))
qr = lambda a, b: torch.from_numpy(np.linalg.qr(np.vstack([a.numpy(), b.numpy()]), mode="r"))
Rp, Rd = tsqr_tree(Rt, qr), tsqr_tree(Rt, qr, packed=False)
assert torch.equal(Rp, Rd) and bool((torch.tril(Rp, -1) == 0).all())
Rn = tsqr_tree(Rt, qr, broadcast=False)
assert (Rn is not None and torch.equal(Rn, Rp)) if rank == 0 else Rn is None
assert len(_tree_edges(world)) == world - 1 and sorted(s for _, s in _tree_edges(world)) == list(range(1, world))
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
# sub-group (1, 2) of a 3-rank world: group-local ranks 0, 1 are global ranks 1, 2 (send / recv / broadcast take global ranks)
if world == 3:
    sub = dist.new_group([1, 2])
    if rank in (1, 2):
        warm_p2p(group=sub)
        Ab = A[om.rows * (0 if rank == 1 else 150):om.rows * (150 if rank == 1 else S)]
        Rg = tsqr_tree(torch.from_numpy(np.linalg.qr(Ab, mode="r")), merge, group=sub)
        assert np.linalg.norm(Rg.numpy().T @ Rg.numpy() - A.T @ A) <= 1e-12 * np.linalg.norm(A.T @ A)
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
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + world), FBR_TEST_BAD_MERGE="1", FBR_TEST_BAD_EDGE="1")
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=240)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        procs = []


def test_watchdog_names_the_step_and_ends_the_process(tmp_path):
    """A distributed step that never finishes: the watchdog prints the step and the hints and ends the process with code 3 (here a rank
    waiting for a message nobody sends)."""
    script = tmp_path / "stuck.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        from flobaroid_amd.dist import Watchdog
        with Watchdog(1.0, "recv from rank 7 that never sends", rank=0):
            time.sleep(30)
        print("not reached")
    """ % ROOT))
    r = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 3 and b"recv from rank 7 that never sends" in r.stderr and b"NCCL_DEBUG" in r.stderr and b"not reached" not in r.stdout


def _run_bench(extra, env_extra=None, timeout=600):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]))
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    if env_extra:
        env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--engine", "cpu_engine:make_engine", "--samples", "192",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0"] + extra
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_bench_gpus_2_spawns_two_ranks_end_to_end():
    """``python bench.py --gpus 2`` starts its two ranks itself (re-exec under torch.distributed.run), shards the samples,
    all-reduces the Gram, runs the TSQR rank tree with its in-bench check and the weak-scaling leg, and prints ONE JSON line
    whose n_gpus is the process group's size.  gloo + the CPU stand-in engine: the launch logic is the product's, the arithmetic
    is the oracle's.  The one-rank run must reduce the same 192 samples to the same Gram."""
    import json

    lines = {}
    for n in (1, 2):
        r = _run_bench(["--gpus", str(n)])
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        js = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        assert len(js) == 1, r.stdout.decode()
        lines[n] = json.loads(js[0])
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and two["ranks_seen_by_process_group"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == "strong" and two["config"]["samples_per_gpu"] == 96 and two["config"]["samples_per_step"] == 192
    for key in ("trace", "fro"):
        assert abs(one["gram_checksum"][key] - two["gram_checksum"][key]) <= 1e-11 * abs(one["gram_checksum"][key])
    assert two["tsqr"]["rank_tree_levels"] == 1 and two["tsqr"]["relerr_RtR_vs_allreduced_gram"] <= 1e-11
    assert two["weak_scaling"]["samples_per_gpu"] == 192 and two["weak_scaling"]["relerr_vs_world_x_sharded_gram"] <= 1e-11
    assert two["selfcheck_dist"]["world"] == 2 and two["selfcheck_dist"]["tsqr_tree_relerr"] <= 1e-12 and "selfcheck_dist" not in one
    assert len(two["per_rank_ms_per_step"]) == 2 and len(two["per_rank_allreduce_wait_ms_per_step"]) == 2 and len(one["per_rank_ms_per_step"]) == 1
    r = _run_bench(["--gpus", "2", "--selfcheck-dist"])   # the self-check alone
    assert r.returncode == 0 and b'"selfcheck_dist": "ok"' in r.stdout, r.stderr.decode()[-2000:]
    assert "weak_scaling" not in one and one["tsqr"]["rank_tree_levels"] == 0


def test_bench_refuses_a_mislabelled_world():
    """Launched by hand with fewer ranks than --gpus says, the bench fails instead of printing n_gpus: N for one rank."""
    r = _run_bench(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert r.returncode != 0 and b"refusing" in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


@pytest.mark.timeout(1800)   # (8 ranks on the build container's cores: the subprocess has its own 900 s limit, below pytest's)
def test_bench_gpus_8_dry_run_over_gloo():
    """The world size the driver's scaling run uses, before the first real node sees it: ``bench.py --gpus 8`` over gloo with the CPU
    stand-in -- eight ranks spawned by the bench itself, the self-check over every one of the 7 edges of the depth-3 rank tree,
    ``warm_p2p``, the watchdog armed around every exchange step, the sharded Gram + all-reduce, the TSQR rank tree (3 levels) checked
    against the all-reduced Gram, and the weak-scaling leg.  Same checksums as the one-rank run of the same 512 samples."""
    import json

    r8 = _run_bench(["--gpus", "8", "--samples", "512", "--dist-timeout", "240"], timeout=900)
    assert r8.returncode == 0, r8.stderr.decode()[-3000:]
    js = [l for l in r8.stdout.decode().splitlines() if l.startswith("{")]
    assert len(js) == 1
    eight = json.loads(js[0])
    r1 = _run_bench(["--gpus", "1", "--samples", "512"])
    one = json.loads([l for l in r1.stdout.decode().splitlines() if l.startswith("{")][0])
    assert eight["n_gpus"] == 8 and eight["ranks_seen_by_process_group"] == 8 and eight["config"]["samples_per_gpu"] == 64
    for key in ("trace", "fro"):
        assert abs(one["gram_checksum"][key] - eight["gram_checksum"][key]) <= 1e-11 * abs(one["gram_checksum"][key])
    assert eight["tsqr"]["rank_tree_levels"] == 3 and eight["tsqr"]["relerr_RtR_vs_allreduced_gram"] <= 1e-11
    sc = eight["selfcheck_dist"]
    assert sc["world"] == 8 and sc["tsqr_tree_relerr"] <= 1e-12 and sc.get("tree_edges", 7) == 7
    assert len(eight["per_rank_ms_per_step"]) == 8 and eight["weak_scaling"]["relerr_vs_world_x_sharded_gram"] <= 1e-11
