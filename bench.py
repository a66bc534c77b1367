#!/usr/bin/env python3
"""bench.py -- regressor samples/s + TSQR TFLOP/s of the fused hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W

starts N ranks ITSELF (re-executes under ``python -m torch.distributed.run --nproc-per-node N``; a launch that already
runs under torchrun is used as is) and fails loudly when fewer than N devices / ranks come up.  One process per GPU,
RCCL (backend "nccl") over xGMI.

A step = one pass of the hot path over BASELINE.json configs[3] -- WALK-MAN floating base, 1 M synthetic samples SHARDED
over the N ranks (125 k per GPU at 8), inputs resident in HBM: link kinematics + fused regressor -> [Y|tau]^T[Y|tau] Gram
(fp64 MFMA), reduced deterministically on the device, then ONE RCCL all-reduce of the (P+1)^2 fp64 Gram.  ``value`` =
1 M / (max-over-ranks seconds per step): strong scaling, identical to the one-GPU workload at N = 1.  The same JSON line
carries the weak-scaling figure (1 M samples PER GPU), the Householder-TSQR legs (per-rank ``fbr_tsqr`` + binary tree of
``fbr_tsqr_merge`` over the ranks, checked against the all-reduced Gram inside the bench), and at N = 1 the other
BASELINE configs, the PCIe-inclusive rate, the roofline of the dominant kernel and the CPU baselines.

The global data set is made of 8 seeded blocks (seeds 42..49), so every world size sees the same 1 M samples and
``gram_checksum`` must agree across the N = 1, 2, 4, 8 lines.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

# Worker threads of OpenMP / OpenBLAS teams (the CPU baseline legs, NumPy / SciPy in the parity legs) spin after a parallel region by
# default.  On a box whose cgroup grants 16 CPUs they eat the quota the thread that drives the GPU needs: blocking calls of the later legs
# then measured 2 ... 13 ms for 2 ms of device work, run to run.  Passive waiting, set before the runtimes load; a caller's own
# settings win.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # vendor dense fp64 matrix peak of MI355X; not in MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
MFMA_FLOP = 2 * 16 * 16 * 4   # one v_mfma_f64_16x16x4_f64
NUM_BLOCKS = 8                # seeded blocks of the global data set


# ----------------------------------------------------------------------------------------------------------------------
# launch
# ----------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=1_000_000, help="samples of the whole job per step (sharded over the ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the timed steps (profiling runs)")
    ap.add_argument("--sustain-seconds", type=float, default=10.0, help="N = 1: length of the sustained leg after the timed steps")
    ap.add_argument("--selfcheck-dist", action="store_true",
                    help="N > 1: only the self-check of the exchange steps (it also runs at the start of every --gpus N > 1 run)")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds a distributed step may take before the watchdog ends the rank with a message")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: CPU tensors; only meaningful with --engine (the test-suite's stand-in), the product has no CPU path")
    ap.add_argument("--engine", default=None,
                    help="module:factory returning an Engine-like object (tests inject a CPU stand-in to exercise the launch / "
                         "sharding / reduction logic without a GPU); default: flobaroid_amd._lib.Engine (HIP, fails without a device)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="option of every model handle the bench creates (fbr_model_set_option, include/fbr.h); experiments only -- the line records them")
    args = ap.parse_args(argv)
    if args.opt:
        from flobaroid_amd import _lib

        for kv in args.opt:
            key, _, val = kv.partition("=")
            _lib.DEFAULT_OPTIONS[key.strip()] = float(val)
    return args


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args) -> None:
    """``--gpus N`` without a torchrun environment: start the N ranks here and exit with their status."""
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "RANK" in os.environ or args.gpus == 1:
        return
    if args.backend == "nccl":
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but only {have} HIP device(s) are visible; refusing to run fewer ranks")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


# ----------------------------------------------------------------------------------------------------------------------
# synthetic inputs (the distributions of model.py:696-725, tau = ID(state; xStdModel) + N(0, 0.05^2))
# ----------------------------------------------------------------------------------------------------------------------
def synth_block(topo, S, seed, dev, floating=True):
    import torch

    g = torch.Generator(device=dev).manual_seed(int(seed))

    def rand(*shape):
        return torch.rand(shape, dtype=torch.float64, device=dev, generator=g)

    n = topo.num_dofs
    lo = torch.tensor([topo.limits[j]["lower"] for j in topo.dof_names], dtype=torch.float64, device=dev)
    hi = torch.tensor([topo.limits[j]["upper"] for j in topo.dof_names], dtype=torch.float64, device=dev)
    vm = torch.tensor([topo.limits[j]["velocity"] for j in topo.dof_names], dtype=torch.float64, device=dev)
    st = dict(q=lo + (hi - lo) * rand(S, n), dq=(rand(S, n) - 0.5) * 2 * vm, ddq=(rand(S, n) - 0.5) * 2 * np.pi)
    if floating:
        st.update(base_vel=np.pi * rand(S, 6), base_acc=np.pi * rand(S, 6), rpy=0.1 * rand(S, 3))
    st["_noise"] = 0.05 * torch.randn((S, n + (6 if floating else 0)), dtype=torch.float64, device=dev, generator=g)
    return st


def synth_range(topo, total, a, b, dev, floating=True, seed0=42):
    """Samples [a, b) of the global data set of ``total`` samples = NUM_BLOCKS seeded blocks (seed0 + block)."""
    import torch

    bounds = [(total * i) // NUM_BLOCKS for i in range(NUM_BLOCKS + 1)]
    parts = []
    for i in range(NUM_BLOCKS):
        lo, hi = max(a, bounds[i]), min(b, bounds[i + 1])
        if lo >= hi:
            continue
        blk = synth_block(topo, bounds[i + 1] - bounds[i], seed0 + i, dev, floating)
        parts.append({k: v[lo - bounds[i]:hi - bounds[i]] for k, v in blk.items()})
    if not parts:
        parts = [{k: v[:0] for k, v in synth_block(topo, 1, seed0, dev, floating).items()}]
    return {k: torch.cat([p[k] for p in parts]).contiguous() for k in parts[0]}


def with_tau(eng, topo, st):
    """states dict (without the noise entry) and rhs = tau (S*rows, 1) on the device."""
    noise = st.pop("_noise")
    tau = eng.inverse_dynamics(st, topo.x_std())
    tau = tau + noise
    return st, tau.reshape(-1, 1).contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# CPU baselines (oracle = test infrastructure; used here only as the timed / checking CPU leg)
# ----------------------------------------------------------------------------------------------------------------------
def _np_states(topo, S, seed, floating=True):
    rng = np.random.default_rng(seed)
    n = topo.num_dofs
    lo = np.array([topo.limits[j]["lower"] for j in topo.dof_names])
    hi = np.array([topo.limits[j]["upper"] for j in topo.dof_names])
    vm = np.array([topo.limits[j]["velocity"] for j in topo.dof_names])
    st = dict(q=lo + (hi - lo) * rng.random((S, n)), dq=(rng.random((S, n)) - 0.5) * 2 * vm, ddq=(rng.random((S, n)) - 0.5) * 2 * np.pi)
    if floating:
        st.update(base_vel=np.pi * rng.random((S, 6)), base_acc=np.pi * rng.random((S, 6)), rpy=0.1 * rng.random((S, 3)))
    return st


def synth_states(topo, S, seed, floating=True):
    """NumPy states + the generator (tools/*_probe.py)."""
    return _np_states(topo, S, seed, floating), np.random.default_rng(seed + 1)


def cpu_baseline(topo, budget_s=12.0):
    """Phase A of BASELINE.md §3 on ONE host thread: oracle (C restatement) per-sample regressor + RNEA torques + the A^T A
    accumulation of the stacked block with NumPy (BLAS pinned to 1 thread)."""
    from oracle.oracle import OracleModel
    from threadpoolctl import threadpool_limits

    om = OracleModel(topo, floating=True)
    x = topo.x_std()
    block = 256
    st = _np_states(topo, block, 1234)
    done = 0
    G = np.zeros((om.P + 1, om.P + 1))
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        while True:
            Y = om.regressor(st)
            tau = om.inverse_dynamics(st, x).reshape(-1, 1)
            Ya = np.hstack([Y, tau])
            G += Ya.T @ Ya
            done += block
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{done} WALK-MAN floating-base samples: oracle regressor + RNEA (C, 1 thread) + NumPy A^T A (1 BLAS thread), {dt:.1f} s"}


def cpu_baseline_all_cores(topo, one_thread_rate=None, budget_s=10.0):
    """The same port on every host core, ONE process: the reference's loop body (regressor + simulated torques of a sample) AND the A^T A
    accumulation dealt to OpenMP threads inside the C oracle (``orc_stack_gram_omp``: per-thread upper-triangular sums of rank-1 updates,
    added at the end; nothing tall is stored).  SURVEY 8(d): "1 thread, then all host cores with OpenMP".  NumPy's BLAS is no way to
    use the cores here: OpenBLAS serialises level-3 calls that come from several threads (why round 5's thread pool scaled 10.7 x on 256
    threads), and ONE threaded dsyrk of a 482-column block parallelises over the columns only (measured in round 6: 89 GFLOP/s on 256
    threads).  Reports the rate, the one-thread rate of the SAME function and the parallel efficiency between the two."""
    from oracle.oracle import OracleModel

    # what this process may actually use: the affinity mask and the cgroup CPU quota of the box, not the socket's thread count
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except Exception:
            pass
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    om = OracleModel(topo, floating=True)
    x = topo.x_std()
    st1 = _np_states(topo, 512, 4321)
    om.stack_gram(st1, x, threads=1)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 2.0:
        om.stack_gram(st1, x, threads=1)
        n1 += 512
    rate1 = n1 / (time.perf_counter() - t0)
    block = max(512, 64 * cores)
    st = _np_states(topo, block, 4322)
    # thread-count sweep (one block each): the rate of a team is bounded by the cores the box really schedules, whatever it reports
    sweep = {}
    for th in sorted({max(1, cores // 16), max(1, cores // 4), max(1, cores // 2), cores}):
        om.stack_gram(_np_states(topo, 8 * th, 1), x, threads=th)
        t0 = time.perf_counter()
        om.stack_gram(st, x, threads=th)
        sweep[th] = block / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    cores_detected, cores = cores, best
    _, thr = om.stack_gram(st, x, threads=cores)  # (starts the OpenMP team)
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        _, thr = om.stack_gram(st, x, threads=cores)
        done += block
    dt = time.perf_counter() - t0
    rate = done / dt
    out = {"value": rate, "unit": "samples/s", "cores": cores, "omp_threads": thr, "kind": "port", "seconds": dt,
           "one_thread_same_function_samples_per_s": rate1, "speedup_vs_one_thread_same_function": rate / rate1,
           "parallel_efficiency_per_thread": rate / rate1 / max(thr, 1),
           "sample": f"{done} WALK-MAN floating-base samples in blocks of {block}: C oracle regressor + RNEA + rank-1 A^T A accumulation (structural zeros "
                     f"skipped) over {thr} OpenMP threads, per-thread sums added at the end, {dt:.1f} s",
           "threads_available": cores_detected, "cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota,
           "thread_sweep_samples_per_s": {str(k): v for k, v in sweep.items()},
           "note": "the timed loop uses the thread count of the sweep with the best rate; hardware threads are SMT siblings: an efficiency of 0.5 per "
                   "thread is one per core"}
    if one_thread_rate:
        out["speedup_vs_cpu_baseline_one_thread_blas"] = rate / one_thread_rate
    return out


def cpu_phases_bc_and_parity(eng, topo, Sc=4000):
    """Phases B and C of BASELINE.md §3 -- the reference's ACTUAL dense-LA calls through SciPy / NumPy on the host (all BLAS
    threads) -- and, from the same inputs, the end-to-end parity figure of SURVEY 8(d): standard parameters identified from the
    GPU reductions (structural Gram -> pivoted QR -> independent columns; TSQR factor -> lstsq -> pinv(K)) against the CPU path
    (oracle-materialised YStd, numpy.linalg.lstsq on the tall YBase), relative Frobenius error.

    B: scipy.linalg.qr(pivoting=True, mode="economic") on the structural Gram (model.py:809) -> P, rank, K (model.py:871-894).
    C: numpy.linalg.lstsq / pinv / qr on the materialised YBase (identifier.py:709-712, sdp.py:470) at Sc samples."""
    import numpy.linalg as la
    import scipy
    import scipy.linalg as sla

    from flobaroid_amd import estimation as est
    from oracle.oracle import OracleModel

    def np_(x):
        return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

    out = {"numpy": np.__version__, "scipy": scipy.__version__, "host_threads": os.cpu_count()}
    # structural Gram on the GPU from the reference's random-state distributions (model.py:696-725), randomSamples = 10000
    R_struct = np_(eng.gram(_np_states(topo, 10000, 99)))
    t0 = time.perf_counter()
    Q, RQ, PQ = sla.qr(R_struct, pivoting=True, mode="economic")
    tB = time.perf_counter() - t0
    P = R_struct.shape[0]
    r = int(np.count_nonzero(np.abs(np.diag(RQ)) > 0.005))  # minTol of configs/walkman_full.yaml
    ic = PQ[:r]
    Pp = np.zeros((P, P))
    Pp[np.arange(P), PQ] = 1.0  # Pp[i, P[i]] = 1 (model.py:876-878)
    deps = sla.solve_triangular(RQ[:r, :r], RQ[:r, r:])
    deps[np.abs(deps) < 0.005] = 0.0
    K = Pp.T[:, :r].T + deps @ Pp.T[:, r:].T
    out["phase_B"] = {"call": "scipy.linalg.qr(R, pivoting=True, mode='economic')", "matrix": [P, P], "seconds": tB,
                      "GFLOP_per_s": (4.0 / 3.0) * P ** 3 / tB / 1e9, "base_rank": r}
    # phase C on the tall matrix
    om = OracleModel(topo, floating=True)
    st = _np_states(topo, Sc, 77)
    x_true = topo.x_std()
    rng = np.random.default_rng(5)
    Y = om.regressor(st)
    tau = om.inverse_dynamics(st, x_true).reshape(-1) + 0.05 * rng.standard_normal(Sc * om.rows)
    YB = np.ascontiguousarray(Y[:, ic])
    M, nb = YB.shape
    fl = 2.0 * M * nb * nb - (2.0 / 3.0) * nb ** 3
    res = {}
    t0 = time.perf_counter()
    xb_cpu = la.lstsq(YB, tau, rcond=None)[0]
    res["lstsq"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    la.pinv(YB)
    res["pinv"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    la.qr(YB)
    res["qr"] = time.perf_counter() - t0
    out["phase_C"] = {"samples": Sc, "matrix": [M, nb], "seconds": res,
                      "qr_GFLOP_per_s": fl / res["qr"] / 1e9, "flop_model": "2 M P^2 - 2/3 P^3",
                      "note": "costs are linear in the sample count; 1 M samples = x%d" % (1_000_000 // Sc)}
    xstd_cpu = la.pinv(K) @ xb_cpu
    # the same through the GPU reductions
    R_aug = np_(eng.tsqr(st, rhs=tau.reshape(-1, 1)))
    xb_gpu, _, _ = est.identify_base_parameters(R_aug, ic, P, M, add_contacts=False)
    xstd_gpu = est.find_std_from_base(K, xb_gpu)
    # index-set audit (model.py:871-884): pivots of the product's rule (ties -> lowest column index, model.pivoted_qr) against
    # LAPACK's own order on the same Gram bits -- how many of the reference's pivots are coin flips between exactly tied columns,
    # and how many columns of the independent SET that moves (a tie between a link and a link welded to it straddles the rank)
    from flobaroid_amd.model import pivoted_qr

    Rr, Pr = pivoted_qr(R_struct)[1:]
    dr, dl = np.abs(np.diag(Rr)), np.abs(np.diag(RQ))
    differ = np.flatnonzero(Pr[:r] != PQ[:r])
    # the same audit with the tie rule OFF (opt['pivotTieTolerance'] = 0: LAPACK's own last-bit tie breaking) on the GPU Gram, against
    # scipy.linalg.qr on the CPU oracle's Gram of the same states: what differs THEN is the summation order of two implementations of the
    # same sum (coin flips no implementation can reproduce); what the rule adds on top is the difference to the figure above
    st_s = _np_states(topo, 10000, 99)
    om_s = OracleModel(topo, floating=True)
    G_cpu = np.zeros_like(R_struct)
    for i0 in range(0, 10000, 500):   # the reference's raw sum over the samples (model.py:803-806), block by block
        Yb = om_s.regressor({k_: v[i0:i0 + 500] for k_, v in st_s.items()})
        G_cpu += Yb.T @ Yb
    P0 = pivoted_qr(R_struct, tie_eps=0.0)[2]
    Pc = sla.qr(G_cpu, pivoting=True, mode="economic")[2]
    Rc = sla.qr(G_cpu, pivoting=True, mode="economic")[1]
    rc_ = int(np.count_nonzero(np.abs(np.diag(Rc)) > 0.005))
    set0, setc, setr = set(P0[:r].tolist()), set(Pc[:rc_].tolist()), set(Pr[:r].tolist())
    try:
        import cvxpy  # noqa: F401
        sdp_state = ("cvxpy importable here: `python tools/pin_sdp.py --robot walkman --samples 20000 --write` runs the reference's own "
                     "SDP.identifyFeasibleStandardParameters on the CPU path and on this path's TSQR factor and checks xStd at 1e-6")
    except Exception:
        sdp_state = ("NOT executed with a conic solver: cvxpy / CLARABEL are not installed on the build or the GPU image (no network).  tools/pin_sdp.py "
                     "runs the reference's own, unmodified SDP.identifyFeasibleStandardParameters (sdp.py:450-604) twice -- its own la.qr(YBase) on the CPU "
                     "path, the TSQR factor of this path through estimation.sdp_inputs -- and compares xStd at 1e-6 wherever cvxpy exists; here its plumbing "
                     "runs with tests/stub_cvxpy.py (affine expressions exact, LMIs ignored): tests/test_pin_sdp.py, 2e-13 (KUKA) / 1e-14 (WALK-MAN) "
                     "between the two routes; its inputs R1, rho1, rho2_norm_sqr, R1 K, observability weights, CAD rows: tests/test_gpu_endtoend.py")
    out["parity"] = {"sdp_solve": sdp_state, "samples": Sc, "xstd_rel_fro_err_gpu_vs_cpu": float(la.norm(xstd_gpu - xstd_cpu) / la.norm(xstd_cpu)),
                     "xbase_rel_err": float(la.norm(xb_gpu - xb_cpu) / la.norm(xb_cpu)), "bar": 1e-6, "num_base_params": r,
                     "pivot_rule_vs_lapack": {"rank_rule": int(np.count_nonzero(dr > 0.005)), "positions_differing": int(len(differ)),
                                              "largest_rel_gap_at_those": float(max([abs(dr[i] - dl[i]) / dl[i] for i in differ], default=0.0)),
                                              "index_set_columns_differing": int(len(set(Pr[:r].tolist()) - set(PQ[:r].tolist()))),
                                              "index_set_equal_with_pivotTieTolerance_0": bool(set0 == setc),
                                              "with_pivotTieTolerance_0": {
                                                  "gpu_gram_vs_cpu_oracle_gram_lapack_order_columns_differing": int(len(set0 - setc)),
                                                  "gram_rel_diff_gpu_vs_cpu": float(la.norm(R_struct - G_cpu) / la.norm(G_cpu)),
                                                  "tie_rule_on_gpu_gram_vs_cpu_gram_columns_differing": int(len(setr - set(pivoted_qr(G_cpu)[2][:rc_].tolist()))),
                                                  "meaning": "rule off: columns by which two implementations of the same Gram sum (GPU reduction order vs the CPU "
                                                             "oracle's sample loop) disagree through LAPACK's last-bit tie breaking -- coin flips between exactly "
                                                             "tied columns that no implementation reproduces; rule on (the default): the same comparison is exact "
                                                             "(0 columns), which is what the rule is for"},
                                              "note": "every differing position is an exact tie (gap at rounding level); "
                                                      "tests/test_gpu_model.py rrW audits the same against the reference run's own order"}}
    return out, ic, K


# ----------------------------------------------------------------------------------------------------------------------
# rank body
# ----------------------------------------------------------------------------------------------------------------------
def make_engine(args, topo, floating, device, **kw):
    if args.engine:
        mod, fn = args.engine.split(":")
        return getattr(importlib.import_module(mod), fn)(topo, floating=floating, device=device, **kw)
    from flobaroid_amd._lib import Engine

    return Engine(topo, floating=floating, device=device, **kw)


def run_rank(args) -> int:
    import torch
    import torch.distributed as dist

    from flobaroid_amd.dist import Watchdog, selfcheck, shard_range, tsqr_tree, warm_p2p
    from flobaroid_amd.topology import Topology

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); refusing to mislabel the run", file=sys.stderr)
        return 2
    on_gpu = args.backend == "nccl"
    if on_gpu:
        if torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} has no HIP device {local}", file=sys.stderr)
            return 2
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise RuntimeError("process group came up with the wrong size")
        # communicators of the all-reduce and of every edge of the TSQR rank tree exist before anything is timed; a step that does not
        # finish ends the rank with a message naming it (flobaroid_amd/dist.py: Watchdog) instead of hanging the job
        with Watchdog(args.dist_timeout, "warm_p2p (creating the RCCL communicators: one per tree edge + the collective one)", rank):
            warm_p2p(dev)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    def max_over_ranks(x: float) -> float:
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", "walkman_apriori.topology.json"))
    eng = make_engine(args, topo, True, local)
    if on_gpu:
        eng.use_torch_stream()
    check = None
    if use_dist and world > 1:
        # every exchange step once, on data whose result each rank can compute alone: a wrong transport, a missing peer or a bad merge
        # is reported here with the step's name, not as a hang or a wrong number ten minutes into the run
        check = selfcheck(dev, eng.tsqr_merge, timeout=args.dist_timeout)
        if args.selfcheck_dist:
            if rank == 0:
                print(json.dumps({"selfcheck_dist": "ok", **check}))
            dist.destroy_process_group()
            return 0
    elif args.selfcheck_dist:
        print(json.dumps({"selfcheck_dist": "nothing to check with one rank"}))
        return 0
    rows, P = eng.rows, eng.cols
    Pa = P + 1
    S_total = args.samples
    a, b = shard_range(S_total, rank, world)
    S = b - a
    st, rhs = with_tau(eng, topo, synth_range(topo, S_total, a, b, dev))
    G = torch.zeros((Pa, Pa), dtype=torch.float64, device=dev)

    # Steps are SUBMITTED asynchronously where the engine offers it (fbr_gram_submit, device-resident data): two steps in flight, so
    # that the kinematics / packing of step i+1's first chunk run beside the last Gram launches of step i, and the all-reduce of step i
    # (async_op) beside the kernels of step i+1.  Every step still produces its own complete, all-reduced Gram inside the timed region.
    pipelined = on_gpu and hasattr(eng, "gram_submit") and not os.environ.get("FBR_BENCH_BLOCKING")
    Gs = [G, torch.zeros_like(G)]
    state = {"pending": None, "works": [None, None], "i": 0, "ar_wait": 0.0}

    def wait_work(b):  # (the time a rank spends here is the all-reduce it could not hide: reported per rank)
        t_ = time.perf_counter()
        state["works"][b].wait()
        state["ar_wait"] += time.perf_counter() - t_
        state["works"][b] = None

    def finish(p):
        tkt, b = p
        eng.wait(tkt)
        if use_dist:
            state["works"][b] = dist.all_reduce(Gs[b], async_op=True)  # (P+1)^2 fp64 = 1.86 MB: the only exchange step of the pass

    def step():
        if not pipelined:
            eng.gram(st, rhs=rhs, out=Gs[0])
            if use_dist:
                t_ = time.perf_counter()
                dist.all_reduce(Gs[0])
                sync()
                state["ar_wait"] += time.perf_counter() - t_
            state["last"] = 0
            return
        b = state["i"] & 1
        state["i"] += 1
        if state["works"][b] is not None:  # the all-reduce that last used this buffer
            wait_work(b)
        tkt = eng.gram_submit(st, Gs[b], rhs=rhs)
        if state["pending"] is not None:
            finish(state["pending"])
        state["pending"] = (tkt, b)
        state["last"] = b

    def drain():
        if state["pending"] is not None:
            finish(state["pending"])
            state["pending"] = None
        for b in (0, 1):
            if state["works"][b] is not None:
                wait_work(b)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    # ---- the timed steps of the contract
    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    eng.profile_enable(True)
    eng.profile_get()
    state["ar_wait"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    sync()
    dt_rank = time.perf_counter() - t0   # this rank's own time up to its last result (before the closing barrier)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    per_rank = [[dt_rank / args.steps * 1e3, state["ar_wait"] / args.steps * 1e3]]
    if use_dist:
        tt = torch.tensor(per_rank[0], dtype=torch.float64, device=dev)
        gl = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gl, tt)
        per_rank = [g_.cpu().tolist() for g_ in gl]
    prof = eng.profile_get()
    eng.profile_enable(False)
    G_sharded = Gs[state.get("last", 0)].clone()

    ms_per_step = dt / args.steps * 1e3
    value = S_total / (dt / args.steps)
    info = eng.gram_program_info(1, S) if on_gpu else eng.gram_program_info(1)  # (the program a pass over S samples executes)
    # the pass over sample-contiguous images (option gram_lane, csrc/fbr_gram64.h) runs its own tile pairs: its MFMA count per 64-sample block
    lane = eng.gram_lane_info(1, S) if on_gpu and hasattr(eng, "gram_lane_info") else {"active": False}
    if lane.get("active"):
        info = dict(info, mfma_per_sample=lane["mfma_per_block"] / 64.0)
    gram_ms, gram_n = prof["gram"]
    samples_per_launch = S * args.steps / max(gram_n, 1)
    alg_flop_per_sample = rows * P * (P + 1) + 2 * rows * P * 1  # symmetric Gram count + 1 rhs column
    avg_launch_s = gram_ms / max(gram_n, 1) * 1e-3
    executed_flop_per_sample = info["mfma_per_sample"] * MFMA_FLOP
    executed = executed_flop_per_sample * samples_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
    out = {
        "metric": "regressor samples/s (fused regressor->Gram pass) + TSQR TFLOP/s, WALK-MAN 31-DOF float-base",
        "value": value,
        "unit": "samples/s",
        "n_gpus": world,
        "ranks_seen_by_process_group": dist.get_world_size() if use_dist else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: walkman_apriori floating base (29 DOF, 48 links, {rows}x{P} regressor block), "
                        f"{S_total} samples per step sharded over {world} GPU(s) ({S} on rank 0), fused kinematics + "
                        "[Y|tau]^T[Y|tau] Gram" + (" + RCCL all-reduce of the Gram" if world > 1 else ""),
            "samples_per_step": S_total,
            "samples_per_gpu": S,
            "rhs_columns": 1,
            "parallelism": f"samples sharded over {world} GPU(s), one all-reduce of the (P+1)^2 fp64 Gram per step",
            "submission": "asynchronous, two steps in flight (fbr_gram_submit / fbr_wait)" if pipelined else "blocking calls",
            "backend": args.backend,
            # which engine ran the steps: the HIP library unless a test injected a stand-in with --engine (then this names it)
            "engine": f"{type(eng).__module__}.{type(eng).__qualname__}" + (f" (injected with --engine {args.engine})" if args.engine else ""),
            "serialisation": "links / DOFs in iDynTree traversal order (tests/golden/reference_joint_orders.json)",
            # the columns the reductions run on (fbr.h fbr_model_link_merge_info): fixed links merged into the bodies they ride on,
            # the three joint-invariant parameters of every link behind a revolute joint regrouped into the parent body -- exact,
            # constant column dependencies; the (P+1)^2 Gram of ALL columns is expanded from the reduced one inside the timed step
            "reduction": (lambda li: f"{li['cols']} columns reduced over {li['reduced_cols']} ({li['links']} links, {li['moving_links']} moving bodies); "
                                     "G = E^T G_red E expanded on the device every step; the option link_merge = 0 (fbr_model_set_option) runs all columns")(eng.link_merge_info(S) if on_gpu else eng.link_merge_info())
                         if hasattr(eng, "link_merge_info") else "none",
        },
        # `value` is the rate with the inputs resident in HBM when the timed region starts -- the bench contract of this build's task
        # statement ("if the boundary hands over host buffers, note the PCIe-inclusive rate ... it is never `value`"); ms_per_step is that
        # region.  SURVEY 8(d) words the metric "incl. H2D of states": that rate is `value_incl_h2d` below (N = 1), measured in the same
        # run on the same workload with pinned host inputs staged chunk by chunk on a copy stream (the two differ by run-to-run noise: the
        # 1.1 GB per step cross PCIe behind the kernels).
        "value_resident": value,
        "engine_options": dict(getattr(__import__("flobaroid_amd._lib", fromlist=["DEFAULT_OPTIONS"]), "DEFAULT_OPTIONS", {})),
        "value_definition": "samples per second of the whole job with states and tau resident in HBM (bench contract); value_incl_h2d = the same "
                            "pass fed from pinned host memory (SURVEY 8(d) wording), reported beside it",
        # all world sizes reduce the same 1 M samples: these must agree (to rounding) between the N = 1, 2, 4, 8 lines
        "gram_checksum": {"trace": float(torch.trace(G_sharded).item()), "fro": float(torch.linalg.norm(G_sharded).item())},
        "roofline": {
            "bound": "mfma",
            "kernel": "fbr_gram64_kernel" if lane.get("active") else "fbr_gram_kernel",
            "gram_lane": lane,
            # what the MFMA pipe executed (the kernel skips the structurally zero k-steps of the tile pairs; the PMC pass counts
            # exactly mfma_per_sample x samples): executed flop / launch time measured with HIP events on the launch stream
            "achieved": executed,
            "peak": PEAK_FP64_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": executed / PEAK_FP64_MFMA_TFLOPS,
            "traffic": None,
            "executed_mfma_per_sample": info["mfma_per_sample"],
            "executed_flop_per_sample": executed_flop_per_sample,
            "algorithmic_flop_per_sample": alg_flop_per_sample,
            # SURVEY 8(d)'s dense-symmetric flop count over the same launch time (exceeds what the hardware ran: not a fraction)
            "effective_dense_TFLOP_per_s": alg_flop_per_sample * samples_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0,
            # SURVEY 8(d) contract figure: dense-symmetric algorithmic flop / launch time / peak.  > 1 is possible and expected: the
            # kernel skips the structurally zero k-steps, so this is NOT a hardware utilisation (`frac` is)
            "contract_frac_dense": (alg_flop_per_sample * samples_per_launch / avg_launch_s / 1e12 / PEAK_FP64_MFMA_TFLOPS) if avg_launch_s > 0 else 0.0,
            "contract_frac_dense_note": ">1: structural zeros skipped and, since round 4, linearly dependent columns not recomputed (config.reduction); "
                                        "executed fraction is `frac`",
            "avg_launch_ms": avg_launch_s * 1e3,
            "launches": gram_n,
            "samples_per_launch": samples_per_launch,
            # the same executed flop over the WHOLE STEP (kinematics, packing, reduction and expansion included): what the pass, not its
            # dominant kernel, makes of the MFMA peak
            "step_level": {"executed_TFLOP_per_s": executed_flop_per_sample * S / (ms_per_step * 1e-3) / 1e12 * (S_total / max(S, 1)) / max(world, 1),
                           "frac": executed_flop_per_sample * S / (ms_per_step * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                           "note": "per GPU: executed MFMA flop of this rank's samples / ms_per_step / 78.6 TFLOP/s"},
        },
        "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1]},
        # every rank's own time per step up to its last result, and the part of it spent waiting for the Gram all-reduce it could not
        # hide behind the next step's kernels: a slow rank or a slow link shows here
        "per_rank_ms_per_step": [p_[0] for p_ in per_rank],
        "per_rank_allreduce_wait_ms_per_step": [p_[1] for p_ in per_rank],
    }
    if check is not None:
        out["selfcheck_dist"] = check
    try:  # HBM traffic of the dominant kernel: committed rocprofv3 PMC passes, scaled to this run's samples per launch
        import glob

        src = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gram_pmc_traffic.json")))[-1]
        pm = json.load(open(src))
        if pm.get("gram_kernel_name", "fbr_gram_kernel").startswith(out["roofline"]["kernel"]):
            out["roofline"]["traffic"] = pm["hbm_bytes_per_sample"] * samples_per_launch
            out["roofline"]["traffic_source"] = f"profiles/{os.path.basename(src)} (HBM bytes per sample x samples per launch)"
        else:  # (counters of another kernel: the image of the pass is read once -- its size is the traffic of a launch, to the byte)
            raise KeyError("stale counters")
        # staging traffic of the whole pass (kinematic records written and read, tile images written and streamed): counter bytes per
        # sample of every kernel of the step at this run's rate, against the HBM peak -- the pass's algorithmic I/O is 0.8 KB per sample
        if "staging" in pm:
            tot = pm["staging"]["hbm_bytes_per_sample_all_kernels"]
            out["roofline"]["staging"] = {"hbm_bytes_per_sample": tot, "per_kernel": pm["staging"]["per_kernel_bytes_per_sample"],
                                          "GB_per_s_at_this_rate": tot * S / (ms_per_step * 1e-3) / 1e9,
                                          "frac_of_hbm_peak": tot * S / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                          "algorithmic_io_bytes_per_sample": 8 * (3 * topo.num_dofs + 15 + rows), "source": f"profiles/{os.path.basename(src)}"}
    except Exception:
        pass
    if out["roofline"]["traffic"] is None and lane.get("active"):
        out["roofline"]["traffic"] = lane["block_image_bytes"] / 64.0 * samples_per_launch
        out["roofline"]["traffic_source"] = "fbr_gram_lane_info: bytes of the block images one launch reads (no committed PMC pass of this kernel found)"

    if not args.no_secondary:
        secondary(args, out, eng, topo, st, rhs, G_sharded, dev, world, rank, use_dist, on_gpu, timed, barrier, sync, max_over_ranks)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    return 0


def secondary(args, out, eng, topo, st, rhs, G_sharded, dev, world, rank, use_dist, on_gpu, timed, barrier, sync, max_over_ranks):
    """Everything outside the contract's timed region (all ranks take part in the distributed legs)."""
    import torch
    import torch.distributed as dist

    from flobaroid_amd.dist import shard_range, tsqr_tree

    rows, P = eng.rows, eng.cols
    S_total = args.samples
    S = st["q"].shape[0]
    reps = 3

    # ---- Householder TSQR of the same sharded 1 M samples: per-rank fbr_tsqr, binary tree of fbr_tsqr_merge over the ranks
    def tsqr_sharded():
        R = eng.tsqr(st, rhs=rhs)
        return tsqr_tree(R, eng.tsqr_merge) if use_dist else R

    R = tsqr_sharded()
    RtR = R.T @ R
    err = float((torch.linalg.norm(RtR - G_sharded) / torch.linalg.norm(G_sharded)).item())
    if not err <= 1e-11:
        raise AssertionError(f"TSQR tree: ||R^T R - all-reduced Gram|| / ||Gram|| = {err:.3e} > 1e-11")
    eng.profile_enable(True)
    eng.profile_get()
    dt = timed(tsqr_sharded, reps, 0) / reps
    pr = eng.profile_get()
    eng.profile_enable(False)
    wi = eng.tsqr_work_info(S, k=1)
    merges = 0
    stp = 1
    while stp < world:  # merges on the critical path of the rank tree (one per level)
        merges += 1
        stp *= 2
    wm = eng.tsqr_work_info(0, k=1)  # (no samples: only the shape)
    n_pad, mb = wm["n_padded"], wm["block_rows"]
    NPt = n_pad // 16
    per_update = 8 * (mb // 16) + 4
    merge_mfma = sum(per_update * ((NPt - i0 // 16) * (NPt - i0 // 16 - 1) // 2) for i0 in range(0, n_pad, mb))
    executed_flop = world * wi["flop"] + (world - 1) * merge_mfma * MFMA_FLOP  # all ranks' folds + every merge of the rank tree
    dense_flop = 2.0 * S_total * rows * (P + 1) ** 2
    out["tsqr"] = {
        "workload": f"configs[3] data: {S_total} samples sharded over {world} GPU(s), {rows * S_total} x {P + 1} rows x columns",
        "seconds": dt, "repetitions": reps, "samples_per_s": S_total / dt,
        "executed_TFLOP_per_s": executed_flop / dt / 1e12,
        "executed_frac_of_fp64_mfma_peak": executed_flop / dt / 1e12 / (PEAK_FP64_MFMA_TFLOPS * world),
        "dense_model_TFLOP_per_s": dense_flop / dt / 1e12,
        "flop_models": "executed: 2048 flop x MFMAs the folds run (fbr_tsqr_work_info: the factorisation of the reduced column set -- "
                       "config.reduction --, rows grouped along the kinematic tree, every group factorised over the columns it can touch, "
                       "blocks folded from their first supported column, plus the merge that expands R_red E into the factor of all columns); "
                       "dense: 2*rows*(P+k)^2 per sample (SURVEY 8d)",
        "rank_tree_levels": merges, "relerr_RtR_vs_allreduced_gram": err,
        # device time per class on the call's own stream: `tree` = merge tree of the main row group (the latency-bound tail of a call:
        # 8 levels over 256 private factors, a handful of workgroups each), `tsqr` = the level-0 folds + the embedded group factors
        "kernel_ms_per_call_rank0": {k: v[0] / reps for k, v in pr.items() if v[1]},
        "tree_share_of_call": (pr["tree"][0] / reps) / (dt * 1e3) if pr.get("tree", (0, 0))[1] else None,
    }

    def tsqr_pipelined(stx, rhsx, n_sub):
        """the same calls submitted two in flight (fbr_tsqr_submit): the next call's kinematics / first writer beside the trees"""
        Ro = [torch.zeros((P + 1, P + 1), dtype=torch.float64, device=dev) for _ in range(2)]
        eng.wait(eng.tsqr_submit(stx, Ro[0], rhs=rhsx))
        sync()
        t0 = time.perf_counter()
        pend = None
        for i in range(n_sub):
            tk = eng.tsqr_submit(stx, Ro[i & 1], rhs=rhsx)
            if pend is not None:
                eng.wait(pend)
            pend = tk
        eng.wait(pend)
        sync()
        return (time.perf_counter() - t0) / n_sub, Ro[(n_sub - 1) & 1]

    if world == 1 and on_gpu and hasattr(eng, "tsqr_submit"):
        dtp, Rp = tsqr_pipelined(st, rhs, 4)
        out["tsqr"]["pipelined_seconds"] = dtp
        out["tsqr"]["pipelined_executed_frac_of_fp64_mfma_peak"] = executed_flop / dtp / 1e12 / PEAK_FP64_MFMA_TFLOPS
        out["tsqr"]["pipelined_equals_blocking_bitwise"] = bool(torch.equal(Rp, R))
        # ---- one rank's share of the 1 M samples at 8 GPUs: does the TSQR call scale down to a shard?  (its fixed costs -- the merge
        # trees, the first chunk's producer -- do not shrink with the samples: this leg is the ceiling of the 8-GPU TSQR scaling)
        Ssh = S_total // 8
        sts = {k_: v[:Ssh].contiguous() for k_, v in st.items()}
        rhss = rhs[:Ssh * rows].contiguous()
        eng.tsqr(sts, rhs=rhss)
        eng.profile_enable(True)
        eng.profile_get()
        sync()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.tsqr(sts, rhs=rhss)
        sync()
        dsh = (time.perf_counter() - t0) / 5
        prs = eng.profile_get()
        eng.profile_enable(False)
        dshp, _ = tsqr_pipelined(sts, rhss, 8)
        wsh = eng.tsqr_work_info(Ssh, k=1)
        out["tsqr_shard"] = {
            "samples": Ssh, "seconds": dsh, "pipelined_seconds": dshp, "one_eighth_of_the_1M_call_seconds": dt / 8,
            "ratio_to_one_eighth": dsh / (dt / 8), "pipelined_ratio_to_one_eighth_of_pipelined": dshp / (dtp / 8),
            "scaling_ceiling_at_8_gpus": dt / dsh, "executed_frac_of_fp64_mfma_peak": wsh["flop"] / dsh / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "kernel_ms_per_call": {k_: v[0] / 5 for k_, v in prs.items() if v[1]},
            "tree_share_of_call": (prs["tree"][0] / 5) / (dsh * 1e3) if prs.get("tree", (0, 0))[1] else None,
        }
        # ---- the rank tree that follows the shard at 8 GPUs, timed on this one device: three chained fbr_tsqr_merge (the levels of the
        # binary tree on rank 0's critical path) with the packed-triangle traffic of every hop (pack on the sender, a device copy standing
        # in for the xGMI transfer of 0.93 MB, unpack on the receiver: flobaroid_amd/dist.py tsqr_tree) -- so that the line carries the
        # FULL predicted 8-GPU TSQR time (shard + tree), not only the shard
        from flobaroid_amd.dist import pack_triu, unpack_triu

        Rsh = eng.tsqr(sts, rhs=rhss)
        partners = [Rsh.clone() for _ in range(3)]
        hop = torch.empty(((P + 1) * (P + 2)) // 2, dtype=torch.float64, device=dev)

        def rank_tree_once():
            Rt = Rsh
            for lvl in range(3):
                hop.copy_(pack_triu(partners[lvl]))          # sender: pack; the copy stands in for the link
                Rt = eng.tsqr_merge(Rt, unpack_triu(hop, P + 1))
            return Rt

        Rt = rank_tree_once()
        sync()
        t0 = time.perf_counter()
        for _ in range(10):
            rank_tree_once()
        sync()
        d_tree = (time.perf_counter() - t0) / 10
        sync()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.tsqr_merge(Rsh, partners[0])
        sync()
        d_merge = (time.perf_counter() - t0) / 10
        # four stacked copies of the same factor: R^T R = 4 R_sh^T R_sh
        err_tree = float((torch.linalg.norm(Rt.T @ Rt - 4.0 * (Rsh.T @ Rsh)) / torch.linalg.norm(4.0 * (Rsh.T @ Rsh))).item())
        xgmi_s = 3 * hop.numel() * 8 / 153e9   # three hops of the packed triangle at one link's 153 GB/s (MI355X_MICROARCH.md)
        out["rank_tree"] = {"levels": 3, "seconds_three_merges_with_pack_unpack": d_tree, "one_merge_seconds": d_merge,
                            "packed_triangle_bytes_per_hop": hop.numel() * 8, "dense_square_bytes_per_hop": (P + 1) ** 2 * 8,
                            "xgmi_wire_seconds_three_hops_at_153_GB_per_s": xgmi_s, "relerr_RtR": err_tree,
                            "predicted_8_gpu_tsqr_seconds": dsh + d_tree + xgmi_s,
                            "predicted_scaling_at_8_gpus_shard_plus_tree": dt / (dsh + d_tree + xgmi_s),
                            "note": "N = 1 prediction: one rank's shard (tsqr_shard.seconds) + the three merge levels on rank 0's critical path + "
                                    "wire time of the packed triangles; no broadcast (dist.tsqr_tree(broadcast=False): the consumer of R lives on rank 0)"}
        out["tsqr_shard"]["scaling_ceiling_at_8_gpus_with_rank_tree"] = out["rank_tree"]["predicted_scaling_at_8_gpus_shard_plus_tree"]
        # ---- the fused Gram pass of the same shard: blocking and two in flight, against one eighth of the timed 1 M step
        Gsh = [torch.zeros((P + 1, P + 1), dtype=torch.float64, device=dev) for _ in range(2)]
        for _ in range(3):
            eng.gram(sts, rhs=rhss, out=Gsh[0])
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.gram(sts, rhs=rhss, out=Gsh[0])
        sync()
        g_blk = (time.perf_counter() - t0) / 20
        eng.wait(eng.gram_submit(sts, Gsh[0], rhs=rhss))
        sync()
        t0 = time.perf_counter()
        pend = None
        for i in range(40):
            tk = eng.gram_submit(sts, Gsh[i & 1], rhs=rhss)
            if pend is not None:
                eng.wait(pend)
            pend = tk
        eng.wait(pend)
        sync()
        g_pip = (time.perf_counter() - t0) / 40
        eighth = out["ms_per_step"] * 1e-3 / 8
        out["gram_shard"] = {"samples": Ssh, "blocking_seconds": g_blk, "pipelined_seconds": g_pip, "one_eighth_of_the_1M_step_seconds": eighth,
                             "blocking_ratio_to_one_eighth": g_blk / eighth, "pipelined_ratio_to_one_eighth": g_pip / eighth,
                             "scaling_ceiling_at_8_gpus_pipelined": 8.0 / (g_pip / eighth), "scaling_ceiling_at_8_gpus_blocking": 8.0 / (g_blk / eighth),
                             "allreduce_note": "the (P+1)^2 fp64 Gram (1.86 MB) is all-reduced asynchronously beside the next step's kernels (per_rank_allreduce_wait_ms_per_step)"}
        del sts, rhss, partners

    # ---- A3 inside the step (N = 1): the reference's floating-base loop simulates every sample (model.py:398-413) to obtain the base-wrench
    # rows of tau when only joint torques are measured; its own timers split simulate / regressor (model.py:626-628).  The same timed step
    # with fbr_inverse_dynamics_batch INSIDE the region (a second handle on its own stream: its blocking call does not drain the two Gram
    # submissions in flight), the six base rows copied into tau, then the fused pass.
    if world == 1 and on_gpu and hasattr(eng, "gram_submit"):
        eng2 = make_engine(args, topo, True, dev.index or 0)
        xstd = topo.x_std()
        sim = torch.empty((S, rows), dtype=torch.float64, device=dev)
        rhs_sim = [rhs.clone(), rhs.clone()]
        Gv = [torch.zeros((P + 1, P + 1), dtype=torch.float64, device=dev) for _ in range(2)]
        t_sim = [0.0]

        def sim_step(i, pend):
            b = i & 1
            t_ = time.perf_counter()
            eng2.inverse_dynamics(st, xstd, out=sim)                 # A3: every sample, a-priori parameters (blocking on eng2's stream)
            t_sim[0] += time.perf_counter() - t_
            rhs_sim[b].view(S, rows)[:, :6].copy_(sim[:, :6])        # base-wrench rows of tau := simulated (model.py:412-413)
            tk = eng.gram_submit(st, Gv[b], rhs=rhs_sim[b])
            if pend is not None:
                eng.wait(pend)
            return tk

        pend = None
        for i in range(3):
            pend = sim_step(i, pend)
        eng.wait(pend)
        sync()
        t_sim[0] = 0.0
        ks = max(4, args.steps // 2)
        t0 = time.perf_counter()
        pend = None
        for i in range(ks):
            pend = sim_step(i, pend)
        eng.wait(pend)
        sync()
        d_sim = (time.perf_counter() - t0) / ks
        # the rows of tau the simulation overwrites ARE the simulated ones in this data set (tau = ID + noise): the Gram must agree with the
        # timed steps' up to the noise of those six rows; checked on the noise-free part: [Y]^T[Y] block is identical
        same = float((torch.linalg.norm(Gv[(ks - 1) & 1][:P, :P] - G_sharded[:P, :P]) / torch.linalg.norm(G_sharded[:P, :P])).item())
        out["value_incl_simulate"] = S_total / d_sim
        out["simulate_in_step"] = {"ms_per_step": d_sim * 1e3, "steps": ks, "simulate_ms_per_step_host_clock": t_sim[0] / ks * 1e3,
                                   "regressor_and_gram_ms_per_step": (d_sim - t_sim[0] / ks) * 1e3, "ms_per_step_without_simulate": out["ms_per_step"],
                                   "relerr_YtY_block_vs_timed_steps": same,
                                   "what": "per step: fbr_inverse_dynamics_batch of all samples (A3, model.py:398-413; fused kinematics + RNEA kernel) -> the 6 "
                                           "base-wrench rows of tau replaced by the simulated ones -> fused regressor -> Gram pass (two submissions in flight); "
                                           "the split mirrors the reference's own two timers (model.py:626-628)"}
        eng2.close()
        del sim, rhs_sim, Gv

    # ---- weak scaling: 1 M samples PER GPU (N = 1: identical to the timed steps)
    if world > 1:
        off = (rank * S_total) // world  # every rank takes the whole data set, rotated by its shard offset
        stw_a, rhsw_a = with_tau(eng, topo, synth_range(topo, S_total, off, S_total, dev))
        stw_b, rhsw_b = with_tau(eng, topo, synth_range(topo, S_total, 0, off, dev))
        stw = {k: torch.cat([stw_a[k], stw_b[k]]).contiguous() for k in stw_a}
        rhsw = torch.cat([rhsw_a, rhsw_b]).contiguous()
        del stw_a, stw_b, rhsw_a, rhsw_b
        Gw = torch.zeros_like(G_sharded)

        def step_w():
            eng.gram(stw, rhs=rhsw, out=Gw)
            dist.all_reduce(Gw)

        ksteps = max(3, args.steps // 4)
        dtw = timed(step_w, ksteps, 1) / ksteps
        ok = float((torch.linalg.norm(Gw - world * G_sharded) / torch.linalg.norm(Gw)).item())
        out["weak_scaling"] = {"samples_per_gpu": S_total, "steps": ksteps, "ms_per_step": dtw * 1e3, "value": world * S_total / dtw,
                               "unit": "samples/s", "relerr_vs_world_x_sharded_gram": ok}
        del stw, rhsw
        return  # the single-GPU legs below are reported by the N = 1 line

    # ---- N = 1 only ---------------------------------------------------------------------------------------------------
    if args.sustain_seconds > 0:  # a long steady run of the same step (the driver's own GPU-busy sampling sees it)
        G2 = torch.zeros_like(G_sharded)
        sync()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < args.sustain_seconds:
            for _ in range(5):
                eng.gram(st, rhs=rhs, out=G2)
            sync()
            n += 5
        ds = time.perf_counter() - t0
        out["sustained"] = {"seconds": ds, "steps": n, "samples_per_s": n * S / ds}

    if on_gpu:
        # PCIe-inclusive rate (SURVEY 8(d) "incl. H2D of states"): pinned host inputs, staged chunk by chunk on the producer stream
        hst = {k: v.cpu().pin_memory() for k, v in st.items()}
        hrhs = rhs.cpu().pin_memory()
        Gh = np.zeros((P + 1, P + 1))
        eng.gram(hst, rhs=hrhs, out=Gh)
        eng.profile_enable(True)
        eng.profile_get()
        k2 = 8
        resident_same_moment = dth_cold = rounds = None
        if hasattr(eng, "gram_submit") and not os.environ.get("FBR_BENCH_BLOCKING"):
            # the same two-in-flight submission as the timed steps: the copy of a step's first chunk runs beside the previous step's
            # kernels; every step's Gram is brought back to a (pinned) host buffer inside the timed region.  The device-resident
            # inputs are timed the same way right before (the chip is warmer here than during the timed steps of the contract).
            Gd = [torch.zeros((P + 1, P + 1), dtype=torch.float64, device=dev) for _ in range(2)]
            Ghp = torch.zeros((P + 1, P + 1), dtype=torch.float64).pin_memory()

            def pipelined(stx, rhsx, warm=2):
                for i in range(warm):  # warm-up of the submission path with these inputs (staging buffers, copy stream)
                    eng.wait(eng.gram_submit(stx, Gd[i & 1], rhs=rhsx))
                eng.profile_get()
                pend = None
                sync()
                t0 = time.perf_counter()
                for i in range(k2):
                    tk = eng.gram_submit(stx, Gd[i & 1], rhs=rhsx)
                    if pend is not None:
                        eng.wait(pend[0])
                        Ghp.copy_(Gd[pend[1]])
                    pend = (tk, i & 1)
                eng.wait(pend[0])
                Ghp.copy_(Gd[pend[1]])
                sync()
                return (time.perf_counter() - t0) / k2

            resident_same_moment = pipelined(st, rhs)
            dth_cold = pipelined(hst, hrhs, warm=0)
            # rounds of k2 steps fall into one of two regimes (74.7 or 78 ms per step on the same box, the Gram launches themselves
            # 3 ms slower in the slow one: how the copies / producer kernels of a round happen to line up with its Gram launches):
            # three rounds, the median is reported, the spread beside it
            rounds = sorted([dth_cold] + [pipelined(hst, hrhs, warm=1) for _ in range(2)])
            dth = rounds[1]
            Gh = Ghp.numpy().copy()
            how = ("pinned host states + tau, hipMemcpyAsync per chunk on a copy stream (overlaps the kernels of the previous chunks and, across the "
                   "two submissions in flight, of the previous step), Gram copied back to pinned host memory every step")
        else:
            t0 = time.perf_counter()
            for _ in range(k2):
                eng.gram(hst, rhs=hrhs, out=Gh)
            dth = (time.perf_counter() - t0) / k2
            how = "pinned host states + tau, hipMemcpyAsync per chunk on a copy stream (overlaps the kernels of the previous chunks), host Gram out"
        prh = eng.profile_get()
        eng.profile_enable(False)
        nbytes = sum(v.numel() * 8 for v in hst.values()) + hrhs.numel() * 8
        out["value_incl_h2d"] = S / dth
        out["h2d"] = {"ms_per_step": dth * 1e3, "bytes_per_step": nbytes, "copy_ms_per_step": prh["h2d"][0] / k2,
                      "copy_GB_per_s": nbytes / (prh["h2d"][0] / k2 * 1e-3) / 1e9 if prh["h2d"][0] > 0 else None,
                      "relerr_vs_resident": float(np.linalg.norm(Gh - G_sharded.cpu().numpy()) / np.linalg.norm(Gh)),
                      "kernel_ms_per_step": {k_: v[0] / k2 for k_, v in prh.items() if v[1]},
                      # the device-resident inputs timed the same way seconds before: what the transfer itself costs
                      "resident_ms_per_step_same_moment": resident_same_moment * 1e3 if resident_same_moment else None,
                      "rounds_ms_per_step_min_median_max": [r * 1e3 for r in rounds] if resident_same_moment else None, "how": how}
        del hst, hrhs

        # materialising regressor kernel against the HBM roofline
        S3 = min(S, 60_000)
        sub3 = {k2_: v[:S3].contiguous() for k2_, v in st.items()}
        Y = torch.empty((S3 * rows, P), dtype=torch.float64, device=dev)
        eng.regressor(sub3, out=Y)
        eng.profile_enable(True)
        eng.profile_get()
        eng.regressor(sub3, out=Y)
        pr3 = eng.profile_get()
        eng.profile_enable(False)
        kms = pr3["regressor"][0]
        gbs = 8.0 * rows * P * S3 / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        out["assembly"] = {"samples": S3, "kernel": "fbr_regressor2_kernel", "kernel_ms": kms, "GB_per_s": gbs,
                           "frac_of_hbm_peak": gbs / PEAK_HBM_GBS, "bytes_per_sample": 8 * rows * P}
        del Y, sub3
        torch.cuda.empty_cache()
        out["predict"] = predict_leg(eng, topo, st, dev)
        out["fd_scores"] = fd_scores_leg(eng, topo, st, dev)
        out["other_configs"] = other_configs(args, dev, eng, topo, st, rhs)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(topo)
        try:
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(topo, out["cpu_baseline"]["value"])
        except Exception as e:  # secondary figure: never lose the bench line over it
            out["cpu_baseline_all_cores"] = {"error": repr(e)}
        try:
            bc, _, _ = cpu_phases_bc_and_parity(eng, topo)
            out["parity"] = bc.pop("parity", None)  # (top level: GPU vs CPU path on the same inputs, the SDP statement)
            out["cpu_baseline_phases"] = bc
        except Exception as e:
            out["cpu_baseline_phases"] = {"error": repr(e)}
    else:
        out["cpu_baseline"] = None


def _profiled(eng, fn, reps=5):
    """(seconds per call, {class: device ms per call}) of fn after a warm-up (the GPU clocks down during host-side work)."""
    import torch

    t_w, n_w = time.perf_counter(), 0
    while n_w < 1 or (time.perf_counter() - t_w < 0.05 and n_w < 50):
        fn()
        torch.cuda.synchronize()
        n_w += 1
    eng.profile_enable(True)
    eng.profile_get()
    # every call timed on its own, the MEDIAN reported: the GPU box grants the process 16 CPUs and a blocking wait of the driver thread
    # now and then takes tens of milliseconds -- one such call in five would otherwise be the figure
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    dt = sorted(each)[len(each) // 2]
    pr = eng.profile_get()
    eng.profile_enable(False)
    return dt, {k: v[0] / reps for k, v in pr.items() if v[1]}


def predict_leg(eng, topo, st, dev):
    """A9: tau = Y_s x streamed without materialising Y (fbr_predict, identifier.py:135-141), all resident samples.  HBM-bound by its
    algorithmic I/O: the states in, rows doubles out per sample; the kinematic records it stages in between are the traffic on top."""
    import torch

    S = st["q"].shape[0]
    rows, P, n = eng.rows, eng.cols, topo.num_dofs
    x = np.random.default_rng(3).standard_normal(P)
    tau = torch.empty((S, rows), dtype=torch.float64, device=dev)
    dt, kms = _profiled(eng, lambda: eng.predict(st, x, out=tau))
    alg = 8.0 * (3 * n + 15 + rows)
    fused = hasattr(eng, "get_option") and eng.get_option("fused_id") != 0
    if fused:
        # fbr_kinid_kernel (round 6): no records in HBM; what is staged are the branch-point records of the link-merged tree, written once and
        # read once per later child, lanes interleaved in a per-wave scratch (WALK-MAN: 2 saves + 4 loads of 21 doubles per sample)
        staged = 8.0 * 21 * 6
        kern, note = "fbr_kinid_kernel", ("one lane per sample, link records in registers, joint stack behind scalar branches (csrc/fbr_kinid.h); compute / "
                                           "latency bound at one wave per SIMD (342 VGPRs); round 5's two-kernel path staged 18.9 KB per sample")
    else:
        staged = 8.0 * (21 * topo.num_links + 6 * n) * 2  # kinematic records written by fbr_kin_kernel, read by fbr_id_kernel
        kern, note = "fbr_kin_kernel + fbr_id_kernel", "one lane (kinematics) / one wave (torques) per sample, records staged through HBM"
    kernel_s = sum(kms.values()) * 1e-3
    return {"entry_point": "fbr_predict", "samples": S, "seconds": dt, "samples_per_s": S / dt, "kernel_ms_per_call": kms,
            "roofline": {"bound": "hbm", "kernel": kern, "unit": "GB/s", "peak": PEAK_HBM_GBS,
                         "algorithmic_bytes_per_sample": alg, "achieved": alg * S / kernel_s / 1e9, "frac": alg * S / kernel_s / 1e9 / PEAK_HBM_GBS,
                         "staged_bytes_per_sample": staged, "achieved_incl_staging_GB_per_s": (alg + staged) * S / kernel_s / 1e9,
                         "note": note}}


def fd_scores_leg(eng, topo, st, dev, S=4096):
    """N1: the finite-difference sweep of the trajectory optimiser's gradient (fbr_fd_scores, analyticalGradient.py:92-185): 1 + 3 n
    regressor evaluations per sample, never stored; the weight blocks W are the only bulk input (its algorithmic HBM floor)."""
    import torch

    rows, P, n = eng.rows, eng.cols, topo.num_dofs
    sub = {k: v[:S].contiguous() for k, v in st.items()}
    W = torch.randn((S * rows, P), dtype=torch.float64, device=dev)
    out_ = torch.empty((S, 1 + 3 * n), dtype=torch.float64, device=dev)
    dt, kms = _profiled(eng, lambda: eng.fd_scores(sub, W, 1e-6, out=out_))
    evals = S * (1 + 3 * n)
    alg = 8.0 * (rows * P + 3 * n + 15 + 1 + 3 * n)
    kernel_s = sum(kms.values()) * 1e-3
    return {"entry_point": "fbr_fd_scores", "samples": S, "regressor_evaluations_per_sample": 1 + 3 * n, "seconds": dt,
            "evaluations_per_s": evals / dt, "samples_per_s": S / dt, "kernel_ms_per_call": kms,
            "roofline": {"bound": "hbm", "kernel": "fbr_kinfd_kernel (one lane per evaluation, nothing staged; option fused_id = 0: fbr_fd_expand_kernel + "
                                                   "fbr_kin_kernel + fbr_score_kernel)", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                         "algorithmic_bytes_per_sample": alg, "achieved": alg * S / kernel_s / 1e9, "frac": alg * S / kernel_s / 1e9 / PEAK_HBM_GBS,
                         "regressor_entries_evaluated_per_s": evals * rows * P / dt,
                         "note": "88 regressor blocks of 35 x 480 are evaluated per sample against ONE block of weights read: compute bound; the HBM "
                                 "figure is its algorithmic floor (the weight blocks)"}}


def other_configs(args, dev, eng4, topo4, st4, rhs4):
    """BASELINE.json configs[1], [2] and [4] on this GPU (secondary figures, HIP-event / wall timed, >= 3 repetitions)."""
    import scipy.linalg as sla
    import torch

    from flobaroid_amd import estimation as est
    from flobaroid_amd.topology import Topology

    res = {}
    reps = 3

    def timed(fn, reps=reps):
        # warm-up: at least one call and ~50 ms of them -- a short call right after host-side work meets a GPU that has clocked down
        # (measured: the first 6 ... 10 calls of a 5 ms kernel sequence then take 2 ... 3 x as long)
        t_w, n_w = time.perf_counter(), 0
        while n_w < 1 or (time.perf_counter() - t_w < 0.05 and n_w < 50):
            fn()
            torch.cuda.synchronize()
            n_w += 1
        # three batches of back-to-back calls, the median batch (a blocking wait of the driver thread now and then takes tens of
        # milliseconds on the GPU box: `_profiled`)
        per = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / reps)
        return sorted(per)[1]

    # configs[1]: KUKA LWR4 fixed base, 50 k samples: regressor + fused Gram + base-parameter QR
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", "kuka_lwr4.topology.json"))
    eng = make_engine(args, topo, False, dev.index or 0)
    S = 50_000
    st = synth_range(topo, S, 0, S, dev, floating=False, seed0=7)
    st.pop("_noise")
    tau = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
    Y = torch.empty((S * eng.rows, eng.cols), dtype=torch.float64, device=dev)
    t_reg = timed(lambda: eng.regressor(st, out=Y))
    t_gram = timed(lambda: eng.gram(st, rhs=tau))
    Gs = eng.gram(_np_states(topo, 5000, 3, floating=False))
    t0 = time.perf_counter()
    Rq = sla.qr(Gs, pivoting=True, mode="economic")[1]
    nb = int(np.count_nonzero(np.abs(np.diag(Rq)) > 1e-4))
    t_qr = time.perf_counter() - t0
    res["kuka_lwr4_fixed_50k"] = {"samples": S, "block": [eng.rows, eng.cols], "regressor_ms": t_reg * 1e3,
                                  "regressor_GB_per_s": 8.0 * S * eng.rows * eng.cols / t_reg / 1e9, "fused_gram_ms": t_gram * 1e3,
                                  "fused_gram_samples_per_s": S / t_gram, "base_param_qr_host_ms": t_qr * 1e3, "base_rank": nb}
    eng.close()
    del Y, st
    # configs[2]: WALK-MAN left arm (floating base per configs/walkman_left_arm.yaml: 13 x 90 block), 500 k samples: TSQR
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", "walkman_left_arm.topology.json"))
    eng = make_engine(args, topo, True, dev.index or 0)
    S = 500_000
    st = synth_range(topo, S, 0, S, dev, floating=True, seed0=8)
    st.pop("_noise")
    tau = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
    t_q = timed(lambda: eng.tsqr(st, rhs=tau))
    t_g = timed(lambda: eng.gram(st, rhs=tau))
    wi = eng.tsqr_work_info(S, k=1)
    fl = 2.0 * S * eng.rows * (eng.cols + 1) ** 2
    res["walkman_left_arm_floating_500k"] = {
        "samples": S, "block": [eng.rows, eng.cols], "tsqr_ms": t_q * 1e3, "repetitions": reps,
        "tsqr_executed_TFLOP_per_s": wi["flop"] / t_q / 1e12, "tsqr_executed_frac_of_fp64_mfma_peak": wi["flop"] / t_q / 1e12 / PEAK_FP64_MFMA_TFLOPS,
        "tsqr_dense_model_TFLOP_per_s": fl / t_q / 1e12, "tsqr_samples_per_s": S / t_q,
        "tsqr_hbm_floor_ms": 2 * 8.0 * S * eng.rows * wi["n_padded"] / (PEAK_HBM_GBS * 1e9) * 1e3,
        "fused_gram_ms": t_g * 1e3, "fused_gram_samples_per_s": S / t_g}
    eng.close()
    del st, tau
    torch.cuda.empty_cache()
    # configs[4]: WALK-MAN, 4 M samples: fused Gram + TSQR of the base regressor [YBase | tau] -> SDP inputs -> xStd
    S1 = st4["q"].shape[0]
    mult = max(1, 4_000_000 // S1)
    Rs = eng4.gram(_np_states(topo4, 10000, 99))
    Rs = Rs.cpu().numpy() if hasattr(Rs, "cpu") else Rs
    # (the deterministic tie rule of the host layer, model.pivoted_qr: the independent columns -- and with them the cost of this leg, which
    # depends on WHICH links' columns are factorised -- must not change with the last bits of the Gram)
    from flobaroid_amd.model import pivoted_qr

    Q, RQ, PQ = pivoted_qr(Rs)
    r = int(np.count_nonzero(np.abs(np.diag(RQ)) > 0.005))
    ic = np.sort(PQ[:r]).astype(np.int32)
    P = eng4.cols
    # 4 M samples = the 1 M resident ones and three more seeded blocks, streamed through R_in / accumulate (what a host with less
    # memory than the data set does); every pass is timed
    R_b = None
    G5 = torch.zeros((P + 1, P + 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t_tsqr = t_gram = 0.0
    for i in range(mult):
        if i == 0:
            sti, rhsi = st4, rhs4
        else:
            sti, rhsi = with_tau(eng4, topo4, synth_range(topo4, S1, 0, S1, dev, seed0=1000 + 8 * i))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng4.gram(sti, rhs=rhsi, out=G5, accumulate=i > 0)
        torch.cuda.synchronize()
        t_gram += time.perf_counter() - t0
        t0 = time.perf_counter()
        R_b = eng4.tsqr(sti, rhs=rhsi, cols=ic, R_in=R_b)
        torch.cuda.synchronize()
        t_tsqr += time.perf_counter() - t0
    Rb = R_b.cpu().numpy()
    G5n = G5.cpu().numpy()
    sel = np.concatenate((ic, [P]))
    Gb = G5n[np.ix_(sel, sel)]
    relerr = float(np.linalg.norm(Rb.T @ Rb - Gb) / np.linalg.norm(Gb))
    M = mult * S1 * eng4.rows
    nbp = len(ic)
    xb, sv = est.lstsq_from_R(Rb, nbp, 0, M)
    wi = eng4.tsqr_work_info(S1, k=1, cols=ic)
    res["walkman_full_4M_gram_tsqr_sdp_inputs"] = {
        "samples": mult * S1, "passes": mult, "base_params": nbp, "gram_seconds": t_gram, "gram_samples_per_s": mult * S1 / t_gram,
        "tsqr_base_columns": nbp + 1, "tsqr_seconds": t_tsqr, "tsqr_samples_per_s": mult * S1 / t_tsqr,
        "tsqr_executed_TFLOP_per_s": mult * wi["flop"] / t_tsqr / 1e12,
        "tsqr_dense_model_TFLOP_per_s": 2.0 * M * (nbp + 1) ** 2 / t_tsqr / 1e12,
        "relerr_RtR_vs_gram": relerr,
        # ||tau - YBase xBase||^2 of the least-squares fit = the square of the last diagonal entry of the augmented factor; the
        # synthetic torques carry N(0, 0.05^2) noise per row (rho2_norm_sqr of sdp.py:482-485 without contacts)
        "rho2_norm_sqr": float(Rb[nbp, nbp] ** 2), "rho2_expected_from_noise": 0.05 ** 2 * (M - nbp),
        "lstsq_default_rcond_kept_singular_values": int(np.count_nonzero(sv > np.finfo(float).eps * M * sv[0])),
        "note": "R1 = R[:nb,:nb], rho1 = R[:nb,nb] are the SDP inputs of sdp.py:470-487 (estimation.sdp_inputs)"}
    # WALK-MAN's shipped identification mode (walkman_full.yaml:265 useBaseWrenchForBaseParams: only the 6 base-wrench rows of every
    # sample enter the fit, identifier.py:629-636): a 0/1 row mask; masked rows are detected on the device and skipped
    wmask = torch.zeros((S1, eng4.rows), dtype=torch.float64, device=dev)
    wmask[:, :6] = 1.0
    wmask = wmask.reshape(-1)

    def timed(fn, reps=3):
        t_w, n_w = time.perf_counter(), 0
        while n_w < 1 or (time.perf_counter() - t_w < 0.05 and n_w < 50):  # (warm-up as above)
            fn()
            torch.cuda.synchronize()
            n_w += 1
        # three batches of back-to-back calls, the median batch (a blocking wait of the driver thread now and then takes tens of
        # milliseconds on the GPU box: `_profiled`)
        per = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / reps)
        return sorted(per)[1]

    Gm = eng4.gram(st4, rhs=rhs4, w=wmask)
    Rm = eng4.tsqr(st4, rhs=rhs4, w=wmask, cols=ic)
    Gmn = Gm.cpu().numpy()[np.ix_(sel, sel)]
    t_gm = timed(lambda: eng4.gram(st4, rhs=rhs4, w=wmask, out=G5))
    t_qm = timed(lambda: eng4.tsqr(st4, rhs=rhs4, w=wmask, cols=ic))
    res["walkman_full_1M_base_wrench_rows_only"] = {
        "samples": S1, "rows_per_sample_in_the_fit": 6, "fused_gram_ms": t_gm * 1e3, "fused_gram_samples_per_s": S1 / t_gm,
        "tsqr_base_columns": nbp + 1, "tsqr_ms": t_qm * 1e3, "tsqr_samples_per_s": S1 / t_qm,
        "relerr_RtR_vs_gram": float(np.linalg.norm(Rm.cpu().numpy().T @ Rm.cpu().numpy() - Gmn) / np.linalg.norm(Gmn))}
    # one Gram per candidate trajectory in one pass (the optimiser's inner loop, trajectoryOptimizer.py:248-272):
    # 64 candidates x 2000 samples of the resident data, against the Gram of the same samples taken as one batch
    ng, per = 64, 2000
    if S1 >= ng * per and hasattr(eng4, "gram_grouped"):
        sub = {k_: v[: ng * per] for k_, v in st4.items()}
        Gg = eng4.gram_grouped(sub, ng)
        G1 = eng4.gram(sub)
        t_gg = timed(lambda: eng4.gram_grouped(sub, ng, out=Gg), reps=10)  # (out=: no 118 MB allocation inside the timed calls)
        each = []
        for _ in range(12):  # (single calls: a stall that comes and goes shows as a spread here)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng4.gram_grouped(sub, ng, out=Gg)
            torch.cuda.synchronize()
            each.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        host = []
        for _ in range(6):  # (back to back: how long each call keeps the HOST)
            t0 = time.perf_counter()
            eng4.gram_grouped(sub, ng, out=Gg)
            host.append(round((time.perf_counter() - t0) * 1e3, 3))
        torch.cuda.synchronize()
        eng4.profile_enable(True)
        eng4.profile_get()
        eng4.gram_grouped(sub, ng, out=Gg)
        torch.cuda.synchronize()
        prof_gg = {k_: round(v_[0], 3) for k_, v_ in eng4.profile_get().items() if v_[1]}
        eng4.profile_enable(False)
        prof_gg["host_ms_of_back_to_back_calls"] = host
        prof_gg["single_calls_ms_min_median_max"] = [round(min(each), 3), round(sorted(each)[len(each) // 2], 3), round(max(each), 3)]
        eng4.set_option("reduce_grouped_min_samples", 1e18)  # (no group is that long: the grouped pass over all columns)
        try:
            t_gg_all = timed(lambda: eng4.gram_grouped(sub, ng, out=Gg), reps=10)
        finally:
            eng4.set_option("reduce_grouped_min_samples", 512)
        t_single = sorted(each)[len(each) // 2] * 1e-3
        res["walkman_64_candidates_x_2000_grouped_gram"] = {
            # the optimiser's use: one call per iteration, its result read back before the next one -- the synchronised single call.  Ten calls
            # enqueued back to back (`back_to_back_ms`) have measured 4 ... 13 ms per call in this process (the same kernels, 4.0 ms of them, in
            # every case; not reproduced outside bench.py: DESIGN 10)
            "groups": ng, "samples_per_group": per, "grouped_gram_ms": t_single * 1e3, "samples_per_s": ng * per / t_single,
            "back_to_back_ms": t_gg * 1e3,
            "grouped_gram_ms_over_all_columns": t_gg_all * 1e3, "kernel_ms_per_call": prof_gg,
            "relerr_sum_of_groups_vs_one_batch": float(torch.linalg.norm(Gg.sum(dim=0) - G1) / torch.linalg.norm(G1))}
    return res


def main():
    args = parse_args()
    maybe_spawn(args)
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
