#!/usr/bin/env python3
"""bench.py -- regressor samples/s of the fused hot path on MI355X (contract: see the task statement).

A step = one pass of the hot path over this rank's batch of synthetic WALK-MAN floating-base samples
that are already resident in HBM: link kinematics + fused regressor -> [Y|tau]^T[Y|tau] Gram (fp64 MFMA),
reduced deterministically on the device, then (N > 1) an RCCL all-reduce of the (P+1)^2 Gram.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # vendor dense fp64 matrix peak of MI355X; not in MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def synth_states(topo, S, seed, floating=True):
    """Synthetic inputs of SURVEY.md §8(d): seeded, the distributions of model.py:696-725."""
    rng = np.random.default_rng(seed)
    n = topo.num_dofs
    lo = np.array([topo.limits[j]["lower"] for j in topo.dof_names])
    hi = np.array([topo.limits[j]["upper"] for j in topo.dof_names])
    vm = np.array([topo.limits[j]["velocity"] for j in topo.dof_names])
    st = dict(q=lo + (hi - lo) * rng.random((S, n)), dq=(rng.random((S, n)) - 0.5) * 2 * vm,
              ddq=(rng.random((S, n)) - 0.5) * 2 * np.pi)
    if floating:
        st.update(base_vel=np.pi * rng.random((S, 6)), base_acc=np.pi * rng.random((S, 6)), rpy=0.1 * rng.random((S, 3)))
    return st, rng


def cpu_baseline(topo, budget_s=15.0):
    """Oracle (C restatement, 1 thread) timed on the host: per-sample regressor + RNEA torques + the
    A^T A accumulation of the stacked block with NumPy (BLAS pinned to 1 thread)."""
    from oracle.oracle import OracleModel

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    om = OracleModel(topo, floating=True)
    x = topo.x_std()
    block = 256
    st, _ = synth_states(topo, block, 1234)
    done = 0
    G = np.zeros((om.P + 1, om.P + 1))
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
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
    if ctx is not None:
        ctx.unregister() if hasattr(ctx, "unregister") else None
    return {"value": done / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{done} WALK-MAN floating-base samples: oracle regressor + RNEA (C, 1 thread) + NumPy A^T A (1 BLAS thread), {dt:.1f} s"}


def cpu_baseline_all_cores(topo, budget_s=8.0):
    """The same port on every host core: one Python thread per core, each running the C oracle (ctypes releases the GIL)
    and its own 1-thread NumPy A^T A on private sample blocks, Grams summed at the end ("1 thread, then all
    host cores").  The reference itself is single-threaded on this path."""
    import concurrent.futures as cf

    from oracle.oracle import OracleModel

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    cores = os.cpu_count() or 1
    x = topo.x_std()
    block = 256
    st, _ = synth_states(topo, block, 4321)
    oms = [OracleModel(topo, floating=True) for _ in range(cores)]
    t_end = time.perf_counter() + budget_s

    def work(i):
        om = oms[i]
        G = np.zeros((om.P + 1, om.P + 1))
        done = 0
        while time.perf_counter() < t_end:
            Y = om.regressor(st)
            tau = om.inverse_dynamics(st, x).reshape(-1, 1)
            Ya = np.hstack([Y, tau])
            G += Ya.T @ Ya
            done += block
        return done, G

    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    if ctx is not None and hasattr(ctx, "unregister"):
        ctx.unregister()
    done = sum(r[0] for r in res)
    return {"value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{done} WALK-MAN floating-base samples over {cores} threads (C oracle + 1-thread NumPy A^T A each), {dt:.1f} s"}


def other_configs(dev):
    """BASELINE.json configs[1] and configs[2] on this GPU (secondary figures, outside the timed steps):
    KUKA LWR4 fixed base, 50 k samples: regressor assembly + fused Gram + base-parameter QR (host, 80 x 80);
    WALK-MAN left arm (floating base, 13 x 90 block), 500 k samples: Householder TSQR."""
    import torch

    from flobaroid_amd._lib import Engine
    from flobaroid_amd.topology import Topology
    from flobaroid_amd import estimation as est

    res = {}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # configs[1]
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", "kuka_lwr4.topology.json"))
    eng = Engine(topo, floating=False, device=dev.index or 0)
    S = 50_000
    st_np, rng = synth_states(topo, S, 7, False)
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
    tau = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
    Y = torch.empty((S * eng.rows, eng.cols), dtype=torch.float64, device=dev)
    t_reg = timed(lambda: eng.regressor(st, out=Y))
    t_gram = timed(lambda: eng.gram(st, rhs=tau))
    G = eng.gram(st, rhs=tau).cpu().numpy()
    t0 = time.perf_counter()
    import scipy.linalg as sla

    Gs = G[: eng.cols, : eng.cols]
    Rq = sla.qr(Gs, pivoting=True, mode="economic")[1]
    nb = int(np.count_nonzero(np.abs(np.diag(Rq)) > 1e-4 * np.abs(Rq[0, 0])))
    t_qr = time.perf_counter() - t0
    res["kuka_lwr4_fixed_50k"] = {"samples": S, "block": [eng.rows, eng.cols], "regressor_ms": t_reg * 1e3,
                                  "regressor_GB_per_s": 8.0 * S * eng.rows * eng.cols / t_reg / 1e9, "fused_gram_ms": t_gram * 1e3,
                                  "fused_gram_samples_per_s": S / t_gram, "base_param_qr_host_ms": t_qr * 1e3, "base_rank": nb}
    eng.close()
    del Y, st
    # configs[2]
    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", "walkman_left_arm.topology.json"))
    eng = Engine(topo, floating=True, device=dev.index or 0)
    S = 500_000
    st_np, rng = synth_states(topo, S, 8, True)
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
    tau = torch.randn((S * eng.rows, 1), dtype=torch.float64, device=dev)
    t_q = timed(lambda: eng.tsqr(st, rhs=tau), reps=2)
    t_g = timed(lambda: eng.gram(st, rhs=tau), reps=2)
    fl = 2.0 * S * eng.rows * (eng.cols + 1) ** 2
    res["walkman_left_arm_floating_500k"] = {"samples": S, "block": [eng.rows, eng.cols], "tsqr_ms": t_q * 1e3, "tsqr_TFLOP_per_s": fl / t_q / 1e12,
                                             "tsqr_samples_per_s": S / t_q, "fused_gram_ms": t_g * 1e3, "fused_gram_samples_per_s": S / t_g}
    eng.close()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1_000_000, help="samples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-configs", action="store_true",
                    help="also time BASELINE.json configs[1] (KUKA 50 k) and configs[2] (left arm 500 k) as secondary figures; "
                         "off by default so that the kernel statistics of the default command contain the headline workload only")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run: one process per GPU over RCCL
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from flobaroid_amd._lib import Engine
    from flobaroid_amd.topology import Topology

    topo = Topology.load(os.path.join(ROOT, "flobaroid_amd", "robots", "walkman_apriori.topology.json"))
    eng = Engine(topo, floating=True, device=local)
    eng.use_torch_stream()
    S = args.samples
    st_np, rng = synth_states(topo, S, 42 + rank)
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
    # tau = ID(state; xStdModel) + N(0, 0.05^2)  (tests/test_identification.py:79), computed on the device
    tau = eng.inverse_dynamics(st, topo.x_std())
    tau += 0.05 * torch.randn(tau.shape, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(7 + rank))
    rhs = tau.reshape(-1, 1).contiguous()
    Pa = eng.cols + 1
    G = torch.zeros((Pa, Pa), dtype=torch.float64, device=dev)

    def step():
        eng.gram(st, rhs=rhs, out=G)
        if use_dist:
            dist.all_reduce(G)  # (P+1)^2 fp64 = 1.85 MB: the only exchange step of the pass

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile_enable(True)
    eng.profile_get()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_get()
    eng.profile_enable(False)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = world * S / (dt / args.steps)
    rows, P = eng.rows, eng.cols
    info = eng.gram_program_info(1)
    gram_ms, gram_n = prof["gram"]
    samples_per_launch = S * args.steps / max(gram_n, 1)
    # symmetric Gram count rows*P*(P+1) + 2*rows*P per rhs column, per sample
    alg_flop_per_sample = rows * P * (P + 1) + 2 * rows * P * 1
    avg_launch_s = gram_ms / max(gram_n, 1) * 1e-3
    achieved = alg_flop_per_sample * samples_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
    executed_flop_per_sample = info["mfma_per_sample"] * 2 * 16 * 16 * 4
    out = {
        "metric": "regressor samples/s (fused regressor->Gram pass), WALK-MAN float-base",
        "value": value,
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"walkman_apriori floating base (29 DOF, 48 links, {rows}x{P} regressor block), "
                        f"{S} samples per GPU per step, fused kinematics + [Y|tau]^T[Y|tau] Gram"
                        + (" + RCCL all-reduce" if world > 1 else ""),
            "samples_per_gpu": S,
            "rhs_columns": 1,
            "parallelism": f"samples sharded over {world} GPU(s)",
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "fbr_gram_kernel",
            "achieved": achieved,
            "peak": PEAK_FP64_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
            "traffic": None,
            "algorithmic_flop_per_sample": alg_flop_per_sample,
            "executed_mfma_flop_per_sample": executed_flop_per_sample,
            # what the MFMA pipe really ran: the kernel skips the structurally zero k-steps of the tile pairs, so `achieved`
            # (contract definition: dense-symmetric algorithmic flops / launch time) can exceed the hardware peak
            "executed": executed_flop_per_sample * samples_per_launch / avg_launch_s / 1e12,
            "executed_frac": executed_flop_per_sample * samples_per_launch / avg_launch_s / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "avg_launch_ms": avg_launch_s * 1e3,
            "launches": gram_n,
            "samples_per_launch": samples_per_launch,
        },
        "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1]},
    }
    # HBM traffic of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 x2
    # correction of FETCH_SIZE) are committed under profiles/; scaled here to this run's samples per launch
    try:
        import glob

        src = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gram_pmc_traffic.json")))[-1]  # latest committed PMC summary
        pm = json.load(open(src))
        out["roofline"]["traffic"] = pm["hbm_bytes_per_sample"] * samples_per_launch
        out["roofline"]["traffic_source"] = f"profiles/{os.path.basename(src)} (HBM bytes per sample x samples per launch)"
    except Exception:
        pass
    if world == 1:
        # secondary figures of the path (not part of the timed steps): Householder TSQR GF/s and the
        # materialising regressor kernel against the HBM roofline
        S2 = min(S, 150_000)
        sub = {k2: v[:S2].contiguous() for k2, v in st.items()}
        rhs2 = rhs[: S2 * rows].contiguous()
        eng.tsqr(sub, rhs=rhs2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.tsqr(sub, rhs=rhs2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        tflop = 2.0 * S2 * rows * (P + 1) ** 2 / dt2 / 1e12
        out["tsqr"] = {"samples": S2, "columns": P + 1, "seconds": dt2, "samples_per_s": S2 / dt2, "TFLOP_per_s": tflop,
                       "frac_of_fp64_mfma_peak": tflop / PEAK_FP64_MFMA_TFLOPS,
                       "flop_model": "2*rows*(P+k)^2 per sample (SURVEY 8d)"}
        S3 = min(S, 60_000)
        sub3 = {k2: v[:S3].contiguous() for k2, v in st.items()}
        Y = torch.empty((S3 * rows, P), dtype=torch.float64, device=dev)
        eng.regressor(sub3, out=Y)
        eng.profile_enable(True)
        eng.profile_get()
        eng.regressor(sub3, out=Y)
        pr3 = eng.profile_get()
        eng.profile_enable(False)
        kms = pr3["regressor"][0]
        gbs = 8.0 * rows * P * S3 / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        out["assembly"] = {"samples": S3, "kernel": "fbr_regressor_kernel", "kernel_ms": kms, "GB_per_s": gbs,
                           "frac_of_hbm_peak": gbs / PEAK_HBM_GBS, "bytes_per_sample": 8 * rows * P}
        del Y
        if args.other_configs:
            out["other_configs"] = other_configs(dev)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(topo)
        try:
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(topo)
        except Exception as e:  # secondary figure: never lose the bench line over it
            out["cpu_baseline_all_cores"] = {"error": repr(e)}
    elif world == 1:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
