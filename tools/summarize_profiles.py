#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into the small files committed under profiles/:

  profiles/<tag>_bench_default.json        the bench line of the unprofiled default run
  profiles/<tag>_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of `python bench.py`
  profiles/<tag>_pmc.json                  per-kernel PMC sums per launch (FETCH_SIZE, WRITE_SIZE, MFMA, SQ)
  profiles/<tag>_gram_pmc_traffic.json     HBM bytes per sample of the Gram kernel (FETCH_SIZE x2: MI355X_MICROARCH.md, HBM)
"""
import csv, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n


def counters(prefix):
    path = os.path.join(src, prefix + "_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if not k.startswith("fbr_"):
                continue
            e = acc[row["Counter_Name"]][k]
            e[0] += float(row["Counter_Value"])
            e[1].add(row["Dispatch_Id"])
    return {c: {k: {"launches": len(v[1]), "per_launch": v[0] / max(len(v[1]), 1)} for k, v in ks.items()} for c, ks in acc.items()}


shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_default.json")) if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, tag + "_bench_default.json"), "w"), indent=1)

if os.path.exists(os.path.join(src, "tsqr_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "tsqr_kernel_stats.csv"), os.path.join(dst, tag + "_tsqr_kernel_stats.csv"))

pmc = {}
for prefix in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_sq"):
    pmc.update(counters(prefix))
tpmc = {}
for prefix in ("tsqr_pmc_mfma", "tsqr_pmc_sq", "tsqr_pmc_fetch"):
    tpmc.update(counters(prefix))
if tpmc:
    json.dump({"command": "rocprofv3 --pmc <set> --kernel-trace -- python tools/tsqr_pmc_probe.py (one pass per set; WALK-MAN 150 k samples x 481 columns, "
                          "left arm 500 k x 91; two tsqr calls each -> launches)",
               "unit": "raw counts per launch; SQ_INSTS_VALU_MFMA_MOPS_F64 = 4 per v_mfma_f64_16x16x4_f64 (2048 flop); FETCH_SIZE in KB (x2 on gfx950)",
               "counters": tpmc}, open(os.path.join(dst, tag + "_tsqr_pmc.json"), "w"), indent=1)
json.dump({"command": "rocprofv3 --pmc <set> --kernel-trace -- python bench.py --samples 200000 --steps 1 --warmup 0 --no-secondary "
                      "(one pass per set: FETCH_SIZE | WRITE_SIZE | MFMA | SQ; tools/profile_round.sh)",
           "unit": "FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH_SIZE needs x2 on gfx950), others raw counts", "counters": pmc},
          open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)

S = 200000
def is_gram(k):
    return k.startswith("fbr_gram_kernel") or k.startswith("fbr_gram64_kernel")


gk = sorted((k for k in pmc.get("FETCH_SIZE", {}) if is_gram(k)), key=lambda k: -pmc["FETCH_SIZE"][k]["per_launch"] * pmc["FETCH_SIZE"][k]["launches"])
if gk:
    g = gk[0]
    nl = pmc["FETCH_SIZE"][g]["launches"]
    spl = S / nl
    fetch_kb = pmc["FETCH_SIZE"][g]["per_launch"]
    write_kb = pmc.get("WRITE_SIZE", {}).get(g, {}).get("per_launch", 0.0)
    out = {"command": "tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE passes, 200000 samples)", "samples_per_gram_launch": spl, "gram_kernel_name": g,
           "gram_kernel": {"fetch_KB_per_launch_raw": fetch_kb, "write_KB_per_launch_raw": write_kb,
                           "fetch_bytes_per_sample_corrected": 2.0 * fetch_kb * 1024 / spl,
                           "write_bytes_per_sample": write_kb * 1024 / spl},
           "hbm_bytes_per_sample": 2.0 * fetch_kb * 1024 / spl + write_kb * 1024 / spl,
           "correction": "FETCH_SIZE x2 on gfx950 for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
    # staging traffic of the whole fused pass: every kernel of the step (kinematics, packer, Gram, reductions), counter bytes per sample.
    # FETCH_SIZE x2 only for the Gram kernel's 16-byte-per-lane LDS-DMA stream (the calibrated case of the guide); the 8-byte reads of the
    # kinematics / packer are taken as reported (uncalibrated there: a lower bound)
    # (only the launches of the PASS: the kinematics launches on the packer's queue -- bench.py's data generation runs the kinematics
    # and fbr_id_kernel of the unreduced model on the main stream before the step)
    def raw(prefix):
        path = os.path.join(src, prefix + "_counter_collection.csv")
        return list(csv.DictReader(open(path))) if os.path.exists(path) else []

    per_kernel = {}
    for prefix, field, factor_gram in (("pmc_fetch", "fetch", 2.0), ("pmc_write", "write", 1.0)):
        rows_ = raw(prefix)
        pq = {r["Queue_Id"] for r in rows_ if short(r["Kernel_Name"]).startswith("fbr_pack_kernel")}
        for r in rows_:
            k = short(r["Kernel_Name"])
            if not k.startswith("fbr_") or k.startswith("fbr_id_kernel") or k.startswith("fbr_kinid_kernel") or (k.startswith("fbr_kin_kernel") and r["Queue_Id"] not in pq):
                continue
            v = float(r["Counter_Value"]) * 1024 / S * (factor_gram if is_gram(k) else 1.0)
            per_kernel.setdefault(k, {"fetch": 0.0, "write": 0.0})[field] += v
    per_kernel = {k: {f: round(x, 1) for f, x in v.items()} for k, v in per_kernel.items() if v["fetch"] + v["write"] > 1.0}
    out["staging"] = {"per_kernel_bytes_per_sample": per_kernel,
                      "hbm_bytes_per_sample_all_kernels": round(sum(v["fetch"] + v["write"] for v in per_kernel.values()), 1),
                      "note": "sum over the launches of the 200000-sample pass / 200000"}
    json.dump(out, open(os.path.join(dst, tag + "_gram_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
print("bench:", bench["value"], bench["roofline"]["avg_launch_ms"], bench.get("tsqr", {}).get("executed_TFLOP_per_s"))

# ---- the whole TSQR call (1 M samples) and one rank's shard at 8 GPUs (125 k): kernel split per call
import re
import statistics

splits = {}
for name, reps in (("tsqr_full", 4), ("tsqr_shard", 6)):   # tools/tsqr_probe.py: one warm-up call + the timed ones
    f = os.path.join(src, name + "_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    shutil.copy(f, os.path.join(dst, tag + "_" + name + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    cls = collections.defaultdict(float)
    for r in rows:
        n = short(r["Name"])
        key = ("merge trees" if "tree" in n else "level-0 folds" if "level0" in n else "regressor writer (lane per sample, kinematics fused)" if "kinwrite" in n else "regressor writer" if "regressor" in n else
               "kinematics" if "kin" in n else "other")
        cls[key] += float(r["TotalDurationNs"]) / reps / 1e6
    splits[name] = {"kernel_ms_per_call": {k: round(v, 3) for k, v in sorted(cls.items(), key=lambda kv: -kv[1])},
                    "sum_ms_per_call": round(tot / reps / 1e6, 3), "tree_share_of_kernel_time": round(cls["merge trees"] / (tot / reps / 1e6), 4),
                    "stdout": open(os.path.join(src, name + "_stdout.txt")).read().strip().splitlines()[-1][:400]}
if splits:
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- python tools/tsqr_probe.py 1000000 3 | 125000 5 (tools/profile_round.sh); kernel time "
                          "summed per class and divided by the calls (trees of the row groups overlap on side streams: the sum exceeds the wall time)",
               "splits": splits}, open(os.path.join(dst, tag + "_tsqr_call_split.json"), "w"), indent=1)

# ---- Gram launches of the bench command: the full-size launches (62 500 samples) apart from the shorter ones of the other legs
gl = []
tr = os.path.join(src, "bench_kernel_trace.csv")
if os.path.exists(tr):
    for r in csv.DictReader(open(tr)):
        if "fbr_gram_kernel" in r["Kernel_Name"] or "fbr_gram64_kernel" in r["Kernel_Name"]:
            gl.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
full = [x for x in gl if x > 0.8 * statistics.median(gl)] if gl else []
gram_line = (f"`{bench['roofline'].get('kernel', 'fbr_gram_kernel')}`: {len(gl)} launches in the bench command, {len(full)} of them full-size ({bench['roofline']['samples_per_launch']:.0f} samples): median {statistics.median(full):.3f} ms, "
             f"mean {statistics.mean(full):.3f} ms (the bench's HIP events: {bench['roofline']['avg_launch_ms']:.3f} ms over the timed steps)") if full else ""

# ---- profiles/README.md is GENERATED here (numbers cannot go stale)
traffic = json.load(open(os.path.join(dst, tag + "_gram_pmc_traffic.json")))["hbm_bytes_per_sample"] if os.path.exists(os.path.join(dst, tag + "_gram_pmc_traffic.json")) else None
mops = None
for k, v in pmc.get("SQ_INSTS_VALU_MFMA_MOPS_F64", {}).items():
    if is_gram(k):
        mops = v
lines = [
    "# profiles/", "",
    f"Generated by `tools/summarize_profiles.py {tag}` from `gpurun_out/prof_{tag}/` (`tools/profile_round.sh {tag}`, one MI355X, the final code of the round). "
    f"`{tag}_*` is the current evidence; older tags are the earlier rounds, kept for the history in DESIGN §11.", "",
    f"* `{tag}_bench_default.json` – the line printed by `python bench.py` (unprofiled run, CPU baselines included): "
    f"{bench['value'] / 1e6:.2f} M samples/s, {bench['ms_per_step']:.2f} ms/step, Gram kernel {bench['roofline']['frac']:.3f} of the fp64 MFMA peak (executed); "
    f"TSQR {bench['tsqr']['seconds'] * 1e3:.1f} ms per 1 M samples = {bench['tsqr']['executed_frac_of_fp64_mfma_peak']:.3f} executed"
    + (f"; 125 k-sample shard {bench['tsqr_shard']['seconds'] * 1e3:.1f} ms = {bench['tsqr_shard']['ratio_to_one_eighth']:.2f} x one eighth of the 1 M call" if "tsqr_shard" in bench else ""),
    f"* `{tag}_bench_kernel_stats.csv` – `rocprofv3 --kernel-trace --stats` summary of `python bench.py --no-cpu-baseline --sustain-seconds 0` "
    f"(timed steps + TSQR / shard / H2D / assembly / other-config legs). {gram_line}",
    f"* `{tag}_pmc.json` – per-kernel PMC sums per launch of the fused pass (FETCH_SIZE, WRITE_SIZE, MFMA, SQ counters; separate passes, "
    "`bench.py --samples 200000 --steps 1 --warmup 0 --no-secondary`)"
    + (f": `SQ_INSTS_VALU_MFMA_MOPS_F64` = {mops['per_launch']:.0f} per launch of {S // mops['launches']} samples = "
       f"{mops['per_launch'] / (S / mops['launches']) / 4:.0f} MFMAs per sample (the program's count: {bench['roofline']['executed_mfma_per_sample']})" if mops else ""),
    f"* `{tag}_gram_pmc_traffic.json` – HBM-side bytes per sample of the Gram kernel (FETCH_SIZE × 2 on gfx950, MI355X_MICROARCH.md)"
    + (f": {traffic / 1024:.0f} KB; read by bench.py for `roofline.traffic`" if traffic else ""),
    f"* `{tag}_tsqr_pmc.json`, `{tag}_tsqr_kernel_stats.csv` – the same counter sets and the kernel trace of `tools/tsqr_pmc_probe.py` "
    "(WALK-MAN 150 k samples × 481 columns -- factorised over the regrouped 214 --, left arm 500 k × 91)",
]
# executed-MFMA model of the TSQR (fbr_tsqr_work_info, what bench.py's TSQR fractions are computed from) against the PMC count of the same calls
try:
    import ast
    wi_sum = 0
    for l in open(os.path.join(src, "tsqr_pmc_stdout.txt")):
        if "'mfma_level0'" in l:
            dct = ast.literal_eval(l[l.index("{"):].strip())
            wi_sum += dct["mfma_level0"] + dct["mfma_tree"]
    mops_sum = sum(v["per_launch"] * v["launches"] for v in tpmc.get("SQ_INSTS_VALU_MFMA_MOPS_F64", {}).values())
    if wi_sum and mops_sum:
        lines.append(f"  - executed-MFMA model against the counters: `fbr_tsqr_work_info` of the two probe calls {wi_sum} MFMAs, "
                     f"`SQ_INSTS_VALU_MFMA_MOPS_F64` / 4 summed over their kernels {mops_sum / 4:.0f} (ratio {mops_sum / 4 / wi_sum:.4f})")
except Exception as e:  # noqa: BLE001
    print("tsqr work-info check skipped:", e)
if splits:
    for name, what in (("tsqr_full", "1 M samples"), ("tsqr_shard", "125 k samples (one rank's share at 8 GPUs)")):
        if name in splits:
            sp = splits[name]
            lines.append(f"* `{tag}_{name}_kernel_stats.csv`, `{tag}_tsqr_call_split.json` – kernel split of the whole `fbr_tsqr` call, {what}: "
                         + ", ".join(f"{k} {v} ms" for k, v in sp["kernel_ms_per_call"].items())
                         + f" per call; merge trees = {100 * sp['tree_share_of_kernel_time']:.1f} % of the kernel time")
lines += [
    "* `r02_coissue_probe.txt` – `tools/coissue_probe.hip`: fp64 VALU beside fp64 MFMA on one SIMD (they share the DP pipe)",
    "* `r02_tsqr_timing.txt` – `tools/tsqr_timing_probe.py` (engine option `tsqr_timing`): per-phase cycles of the wide TSQR fold",
    "* `r05a/b/c_*` – round 5 (kinematics kernel + workgroup-per-sample writers / torque kernels, before the fused one-lane-per-sample kernels of round 6)",
    "* `r04_*` – this round BEFORE the column reductions (all 480 columns: 12.7 M samples/s on that box, Gram kernel 0.41, TSQR 160.6 ms); "
    "`r03b_*` – round 3; `r02a/b/c_*` – round 2 (before / after the link-depth column order / with the tree-structured TSQR); `r01n_*`, `r01_mfma_f64_peak.txt` – round 1",
    "",
]
open(os.path.join(dst, "README.md"), "w").write("\n".join(lines))
print(gram_line)
