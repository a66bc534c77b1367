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
gk = [k for k in pmc.get("FETCH_SIZE", {}) if k.startswith("fbr_gram_kernel")]
if gk:
    g = gk[0]
    nl = pmc["FETCH_SIZE"][g]["launches"]
    spl = S / nl
    fetch_kb = pmc["FETCH_SIZE"][g]["per_launch"]
    write_kb = pmc.get("WRITE_SIZE", {}).get(g, {}).get("per_launch", 0.0)
    out = {"command": "tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE passes, 200000 samples)", "samples_per_gram_launch": spl,
           "gram_kernel": {"fetch_KB_per_launch_raw": fetch_kb, "write_KB_per_launch_raw": write_kb,
                           "fetch_bytes_per_sample_corrected": 2.0 * fetch_kb * 1024 / spl,
                           "write_bytes_per_sample": write_kb * 1024 / spl},
           "hbm_bytes_per_sample": 2.0 * fetch_kb * 1024 / spl + write_kb * 1024 / spl,
           "correction": "FETCH_SIZE x2 on gfx950 for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported"}
    json.dump(out, open(os.path.join(dst, tag + "_gram_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
print("bench:", bench["value"], bench["roofline"]["avg_launch_ms"], bench.get("tsqr", {}).get("executed_TFLOP_per_s"))
