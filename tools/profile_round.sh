#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>        e.g. r01d   -> gpurun_out/prof_<tag>/...  (summarise with tools/summarize_profiles.py)
# Kernel-trace/stats and every PMC set are separate runs (no sys/hip/hsa tracing together with --pmc).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
B="python $REPO/bench.py"
# 1. the default bench command under the kernel trace (per-kernel average durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o bench -- $B --no-cpu-baseline > "$REPO/$OUT/bench_rocprof_stdout.txt" 2>&1
# 2. the same command without the profiler (the bench line the summary quotes)
$B > "$REPO/$OUT/bench_default.json" 2> "$REPO/$OUT/bench_default.err"
# 2b. secondary figures of the other BASELINE configs (KUKA 50 k, left arm 500 k)
$B --other-configs --no-cpu-baseline > "$REPO/$OUT/bench_other_configs.json" 2>/dev/null
# 3. HBM traffic counters, one pass each, on a shorter run of the same workload
S="--samples 200000 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_fetch -- $B $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_write -- $B $S > /dev/null 2>&1
# 4. MFMA / issue counters
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_mfma -- $B $S > "$REPO/$OUT/pmc_mfma_stdout.txt" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_sq -- $B $S > "$REPO/$OUT/pmc_sq_stdout.txt" 2>&1
ls -la "$REPO/$OUT"
