#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>        e.g. r02a   -> gpurun_out/prof_<tag>/...  (summarise with tools/summarize_profiles.py <tag>)
# Kernel-trace/stats and every PMC set are separate runs (no sys/hip/hsa tracing together with --pmc).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
B="python $REPO/bench.py"
# 1. the default bench command (all legs but the CPU baselines and the 10 s sustained loop) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o bench -- $B --no-cpu-baseline --sustain-seconds 0 > "$REPO/$OUT/bench_rocprof_stdout.txt" 2>&1
# 2. the same command without the profiler, CPU baselines included (the bench line the summary quotes)
$B > "$REPO/$OUT/bench_default.json" 2> "$REPO/$OUT/bench_default.err"
# 3. HBM traffic counters of the fused pass, one pass each, on a shorter run of the same workload
S="--samples 200000 --steps 1 --warmup 0 --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_fetch -- $B $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_write -- $B $S > /dev/null 2>&1
# 4. MFMA / issue counters of the fused pass
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_mfma -- $B $S > "$REPO/$OUT/pmc_mfma_stdout.txt" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$REPO/$OUT" -o pmc_sq -- $B $S > "$REPO/$OUT/pmc_sq_stdout.txt" 2>&1
# 5. the same two counter sets on the Householder TSQR (WALK-MAN 150 k samples x 481 columns, left arm 500 k x 91)
T="python $REPO/tools/tsqr_pmc_probe.py"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$REPO/$OUT" -o tsqr_pmc_mfma -- $T > "$REPO/$OUT/tsqr_pmc_stdout.txt" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$REPO/$OUT" -o tsqr_pmc_sq -- $T > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT" -o tsqr_pmc_fetch -- $T > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o tsqr -- $T > /dev/null 2>&1
# 6. the whole TSQR call of the bench (1 M samples) and of one rank's shard at 8 GPUs (125 k) under the kernel trace: the per-kernel split
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o tsqr_full -- python $REPO/tools/tsqr_probe.py 1000000 3 > "$REPO/$OUT/tsqr_full_stdout.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o tsqr_shard -- python $REPO/tools/tsqr_probe.py 125000 5 > "$REPO/$OUT/tsqr_shard_stdout.txt" 2>&1
ls -la "$REPO/$OUT"
