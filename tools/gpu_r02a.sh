#!/bin/bash
# round 2, first GPU checkpoint: full -m gpu suite, default bench line, TSQR phase timing, chunk-count sweep for short batches
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02a_pytest.log
timeout 900 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r02a_bench.err
timeout 300 python tools/tsqr_timing_probe.py > gpurun_out/r02a_tsqr_timing.txt 2>&1; tail -5 gpurun_out/r02a_tsqr_timing.txt
for mc in 1 2 4 8 16; do FBR_MIN_CHUNKS=$mc timeout 200 python tools/chunk_probe.py 125000; done > gpurun_out/r02a_chunks.txt 2>&1
for mc in 1 8 16; do FBR_MIN_CHUNKS=$mc timeout 200 python tools/chunk_probe.py 250000; done >> gpurun_out/r02a_chunks.txt 2>&1
FBR_MIN_CHUNKS=8 timeout 200 python tools/chunk_probe.py 1000000 >> gpurun_out/r02a_chunks.txt 2>&1
cat gpurun_out/r02a_chunks.txt
