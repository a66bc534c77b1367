#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r02c_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02c_pytest.log
timeout 300 python tools/tsqr_timing_probe.py > gpurun_out/r02c_tsqr_timing.txt 2>&1; grep "fbr tsqr" gpurun_out/r02c_tsqr_timing.txt
timeout 300 python tools/perf_probe.py tsqr > gpurun_out/r02c_perf_tsqr.json 2>&1; cat gpurun_out/r02c_perf_tsqr.json | tr -d '\n ' ; echo
