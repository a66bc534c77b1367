#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r02b_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02b_pytest.log
timeout 300 python tools/tsqr_timing_probe.py > gpurun_out/r02b_tsqr_timing.txt 2>&1; grep "fbr tsqr" gpurun_out/r02b_tsqr_timing.txt
