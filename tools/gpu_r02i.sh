#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "tsqr" 2>&1 | tail -6
timeout 300 python tools/perf_probe.py tsqr 2>&1 | grep -v amdgpu | tr -d '\n ' ; echo
FBR_TSQR_TIMING=1 timeout 300 python tools/tsqr_timing_probe.py 2>&1 | grep "fbr tsqr" | tail -1
