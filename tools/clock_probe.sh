#!/bin/bash
# GPU clock / power while the fused pass runs (is the pass power-limited?): tools/clock_probe.sh [env assignments for bench.py]
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Average Graphics Package Power|Socket Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clocks_$1.txt &
SMI=$!
env $2 python bench.py --no-secondary --no-cpu-baseline --steps 300 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k:round(x,1) for k,x in d['kernel_ms_per_step'].items()})"
wait $SMI
sort gpurun_out/clocks_$1.txt | uniq -c | sort -rn | head -6
