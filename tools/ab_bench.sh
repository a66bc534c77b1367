#!/bin/bash
# tools/ab_bench.sh lib1.so lib2.so ... : the timed steps of bench.py (1 M samples) once per library build under tools/_build/
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["avg_launch_ms"],4), d["kernel_ms_per_step"], d["gram_checksum"]["fro"])'
for lib in "$@"; do
  echo "== $lib"
  for i in 1 2; do FBR_LIB_PATH=$PWD/tools/_build/$lib timeout 600 python bench.py --steps 20 --warmup 3 --no-secondary 2>/dev/null | python -c "$P"; done
done
