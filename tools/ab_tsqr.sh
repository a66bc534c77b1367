#!/bin/bash
# A/B of TSQR kernel builds: tools/ab_tsqr.sh <S> lib1.so lib2.so ...   (libs under tools/_build/)
export TMPDIR=/tmp
S=${1:-300000}; shift
for lib in "$@"; do
  echo "== $lib"
  FBR_LIB_PATH=$PWD/tools/_build/$lib timeout 600 python tools/tsqr_probe.py $S 3 2>&1 | grep -v amdgpu.ids | cut -c1-200
  FBR_LIB_PATH=$PWD/tools/_build/$lib timeout 300 python tools/tsqr_timing_probe.py 2>&1 | grep "tsqr timing" | head -2 | cut -c1-330
done
