#!/bin/bash
# tools/ab_run.sh <script.py> <arg> lib1.so lib2.so ...  -- runs the script once per library build under tools/_build/
export TMPDIR=/tmp
SCR=$1; ARG=$2; shift; shift
for lib in "$@"; do
  echo "== $lib"
  FBR_LIB_PATH=$PWD/tools/_build/$lib timeout 900 python $SCR $ARG 2>&1 | grep -v amdgpu.ids | cut -c1-400
done
