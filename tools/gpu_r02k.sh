#!/bin/bash
export TMPDIR=/tmp
for v in default prio1 default prio1; do
  if [ $v = default ]; then unset FBR_LIB_PATH; else export FBR_LIB_PATH=$PWD/tools/_build/libfbr_$v.so; fi
  timeout 200 python tools/chunk_probe.py 1000000 2>&1 | grep "^S=" | sed "s/^/$v /"
done
