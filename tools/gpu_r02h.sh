#!/bin/bash
export TMPDIR=/tmp
timeout 200 python tools/chunk_probe.py 1000000 2>&1 | grep "^S=" | sed "s/^/default(6x3) /"
for v in 3x6 4x4 5x3 6x2; do FBR_LIB_PATH=$PWD/tools/_build/libfbr_$v.so timeout 200 python tools/chunk_probe.py 1000000 2>&1 | grep "^S=\|rror" | sed "s/^/$v /"; done
