#!/bin/bash
export TMPDIR=/tmp
for S in 125000 1000000; do
for fd in 1 2 4 8; do FBR_FIRST_CHUNK_DIV=$fd timeout 200 python tools/chunk_probe.py $S 2>&1 | grep "^S=" | sed "s/^/first_div=$fd /"; done
done
for mc in 2 4 6; do FBR_FIRST_CHUNK_DIV=4 FBR_MIN_CHUNKS=$mc timeout 200 python tools/chunk_probe.py 125000 2>&1 | grep "^S=" | sed "s/^/first_div=4 /"; done
