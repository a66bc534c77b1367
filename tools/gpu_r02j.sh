#!/bin/bash
export TMPDIR=/tmp
for v in cur nopsh cur nopsh; do
  if [ $v = nopsh ]; then export FBR_LIB_PATH=$PWD/tools/_build/libfbr_nopsh.so; else unset FBR_LIB_PATH; fi
  timeout 300 python tools/perf_probe.py tsqr 2>&1 | grep -v amdgpu | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', {k:(round(v['tsqr']['wall_ms'],2)) for k,v in d.items()})"
done
