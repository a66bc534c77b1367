#!/bin/bash
export TMPDIR=/tmp
for mode in overlap serial; do
  if [ $mode = serial ]; then export FBR_GRAM_SERIAL=1; else unset FBR_GRAM_SERIAL; fi
  timeout 300 python bench.py --no-secondary --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$mode', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
