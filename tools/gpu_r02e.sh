#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for tag in reorder noreorder reorder2; do
  if [ $tag = noreorder ]; then export FBR_TSQR_NO_REORDER=1; else unset FBR_TSQR_NO_REORDER; fi
  timeout 900 python bench.py --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r02e_$tag.json 2> gpurun_out/r02e_$tag.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r02e_$tag.json'))
oc=d['other_configs']['walkman_full_4M_gram_tsqr_sdp_inputs']
print('$tag','value',round(d['value']),'tsqr1M',round(d['tsqr']['seconds'],4),'cfg5 gram',round(oc['gram_seconds'],4),'cfg5 tsqr',round(oc['tsqr_seconds'],4),'h2d',round(d['value_incl_h2d']), 'leftarm', round(d['other_configs']['walkman_left_arm_floating_500k']['tsqr_ms'],3))
PY
done
