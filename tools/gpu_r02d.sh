#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r02d_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r02d_pytest.log
timeout 900 python bench.py --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r02d_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d_bench.json'))
print('value',d['value'],'frac',d['roofline']['frac'])
print('tsqr',{k:d['tsqr'][k] for k in ('seconds','executed_TFLOP_per_s','executed_frac_of_fp64_mfma_peak','dense_model_TFLOP_per_s','kernel_ms_per_call_rank0')})
oc=d['other_configs']
print('leftarm',oc['walkman_left_arm_floating_500k'])
print('cfg5',oc['walkman_full_4M_gram_tsqr_sdp_inputs'])
print('h2d',d.get('value_incl_h2d'))
PY
