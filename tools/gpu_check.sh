#!/bin/bash
# gpurun -- tools/gpu_check.sh <tag>: the -m gpu suite, smoke() and the default bench line into gpurun_out/<tag>_*
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
grep -h "pivots differing" gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_bench.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','n_gpus','steps','ms_per_step','scaling','value_incl_h2d','value_resident')})
print(d['roofline']['frac'], d['tsqr'].get('seconds'), d['tsqr'].get('executed_frac_of_fp64_mfma_peak'), d['cpu_baseline']['value'], d['cpu_baseline_phases']['parity'])
print(json.dumps(d.get('kernel_ms_per_step')))
PY
