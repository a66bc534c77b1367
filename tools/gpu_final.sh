#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02z_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02z_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r02z_bench.json 2> gpurun_out/r02z_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02z_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','n_gpus','steps','ms_per_step','scaling','value_incl_h2d')})
print(d['roofline']['frac'], d['tsqr']['seconds'], d['cpu_baseline']['value'], d['cpu_baseline_phases']['parity'])
PY
