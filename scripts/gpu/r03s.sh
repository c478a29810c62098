#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03s_bench_driver.json 2> gpurun_out/r03s_bench_driver.err; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03s_bench_driver.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['config']['parallelism'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('opt_in_path_bf16x3',{}).get('value'))
PY
for c in cfg1 cfg3; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
