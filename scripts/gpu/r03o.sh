#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_bench_launch.py -x -q 2>&1 | tail -25
timeout 900 python bench.py --gpus 1 --force-exchange --steps 200 --warmup 20 > gpurun_out/r03o_rccl1.json 2> gpurun_out/r03o_rccl1.err
echo "rc=$?"; tail -3 gpurun_out/r03o_rccl1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03o_rccl1.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'])
c=d['comm']; print(json.dumps({k:c[k] for k in c if k!='note'}, indent=1))
print(json.dumps(d.get('sharding_hp'), indent=1))
PY
timeout 900 python bench.py --gpus 1 --force-exchange --parallelism dp --no-extras --steps 200 --warmup 20 > gpurun_out/r03o_rccl1_hp.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03o_rccl1_hp.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'])
c=d['comm']; print(json.dumps({k:c[k] for k in c if k!='note'}, indent=1))
PY
