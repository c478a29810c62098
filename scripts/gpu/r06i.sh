#!/bin/bash
out=/root/repo/gpurun_out/r06i
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests/test_tower_gpu.py tests/test_cdk_step_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout 300 python scripts/dev/tcol_blocks.py 2>&1 | grep -v amdgpu.ids | tail -4
for i in 1 2; do
timeout 300 python bench.py --config cfg5 --amp --no-cpu-baseline > $out/b.json 2> $out/b.err
python -c "
import json; d = json.load(open('$out/b.json')); r = d['roofline']; print('cfg5 amp', d['value'], d['ms_per_step'], r['kernel_avg_us'])"
done
