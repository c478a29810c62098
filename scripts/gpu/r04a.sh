#!/bin/bash
# round 4, first contact: the new tests (graph replay, per-step loss, configs[2] at its global batch, trajectory,
# bottleneck towers), a short bench line with the graph mode, the eigenvalue-parity records
out=/root/repo/gpurun_out/r04a
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_tower_gpu.py -x -q > $out/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -5 $out/pytest_new.log
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py -x -q > $out/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -5 $out/pytest_parity.log
timeout 600 python bench.py --steps 200 --warmup 20 --accuracy off --no-cpu-baseline --no-extras > $out/bench_short.json 2> $out/bench_short.err; echo "bench rc=$?"; tail -c 600 $out/bench_short.err
python -c "
import json;d=json.load(open('$out/bench_short.json'));print(d['value'],d['timing']['mode']);print(json.dumps(d['timing']['modes'],indent=1)[:1500])"
timeout 900 python scripts/parity_spectrum_cfg2.py --steps 20000 --out $out/parity_spectrum_cfg2.json > $out/parity.log 2>&1; echo "parity rc=$?"; tail -4 $out/parity.log
