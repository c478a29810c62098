#!/bin/bash
# round 4: torch binding (bit-identity test, host cost per call), bench launch tests
out=/root/repo/gpurun_out/r04e
mkdir -p $out
cd /root/repo
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q > $out/pytest_graph.log 2>&1; echo "pytest graph rc=$?"; tail -3 $out/pytest_graph.log
timeout 300 python scripts/dev/binding_cost.py > $out/binding_cost.log 2>&1; echo "binding_cost rc=$?"; grep -E "ctypes|torch|RECORD" $out/binding_cost.log
timeout 1500 python -m pytest tests/test_bench_launch.py -x -q -m gpu > $out/pytest_bench_launch.log 2>&1; echo "pytest bench_launch rc=$?"; tail -5 $out/pytest_bench_launch.log
