#!/bin/bash
# round 4: captured plain loop (tests + timing), the whole GPU suite
out=/root/repo/gpurun_out/r04b
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_dropin_gpu.py -x -q > $out/pytest_dropin.log 2>&1; echo "pytest dropin rc=$?"; tail -5 $out/pytest_dropin.log
timeout 600 python scripts/dev/dropin_time.py > $out/dropin_time.log 2>&1; echo "dropin_time rc=$?"; grep -E "steps/s|RECORD" $out/dropin_time.log
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $out/pytest_all.log
