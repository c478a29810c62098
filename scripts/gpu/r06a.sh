#!/bin/bash
# round 6, first contact: the new tests (independent f1 / f2, uneven head blocks, the scripts' head counts, hp with
# heads that do not divide) on the GPU
out=/root/repo/gpurun_out/r06a
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "independent or gather_head or reference_scripts or evd_loss" > $out/pytest_a.log 2>&1; echo "a rc=$?"; tail -3 $out/pytest_a.log
timeout 1500 python -m pytest tests/test_multirank_gpu.py -m gpu -q -x -k "hp" > $out/pytest_b.log 2>&1; echo "b rc=$?"; tail -3 $out/pytest_b.log
