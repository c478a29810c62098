#!/bin/bash
# round 4: bf16x3 (centre + perturbation form) - accuracy record, eigenvalue parity, the full schedule, full GPU suite
out=/root/repo/gpurun_out/r04i
mkdir -p $out
cd /root/repo
timeout 300 python scripts/dev/bf3_check.py > $out/bf3_check.log 2>&1; echo "bf3_check rc=$?"; grep -E "RECORD|forward" $out/bf3_check.log | cut -c1-400
timeout 2700 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $out/pytest_all.log
timeout 600 python scripts/parity_spectrum_cfg2.py --steps 20000 --out $out/parity_spectrum_cfg2.json > $out/parity.log 2>&1; echo "parity rc=$?"; tail -4 $out/parity.log
timeout 600 python scripts/train_hydrogen.py --path bf16x3 --out $out/train_cfg2_bf16x3.json > $out/train_bf16x3.log 2>&1; echo "train rc=$?"; tail -1 $out/train_bf16x3.log | cut -c1-300
