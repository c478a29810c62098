#!/bin/bash
# round 4: even / odd form on both fused paths - full GPU suite, eigenvalue parity at configs[1], in-run accuracy (native)
out=/root/repo/gpurun_out/r04l
mkdir -p $out
cd /root/repo
timeout 2700 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $out/pytest_all.log
timeout 600 python scripts/parity_spectrum_cfg2.py --steps 20000 --out $out/parity_spectrum_cfg2.json > $out/parity.log 2>&1; echo "parity rc=$?"; tail -4 $out/parity.log
timeout 600 python scripts/train_hydrogen.py --out $out/train_cfg2_fp32.json > $out/train_fp32.log 2>&1; echo "train rc=$?"; tail -1 $out/train_fp32.log | cut -c1-400
