#!/bin/bash
# round 4, even / odd tree: the eigenvalue error of the full schedule over five seeds (joint and sequential nesting),
# the bf16x3 path and the exact-Laplacian mode at seed 0, five seeds of the oscillator configuration
out=/root/repo/gpurun_out/r04s5
mkdir -p $out
cd /root/repo
for s in 0 1 2 3 4; do
  python scripts/train_hydrogen.py --seed $s --evals 500000 --out $out/train_cfg2_fp32_seed$s.json > $out/j$s.log 2>&1; tail -1 $out/j$s.log | cut -c1-120
  python scripts/train_hydrogen.py --seed $s --sequential --evals 500000 --out $out/train_cfg2_seq_seed$s.json > $out/s$s.log 2>&1; tail -1 $out/s$s.log | cut -c1-120
done
python scripts/train_hydrogen.py --seed 0 --path bf16x3 --evals 500000 --out $out/train_cfg2_bf16x3_seed0.json > $out/b0.log 2>&1; tail -1 $out/b0.log | cut -c1-120
python scripts/train_hydrogen.py --seed 0 --laplacian-eps 0 --evals 500000 --out $out/train_cfg2_exact_seed0.json > $out/e0.log 2>&1; tail -1 $out/e0.log | cut -c1-120
for s in 0 1 2 3 4; do
  python scripts/train_hydrogen.py --problem oscillator --batch-size 512 --seed $s --evals 100000 --out $out/train_osc_B512_seed$s.json > $out/o$s.log 2>&1; tail -1 $out/o$s.log | cut -c1-120
done
