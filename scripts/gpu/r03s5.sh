#!/bin/bash
# round 3, final tree: the eigenvalue error of the full schedule over five seeds (joint and sequential nesting), and
# ten seeds of the oscillator configuration at the reference script's iteration count
out=/root/repo/gpurun_out/r03s5
mkdir -p $out
cd /root/repo
for s in 1 2 3 4; do
  python scripts/train_hydrogen.py --seed $s --evals 500000 --out $out/train_cfg2_fp32_seed$s.json > $out/j$s.log 2>&1; tail -1 $out/j$s.log | cut -c1-120
  python scripts/train_hydrogen.py --seed $s --sequential --evals 500000 --out $out/train_cfg2_seq_seed$s.json > $out/s$s.log 2>&1; tail -1 $out/s$s.log | cut -c1-120
done
python scripts/train_hydrogen.py --seed 0 --sequential --evals 500000 --out $out/train_cfg2_seq_seed0.json > $out/s0.log 2>&1; tail -1 $out/s0.log | cut -c1-120
for s in 0 1 2 3 4 5 6 7 8 9; do
  python scripts/train_hydrogen.py --problem oscillator --batch-size 512 --seed $s --evals 100000 --out $out/train_osc_B512_seed$s.json > $out/o$s.log 2>&1; tail -1 $out/o$s.log | cut -c1-120
done
