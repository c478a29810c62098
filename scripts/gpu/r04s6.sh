#!/bin/bash
# round 4: four more seeds of the full schedule on the bf16x3 path (seed 0 is in r04s5)
out=/root/repo/gpurun_out/r04s5
mkdir -p $out
cd /root/repo
for s in 1 2 3 4; do
  python scripts/train_hydrogen.py --seed $s --path bf16x3 --evals 500000 --out $out/train_cfg2_bf16x3_seed$s.json > $out/b$s.log 2>&1; tail -1 $out/b$s.log | cut -c1-120
done
