#!/bin/bash
# round 6: the K-split forward with its finite-difference epilogue folded into the second launch (arrival tickets)
out=/root/repo/gpurun_out/r06g
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "k_split or split_stencil or headline or small" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
for f in 1 0 1 0; do
  NSVD_KSPLIT_FOLD=$f python bench.py --config cfg1 --accuracy off --no-extras --no-cpu-baseline --graph off --steps 1000 --warmup 100 --repeats 7 > $out/b.json 2> $out/b.err
  python -c "
import json; d = json.load(open('$out/b.json')); print('fold=$f', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us'])"
done
