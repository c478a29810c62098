#!/bin/bash
# round 4: sixth-order Taylor softplus on the (centre, even, odd) triple - accuracy at large perturbations, parity, kernel time
out=/root/repo/gpurun_out/r04n
mkdir -p $out
cd /root/repo
for a in "cfg2 1" "cfg3 1" "cfg3 4"; do
  timeout 300 python scripts/dev/bf3_check.py $a > "$out/bf3_check_${a// /_}.log" 2>&1; echo "bf3_check $a rc=$?"; grep -E "rel err" "$out/bf3_check_${a// /_}.log"
done
timeout 600 python scripts/parity_spectrum_cfg2.py --steps 20000 --out $out/parity_spectrum_cfg2.json > $out/parity.log 2>&1; echo "parity rc=$?"; tail -4 $out/parity.log
bash scripts/gpu/r04j.sh
