#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_dropin_gpu.py tests/test_hip_parity.py tests/test_multirank_gpu.py -x -q 2>&1 | tail -5
python scripts/dev/hp_rank_time.py 2>&1 | grep "N="
cd /tmp && export TMPDIR=/tmp
for N in 8; do
out=/root/repo/gpurun_out/r03q_hp$N; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 /root/repo/scripts/dev/hp_rank_prof.py $N > /dev/null 2>&1
rm -f $out/s_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/s_kernel_stats.csv")))
print("== hp rank shape N=$N")
for r in rows[:9]:
    print(f"{r['Name'][:90]:<92}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
PY
done
cd /root/repo
bash scripts/dev/ab.sh r03q_cfg2 2>&1 | tail -8
