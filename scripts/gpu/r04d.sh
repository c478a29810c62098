#!/bin/bash
# round 4: what the bf16x3 K loop spends its cycles on - stamped diagnostic builds with parts of the loop left out
out=/root/repo/gpurun_out/r04d
mkdir -p $out
cd /root/repo
for v in ${VARIANTS:-0 2}; do
  EXTRA="-DNSVD_BF3_EXP=$v" SUF=_bf3exp$v bash scripts/dev/build_stamps.sh > $out/build_$v.log 2>&1 || { echo "build $v failed"; tail -5 $out/build_$v.log; continue; }
  echo "== NSVD_BF3_EXP=$v"
  NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_stamps_bf3exp$v.so NSVD_DEV_PATH=3 timeout 120 python scripts/dev/stamps.py 2>&1 | grep -vE "libdrm"
done
