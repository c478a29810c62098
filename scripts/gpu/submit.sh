#!/bin/bash
# build-container side: submit a GPU-box call, retrying while no slot / box is free (gpurun exit code 3: nothing charged)
#   scripts/gpu/submit.sh <timeout seconds> <command ...>
t=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" /tmp/gpurun_last.log; then
    tail -40 /tmp/gpurun_last.log
    exit $rc
  fi
  sleep 90
done
echo "no GPU slot after 40 attempts"; exit 3
