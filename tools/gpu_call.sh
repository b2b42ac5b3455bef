#!/bin/bash
# usage: tools/gpu_call.sh <timeout-s> <logname> '<command>'   -- retries while the pod's GPU slots are busy (exit code 3: nothing charged)
t=$1; log=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > gpurun_out/$log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
