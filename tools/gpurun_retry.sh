#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout-seconds> <script> : gpurun with retries while the pod's GPU slots are busy
T=$1; S=$2
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $T -- "bash $S" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.log; then sleep 90; continue; fi
  break
done
cp /tmp/gpurun_last.log /tmp/gpurun_done.log
