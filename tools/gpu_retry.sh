#!/usr/bin/env bash
# gpurun with retries while the pool is busy (exit code 3 = nothing charged):  tools/gpu_retry.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
