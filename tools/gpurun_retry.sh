#!/usr/bin/env bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged):  tools/gpurun_retry.sh <timeout_s> '<command>'
T="$1"; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ "$rc" != 3 ] && exit "$rc"
  sleep 60
done
exit 3
