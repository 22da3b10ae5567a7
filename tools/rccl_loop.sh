#!/usr/bin/env bash
# N > 1 hardening on a 1-GPU box: the one-rank RCCL test (every data-parallel exchange issued on RCCL, HIP-graph captures with
# a process group up, a forced bucket rebuild between replays) N times in fresh processes.  A capture that meets the
# ProcessGroupNCCL watchdog aborts the process, so a single failure in the loop is the race (graphs._quiesce_before_capture).
#   bash tools/rccl_loop.sh [N=10]  ->  gpurun_out/rccl_loop.txt
N="${1:-10}"
O=gpurun_out/rccl_loop.txt
mkdir -p gpurun_out
: > "$O"
ok=0
for i in $(seq 1 "$N"); do
  if python3 -m pytest tests/test_gpu_graphs.py -x -q -m gpu -k one_rank > gpurun_out/rccl_loop_last.txt 2>&1; then
    ok=$((ok + 1)); echo "run $i: pass" >> "$O"
  else
    echo "run $i: FAIL" >> "$O"; grep -m3 "Error\|invalidated\|Captured" gpurun_out/rccl_loop_last.txt >> "$O"
  fi
done
echo "$ok of $N runs passed" >> "$O"
tail -3 "$O"
