#!/usr/bin/env bash
# Multi-GPU bring-up check for an 8-GPU MI355X node (not runnable on the 1-GPU boxes this repository is developed on):
# runs the benchmark over RCCL with NCCL_DEBUG=INFO and prints, per collective size class, which algorithm / protocol /
# channel count RCCL picked over xGMI, the bucket plan of the trainer and the per-N throughput lines.
#   tools/rccl_check.sh [NGPUS]        (default 8)
set -euo pipefail
N="${1:-8}"
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,GRAPH NCCL_DEBUG_FILE=/tmp/rccl_%h_%p.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus "$N" --steps 6 --warmup 2 --no_cpu_baseline --no_vgg_variant | tee /tmp/rccl_bench.json   # eager path (default at N > 1)
echo "---- topology / rings / trees"
grep -h -E "Channel|Ring|Tree|xGMI|XGMI|comm 0x.* rank 0 " /tmp/rccl_*.log | sort | uniq -c | sort -rn | head -40
echo "---- collectives issued by rank 0 (count by size)"
grep -h "AllReduce" /tmp/rccl_*.log | sed -E 's/.*count ([0-9]+).*datatype ([0-9]+).*/\1 elements type \2/' | sort | uniq -c | sort -rn | head -20
echo "---- expected: 6 fp32 all-reduces of <= 16 Mi elements (generator buckets; the last one 4 floats longer: the late-gradient"
echo "     flag), 1-2 small ones per discriminator, ~36 x 2 fp64 all-reduces of 2C elements (SyncBN forward / backward) — these"
echo "     and an all-gather only with --learned_converse.  Same step replayed from HIP graphs (opt-in: the SyncBN messages are"
echo "     then issued from inside the graphs, the bucket all-reduces around them), for the A/B:"
CSG_GRAPHS_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus "$N" --steps 6 --warmup 2 --no_cpu_baseline --no_vgg_variant | tee /tmp/rccl_bench_graphs.json
