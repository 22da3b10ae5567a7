#!/usr/bin/env bash
# The two HBM-traffic counter passes of tools/final_profiles.sh alone + the algorithmic model of the same launches:
#   bash tools/pmc_only.sh <tag>   ->  gpurun_out/<tag>_pmc_traffic.json, gpurun_out/<tag>_traffic_model.txt
set -euo pipefail
TAG="${1:?usage: pmc_only.sh <tag>}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p "$O"
PMC_FLAGS="--steps 3 --warmup 1 --no_c5_leg --no_vgg_variant --no_cpu_baseline --no_gen_metric"
export PMC_FLAGS_NOTE="bench.py $PMC_FLAGS"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 bench.py $PMC_FLAGS > /dev/null 2> "$O/p_err2.txt"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 bench.py $PMC_FLAGS > /dev/null 2> "$O/p_err3.txt"
python3 tools/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" "$TAG" > "$O/${TAG}_pmc_traffic.json"
rm -rf "$O/pmc_fetch" "$O/pmc_write"
cp "$O/${TAG}_pmc_traffic.json" profiles/pmc_traffic.json
python3 tools/wino4_traffic_model.py 2> /dev/null | grep -v amdgpu > "$O/${TAG}_traffic_model.txt"
cat "$O/${TAG}_traffic_model.txt"
