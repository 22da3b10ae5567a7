#!/usr/bin/env bash
# One capture of the final build (run on the GPU box through gpurun):   bash tools/final_profiles.sh r04f
# Writes gpurun_out/<tag>_*: the default bench line, the same command under rocprofv3 --kernel-trace --stats, the two PMC
# traffic passes, the matrix-pipe PMC pass, the other BASELINE configurations and the per-shape convolution table.
# Every pass is checked before its summary is derived; a failed pass leaves no partial <tag>_* file behind.
set -euo pipefail
TAG="${1:?usage: final_profiles.sh <tag>}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p "$O"
# the counter passes see C3 training steps ONLY (no CPU leg, no VGG variant, no generator-only passes, no C5 leg): their
# per-launch averages cover exactly the launch mix of the timed region
PMC_FLAGS="--steps 3 --warmup 1 --no_c5_leg --no_vgg_variant --no_cpu_baseline --no_gen_metric"
export PMC_FLAGS_NOTE="bench.py $PMC_FLAGS"

first_csv() {   # first file matching a pattern under a directory, or fail loudly
  local f
  f=$(find "$1" -name "$2" | head -1)
  [ -n "$f" ] && [ -s "$f" ] || { echo "final_profiles: no $2 under $1" >&2; exit 1; }
  echo "$f"
}

# the driver's exact command (BENCH_r*.json "cmd"): the parity gate is validated at THESE flags
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_bench.json.tmp" 2> "$O/${TAG}_bench.err"
mv "$O/${TAG}_bench.json.tmp" "$O/${TAG}_bench.json"

rocprofv3 --kernel-trace --stats --output-format csv -d "$O/p_stats" -- python3 bench.py --steps 10 --warmup 2 --no_c5_leg \
  > "$O/${TAG}_bench_under_rocprof.json.tmp" 2> "$O/p_err1.txt"
cp "$(first_csv "$O/p_stats" '*kernel_stats.csv')" "$O/${TAG}_kernel_stats.csv"
mv "$O/${TAG}_bench_under_rocprof.json.tmp" "$O/${TAG}_bench_under_rocprof.json"
python3 tools/trace_gaps.py "$O/p_stats" 6 > "$O/${TAG}_trace_gaps.txt" || true
rm -rf "$O/p_stats"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 bench.py $PMC_FLAGS \
  > /dev/null 2> "$O/p_err2.txt"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 bench.py $PMC_FLAGS \
  > /dev/null 2> "$O/p_err3.txt"
first_csv "$O/pmc_fetch" '*counter_collection.csv' > /dev/null
first_csv "$O/pmc_write" '*counter_collection.csv' > /dev/null
python3 tools/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" "$TAG" > "$O/${TAG}_pmc_traffic.json.tmp"
mv "$O/${TAG}_pmc_traffic.json.tmp" "$O/${TAG}_pmc_traffic.json"
rm -rf "$O/pmc_fetch" "$O/pmc_write"

rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_mfma" -- python3 bench.py $PMC_FLAGS > /dev/null 2> "$O/p_err4.txt"
first_csv "$O/pmc_mfma" '*counter_collection.csv' > /dev/null
python3 tools/pmc_summarize.py "$O/pmc_mfma" > "$O/${TAG}_pmc_mfma.txt.tmp"
mv "$O/${TAG}_pmc_mfma.txt.tmp" "$O/${TAG}_pmc_mfma.txt"
rm -rf "$O/pmc_mfma"

python3 bench.py --config C2 --image_size 128 --no_cpu_baseline --no_vgg_variant > "$O/${TAG}_bench_C2.json" 2> /dev/null
python3 bench.py --config C4 --batch 4 --no_cpu_baseline --no_vgg_variant > "$O/${TAG}_bench_C4.json" 2> /dev/null
python3 bench.py --config C5 --batch 6 --no_cpu_baseline --no_vgg_variant \
  > "$O/${TAG}_bench_C5_dense_graphs.json" 2> /dev/null
python3 tools/conv_shapes.py > "$O/${TAG}_conv_shapes.txt" 2> /dev/null
python3 tools/wgrad_bench.py > "$O/${TAG}_wgrad_bench.txt" 2> /dev/null
python3 tools/wino4_ab.py 2> /dev/null | grep -v amdgpu > "$O/${TAG}_wino4_ab.txt" || true
python3 tools/wgrad_ablate.py 2> /dev/null | grep -v amdgpu > "$O/${TAG}_wgrad_ablate.txt" || true
python3 tools/graph_timing.py C4 2> /dev/null | grep '^C4' > "$O/${TAG}_graph_timing_C4.txt" || true

tail -c 400 "$O/${TAG}_bench_under_rocprof.json"; head -c 600 "$O/${TAG}_pmc_traffic.json"; head -20 "$O/${TAG}_pmc_mfma.txt"
