#!/bin/bash
# Kernel trace + stats of bench.py's main loop with two lanes and with one (profiles/<tag>_lanes2_*, <tag>_lanes1_*).
# usage (GPU box, repo root): bash az-net_amd/tools/kt_lanes.sh <tag>
set -u
tag=${1:-prof}
repo=$(pwd)
out=$repo/gpurun_out/$tag
tools=$repo/az-net_amd/tools
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
args="--steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-one-pass --no-two-pass --no-extras --no-rccl --no-sweep --no-stream --no-box --no-one-lane --event-every 1000"
for lanes in 2 1; do
  d=$out/l$lanes
  mkdir -p "$d"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/kt" -- python3 bench.py $args --lanes $lanes > "$out/lanes${lanes}_bench_under_profiler.json" 2> "$d/kt.log"
  ks=$(find "$d/kt" -name '*kernel_stats.csv' | head -1)
  kt=$(find "$d/kt" -name '*kernel_trace.csv' | head -1)
  python3 "$tools/summarize_prof.py" "$ks" "$out/lanes${lanes}_kernel_stats.csv" "rocprofv3 --kernel-trace --stats -- python3 bench.py $args --lanes $lanes"
  python3 "$tools/lane_timeline.py" "$kt" > "$out/lanes${lanes}_timeline.txt" 2>&1
  rm -rf "$d"
done
head -12 "$out"/lanes2_kernel_stats.csv "$out"/lanes1_kernel_stats.csv
