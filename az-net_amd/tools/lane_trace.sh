#!/bin/bash
set -eu
# Steady-state timeline of bench.py's main loop (every kernel of a few consecutive searches, one column per HIP queue) from a
# rocprofv3 kernel trace, with any bench flags: e.g. `--lanes 1`, `--tz 0.1551` (a sparse tree), `--queue-depth 2`.
# usage (GPU box, repo root): bash az-net_amd/tools/lane_trace.sh <tag> [searches] [bench flags]  -> gpurun_out/<tag>/timeline.txt
tag=${1:-lanes}; [ $# -gt 0 ] && shift
n=${1:-4}; [ $# -gt 0 ] && shift
repo=${GRAFT_REPO_ROOT:?run on the GPU box (GRAFT_REPO_ROOT is set there)}
cd /tmp && export TMPDIR=/tmp && cd "$repo"
out="$repo/gpurun_out/$tag"; rm -rf "$out"; mkdir -p "$out"
args="bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-one-pass --no-two-pass --no-extras --no-rccl --no-sweep --no-stream --no-box --no-one-lane --event-every 1000 $*"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out"/kt -- python3 $args > "$out"/bench.json 2> "$out"/kt.log
kt=$(find "$out"/kt -name '*kernel_trace.csv' | head -1)
python3 az-net_amd/tools/lane_timeline.py "$kt" $n > "$out"/timeline.txt
rm -rf "$out"/kt
