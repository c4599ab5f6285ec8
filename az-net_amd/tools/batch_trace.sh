#!/bin/bash
set -eu
# Timeline of lockstep batches (az_batch_launch) on the planted-object set at its tuned threshold: every kernel of a few
# consecutive batches, one column per HIP queue, from a rocprofv3 kernel trace of tests/dev/stream_probe.py.
# usage (GPU box, repo root): bash az-net_amd/tools/batch_trace.sh <tag> [batches] [batch size] [lanes]  -> gpurun_out/<tag>/timeline.txt
tag=${1:-batch}; n=${2:-2}; bs=${3:-8}; lanes=${4:-1}
repo=${GRAFT_REPO_ROOT:?run on the GPU box (GRAFT_REPO_ROOT is set there)}
cd /tmp && export TMPDIR=/tmp && cd "$repo"
out="$repo/gpurun_out/$tag"; rm -rf "$out"; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out"/kt -- python3 tests/dev/stream_probe.py --objects --anchors 20 --batch "$bs" --lanes "$lanes" --passes 2 --replay 0 --batch-only > "$out"/probe.txt 2> "$out"/kt.log
kt=$(find "$out"/kt -name '*kernel_trace.csv' | head -1)
python3 az-net_amd/tools/lane_timeline.py "$kt" "$n" k_final_select_b > "$out"/timeline.txt
ks=$(find "$out"/kt -name '*kernel_stats.csv' | head -1)
[ -n "$ks" ] && cp "$ks" "$out"/kernel_stats.csv
rm -rf "$out"/kt
