#!/bin/bash
set -eu
# Kernel timeline of tools/test_shared.py's image loop (BASELINE config 3: proposals + Fast R-CNN head on the shared map): the
# kernels of two consecutive images from the middle of the run, one column per HIP queue.
# usage (GPU box, repo root): bash az-net_amd/tools/shared_trace.sh <tag> [images=12]   -> gpurun_out/<tag>/timeline.txt
tag=${1:-shared}; n=${2:-12}
repo=${GRAFT_REPO_ROOT:?run on the GPU box (GRAFT_REPO_ROOT is set there)}
cd /tmp && export TMPDIR=/tmp && cd "$repo"
out="$repo/gpurun_out/$tag"; rm -rf "$out"; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out"/kt -- python3 az-net_amd/tools/test_shared.py --gpu 0 \
    --net_az synthetic --net_frcnn synthetic --imdb synthetic_600x1000_$n --tz 0.0 --exp trace_$tag > "$out"/cli.log 2> "$out"/kt.log
kt=$(find "$out"/kt -name '*kernel_trace.csv' | head -1)
python3 az-net_amd/tools/lane_timeline.py "$kt" 2 > "$out"/timeline.txt
rm -rf "$out"/kt "$repo/az-net_amd/output/trace_$tag" "$repo/output/trace_$tag"
tail -3 "$out"/cli.log
