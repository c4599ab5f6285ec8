#!/bin/bash
set -eu
# Kernel timeline of the CLI's image loop (tools/prop_az.py over a synthetic imdb): the kernels of two consecutive images from
# the middle of the run, one column per HIP queue -- where the GPU waits between the backbone and the search.
# usage (GPU box, repo root): bash az-net_amd/tools/cli_trace.sh <tag> [images=24] [prop_az.py flags, e.g. --tune-backbone]   -> gpurun_out/<tag>/timeline.txt
tag=${1:-cli}; n=${2:-24}; [ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
repo=${GRAFT_REPO_ROOT:?run on the GPU box (GRAFT_REPO_ROOT is set there)}
cd /tmp && export TMPDIR=/tmp && cd "$repo"
out="$repo/gpurun_out/$tag"; rm -rf "$out"; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out"/kt -- python3 az-net_amd/tools/prop_az.py --gpu 0 --net synthetic \
    --imdb synthetic_600x1000_$n --tz 0.0 --exp trace_$tag --def x.prototxt --def_fc y.prototxt "$@" > "$out"/cli.log 2> "$out"/kt.log
kt=$(find "$out"/kt -name '*kernel_trace.csv' | head -1)
python3 az-net_amd/tools/lane_timeline.py "$kt" 2 > "$out"/timeline.txt
rm -rf "$out"/kt "$repo/az-net_amd/output/trace_$tag" "$repo/output/trace_$tag"
tail -3 "$out"/cli.log
