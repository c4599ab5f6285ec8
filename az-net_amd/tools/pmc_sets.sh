#!/bin/bash
# Several counter sets on one workload (one-pass search under AZ_GEMM_MODE=<m>), one rocprofv3 pass per set.
# usage (GPU box, repo root): bash az-net_amd/tools/pmc_sets.sh <mode> <kernel substring> "<set 1>" "<set 2>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AZ_GEMM_MODE=${1:-3}; kern=$2; shift; shift
out=gpurun_out/pmcs; rm -rf $out; mkdir -p $out
args="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-level-loop --one-pass --no-extras --no-rccl --event-every 1000"
i=0
for set in "$@"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $args > /dev/null 2> $out/log$i
  python3 az-net_amd/tools/summarize_pmc.py $out/s$i.csv "$(find $out/p$i -name '*counter_collection.csv' | head -1)"
  head -1 $out/s$i.csv; grep "$kern" $out/s$i.csv
  rm -rf $out/p$i
  i=$((i+1))
done
