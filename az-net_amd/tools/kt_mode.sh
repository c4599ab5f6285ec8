#!/bin/bash
# Kernel stats of the level-loop search under AZ_GEMM_MODE=<m>.  usage (GPU box, repo root): bash az-net_amd/tools/kt_mode.sh <mode> [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AZ_GEMM_MODE=${1:-2}
out=gpurun_out/${2:-kt_mode$1}; mkdir -p $out
args="bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-one-pass --no-extras --no-rccl --event-every 1000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $args > $out/bench.json 2> $out/kt.log
ks=$(find $out/kt -name '*kernel_stats.csv' | head -1)
python3 az-net_amd/tools/summarize_prof.py "$ks" $out/kernel_stats.csv "level loop, AZ_GEMM_MODE=$AZ_GEMM_MODE"
rm -rf $out/kt
head -30 $out/kernel_stats.csv
python3 -c "
import json,sys
d=json.loads(open('$out/bench.json').readline())
print(d['ms_per_step'], d['config']['workload'][:200])"
