#!/bin/bash
# FETCH_SIZE of the many-row GEMM under an environment switch (quick A/B of HBM-side traffic).
# usage (GPU box, repo root): [ENV=...] bash az-net_amd/tools/pmc_quick.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcq; rm -rf $out; mkdir -p $out
args="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-one-pass --no-extras --no-rccl --event-every 1000"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p -- python3 $args > /dev/null 2> $out/log
cf=$(find $out/p -name '*counter_collection.csv' | head -1)
python3 az-net_amd/tools/summarize_pmc.py $out/s.csv "$cf"
grep -E "kernel|k_fc_splitk12" $out/s.csv
rm -rf $out/p
