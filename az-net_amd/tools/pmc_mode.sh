#!/bin/bash
# Counters of the int6 GEMM under AZ_GEMM_MODE=<m> (one-pass search, 688 rows): FETCH_SIZE pass, SQ pass, kernel stats.
# usage (GPU box, repo root): bash az-net_amd/tools/pmc_mode.sh <mode> [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AZ_GEMM_MODE=${1:-3}
tag=${2:-pmcm}
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
args="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-level-loop --one-pass --no-extras --no-rccl --event-every 1000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $args > $out/bench.json 2> $out/kt.log
ks=$(find $out/kt -name '*kernel_stats.csv' | head -1)
python3 az-net_amd/tools/summarize_prof.py "$ks" $out/kernel_stats.csv "one pass, AZ_GEMM_MODE=$AZ_GEMM_MODE"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pf -- python3 $args > /dev/null 2> $out/logf
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/ps -- python3 $args > /dev/null 2> $out/logs
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out/pl -- python3 $args > /dev/null 2> $out/logl
python3 az-net_amd/tools/summarize_pmc.py $out/hbm.csv "$(find $out/pf -name '*counter_collection.csv' | head -1)"
python3 az-net_amd/tools/summarize_pmc.py $out/sq.csv "$(find $out/ps -name '*counter_collection.csv' | head -1)"
python3 az-net_amd/tools/summarize_pmc.py $out/lds.csv "$(find $out/pl -name '*counter_collection.csv' | head -1)"
rm -rf $out/kt $out/pf $out/ps $out/pl
grep -E "name|k_fc" $out/kernel_stats.csv; grep -E "kernel|k_fc" $out/hbm.csv $out/sq.csv $out/lds.csv
