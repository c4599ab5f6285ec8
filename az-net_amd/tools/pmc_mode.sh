#!/bin/bash
# Kernel stats and counters of the int6 GEMM under AZ_GEMM_MODE=<m> (one-pass search, 688 rows; level-loop stats too):
# separate rocprofv3 passes for the kernel trace, FETCH_SIZE, WRITE_SIZE and the SQ / LDS sets.
# usage (GPU box, repo root): bash az-net_amd/tools/pmc_mode.sh <mode> [tag]      -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AZ_GEMM_MODE=${1:-3}
tag=${2:-pmcm}
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
common="--no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-extras --no-rccl --event-every 1000"
one="bench.py --steps 30 --warmup 5 $common --no-level-loop --one-pass"
lvl="bench.py --steps 60 --warmup 5 $common --no-one-pass"
ours() { grep -E "^name|^kernel|^# |k_fc_|k_roi_pool|k_feat_scale|k_tail|k_spec|k_level|k_final" "$1"; }
for w in one lvl; do
  [ $w = one ] && args=$one || args=$lvl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $args > $out/bench_$w.json 2> $out/kt.log
  python3 az-net_amd/tools/summarize_prof.py "$(find $out/kt -name '*kernel_stats.csv' | head -1)" $out/k.csv "AZ_GEMM_MODE=$AZ_GEMM_MODE: python3 $args"
  ours $out/k.csv > $out/kernel_stats_$w.csv; rm -rf $out/kt $out/k.csv
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p -- python3 $one > /dev/null 2> $out/log$i
  python3 az-net_amd/tools/summarize_pmc.py $out/s.csv "$(find $out/p -name '*counter_collection.csv' | head -1)"
  ours $out/s.csv > $out/pmc_$i.csv; rm -rf $out/p $out/s.csv
  i=$((i+1))
done
rm -f $out/log* $out/kt.log
cat $out/kernel_stats_one.csv $out/pmc_*.csv
