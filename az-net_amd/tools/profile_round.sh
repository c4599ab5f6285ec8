#!/bin/bash
# One profiling pass of bench.py for profiles/: kernel stats, the kernel trace (gaps), and the three
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM), each summarised with the tools beside this file.
# usage (on the GPU box, from the repo root):  bash az-net_amd/tools/profile_round.sh <tag>
# writes gpurun_out/<tag>/{kernel_stats.csv,gaps.txt,pmc_hbm.csv,pmc_sq.csv,fc_*_by_launch.csv,bench_prof.json}
set -u
tag=${1:-prof}
period=${2:-2}      # k_fc_splitk launches per image: 2 = all levels in one head pass (Tz <= 0), 6 = level loop
EXTRA=${3:-}       # e.g. "--level-loop" (with period 6)
repo=$(pwd)
out=$repo/gpurun_out/$tag
tools=$repo/az-net_amd/tools
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
args="bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-level-loop $EXTRA"

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 $args > "$out/bench_prof.json" 2> "$out/kt.log"
ks=$(find "$out/kt" -name '*kernel_stats.csv' | head -1)
kt=$(find "$out/kt" -name '*kernel_trace.csv' | head -1)
python3 "$tools/summarize_prof.py" "$ks" "$out/kernel_stats.csv" "rocprofv3 --kernel-trace --stats -- python3 $args"
python3 "$tools/trace_gaps.py" "$kt" > "$out/gaps.txt" 2>&1

for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_$ctr" -- python3 $args > /dev/null 2> "$out/pmc_$ctr.log"
done
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_SQ" -- python3 $args > /dev/null 2> "$out/pmc_SQ.log"

cf=$(find "$out/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
cw=$(find "$out/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
cs=$(find "$out/pmc_SQ" -name '*counter_collection.csv' | head -1)
python3 "$tools/summarize_pmc.py" "$out/pmc_hbm.csv" "$cf" "$cw"
python3 "$tools/summarize_pmc.py" "$out/pmc_sq.csv" "$cs"
python3 "$tools/pmc_by_launch.py" "$cf" k_fc_splitk $period > "$out/fc_fetch_by_launch.csv"
python3 "$tools/pmc_by_launch.py" "$cw" k_fc_splitk $period > "$out/fc_write_by_launch.csv"
python3 "$tools/pmc_by_launch.py" "$cs" k_fc_splitk $period > "$out/fc_sq_by_launch.csv"
# raw traces are large: keep only the summaries
rm -rf "$out/kt" "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/pmc_SQ"
ls -la "$out"
