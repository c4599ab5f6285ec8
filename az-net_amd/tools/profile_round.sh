#!/bin/bash
# One profiling pass of bench.py for profiles/: kernel stats and the three separate --pmc passes (FETCH_SIZE, WRITE_SIZE,
# SQ/GRBM), each summarised with the tools beside this file, for three workloads:
#   main     the search at Tz = 0 (bench.py's `value`: whole-tree pass, one lane, three searches queued) -> <tag>_*
#   lanes2   the same on two lanes (az_set_lanes(ctx, 2))                        -> <tag>_lanes2_*
#   twopass  the same with AZ_FULL_SPEC=0 (48-row pass + 670-row pass)          -> <tag>_twopass_*
#   onepass  the Tz <= 0 one-pass form (`one_pass`)                             -> <tag>_onepass_*
#   stream   bench.py's stream_tz leg (distinct images at a tuned threshold)          -> <tag>_stream_*
#   extras   calibrated Tz, deep tree (config 4), shared detection (config 3), az_nms at 100 / 300 / 2000 / 8129 boxes:
#            the geometry / NMS / detection kernels (k_nms_*, k_divide, k_dedup_*, k_level_geom, k_spec_levels, ...)
#                                                                               -> <tag>_extras_*
# usage (on the GPU box, from the repo root):  bash az-net_amd/tools/profile_round.sh <tag> [main|onepass|extras ...]
# writes gpurun_out/<tag>/...; raw traces are deleted, only the summaries stay.
set -u
tag=${1:-prof}
shift || true
sets=${*:-main twopass onepass extras}
repo=$(pwd)
out=$repo/gpurun_out/$tag
tools=$repo/az-net_amd/tools
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$repo"
# the context's one-time measurement of head-pass costs (24 head passes at 48 .. 1408 rows, first launch) is kept out of the
# per-kernel averages: with AZ_PASS_CAL=0 the form choice goes by the built-in figures (same forms at these workloads)
export AZ_PASS_CAL=0
# (bench.py runs its side legs only with --extras; the sets that profile one of them switch the others off)
common="--no-cpu-baseline --no-box"
legs_off="--extras --no-e2e --no-fast --no-sweep --no-one-lane"

run_set() {
  name=$1; args=$2; pfx=$3; period=$4
  d=$out/$name
  mkdir -p "$d"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/kt" -- python3 bench.py $args > "$d/bench_prof.json" 2> "$d/kt.log"
  ks=$(find "$d/kt" -name '*kernel_stats.csv' | head -1)
  kt=$(find "$d/kt" -name '*kernel_trace.csv' | head -1)
  python3 "$tools/summarize_prof.py" "$ks" "$out/${pfx}kernel_stats.csv" "rocprofv3 --kernel-trace --stats -- python3 bench.py $args"
  python3 "$tools/lane_timeline.py" "$kt" > "$out/${pfx}timeline.txt" 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$d/pmc_$ctr" -- python3 bench.py $args > /dev/null 2> "$d/pmc_$ctr.log"
  done
  timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$d/pmc_SQ" -- python3 bench.py $args > /dev/null 2> "$d/pmc_SQ.log"
  cf=$(find "$d/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
  cw=$(find "$d/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
  cs=$(find "$d/pmc_SQ" -name '*counter_collection.csv' | head -1)
  python3 "$tools/summarize_pmc.py" "$out/${pfx}pmc_hbm.csv" "$cf" "$cw"
  python3 "$tools/summarize_pmc.py" "$out/${pfx}pmc_sq.csv" "$cs"
  if [ "$period" != "0" ]; then
    python3 "$tools/pmc_by_launch.py" "$cf" k_fc_splitk $period > "$out/${pfx}fc_fetch_by_launch.csv"
    python3 "$tools/pmc_by_launch.py" "$cw" k_fc_splitk $period > "$out/${pfx}fc_write_by_launch.csv"
    python3 "$tools/pmc_by_launch.py" "$cs" k_fc_splitk $period > "$out/${pfx}fc_sq_by_launch.csv"
  fi
  cp "$d/bench_prof.json" "$out/${pfx}bench_under_profiler.json"
  rm -rf "$d"
}

for s in $sets; do
  case $s in
    # kernels whose name contains k_fc_splitk per image: the whole-tree pass's int6 (k_fc_splitk12, 688 rows) and int7 = 2
    # (the search's first, history-less image takes the two-pass form: one stray pair of launches in the averages)
    main)    run_set main "--steps 100 --warmup 10 $common --no-rccl --event-every 1000" "" 2 ;;
    # the same loop on TWO lanes (az_set_lanes(2)): round 4-5's `value`
    lanes2)  run_set lanes2 "--steps 100 --warmup 10 $common --lanes 2 --no-rccl --event-every 1000" "lanes2_" 2 ;;
    # the level loop without the whole-tree pass (AZ_FULL_SPEC=0): 48-row pass (k_fc_splitk int6, int7), 670-row pass
    # (k_fc_splitk12 int6, int7) = 4
    twopass) AZ_FULL_SPEC=0 run_set twopass "--steps 100 --warmup 10 $common --no-rccl --event-every 1000" "twopass_" 4 ;;
    # one pass: k_fc_splitk12 (int6, 688 rows), k_fc_splitk (int7) = 2
    onepass) run_set onepass "--steps 100 --warmup 10 $common --no-rccl --one-pass --event-every 1000" "onepass_" 2 ;;
    extras)  run_set extras "--steps 10 --warmup 2 $common $legs_off --no-stream --no-one-pass --no-rccl --event-every 1000" "extras_" 0 ;;
    # stream_tz: distinct images in dataset order at a threshold tuned over the set (the small, weight-streaming-bound passes
    # of pruned trees: k_fc_splitk at 9 .. ~200 rows, the single-workgroup geometry kernels, the early ends)
    stream)  run_set stream "--steps 10 --warmup 2 $common $legs_off --lanes 2 --no-calibrated --no-one-pass --no-two-pass --no-extras --no-rccl --event-every 1000" "stream_" 0 ;;
  esac
done
ls -la "$out"
