#!/bin/bash
set -eu
# How busy the matrix pipe is during int6 at a few row counts (1 .. 6 strips, then many): rocprofv3 --pmc GRBM_GUI_ACTIVE
# SQ_VALU_MFMA_BUSY_CYCLES over az-net_amd/tools/perf_head.py, summarised per row count (the int6 launch of every head forward).
# usage (GPU box, repo root): bash az-net_amd/tools/mfma_busy_by_rows.sh <tag> [rows,rows,...]  -> gpurun_out/<tag>/mfma_busy_by_rows.txt
tag=${1:-mfma}; rows=${2:-32,64,96,128,160,192,256,384,688}
repo=${GRAFT_REPO_ROOT:?run on the GPU box (GRAFT_REPO_ROOT is set there)}
cd /tmp && export TMPDIR=/tmp && cd "$repo"
out="$repo/gpurun_out/$tag"; rm -rf "$out"; mkdir -p "$out"
reps=6
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out"/pmc -- python3 az-net_amd/tools/perf_head.py "$rows" $reps > "$out"/perf_head.txt 2> "$out"/pmc.log
cc=$(find "$out"/pmc -name '*counter_collection.csv' | head -1)
python3 - "$cc" "$rows" $reps > "$out"/mfma_busy_by_rows.txt <<'PY'
import collections, csv, sys
f, rows, reps = sys.argv[1], [int(x) for x in sys.argv[2].split(",")], int(sys.argv[3])
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_fc_splitk" not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "k": r["Kernel_Name"][:60]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
# per row count: (1 + reps) head forwards, each an int6 launch then an int7 launch
per = 2 * (1 + reps)
print("# int6 launches of az_head_forward (perf_head.py), counters summed over the chip's SIMDs / shader engines as rocprofv3 reports them")
print("# clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): as profiles/README.md")
print("rows,kernel,launch_us_under_counters,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,clock_GHz,mfma_busy_frac")
for i, R in enumerate(rows):
    chunk = ids[i * per:(i + 1) * per]
    six = [disp[k] for k in chunk[2::2]]           # the int6 launches after the warm-up forward
    if not six:
        continue
    us = sum(d["us"] for d in six) / len(six)
    ga = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in six) / len(six)
    mb = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in six) / len(six)
    print("%d,%s,%.1f,%.4g,%.4g,%.3f,%.3f" % (R, six[0]["k"].replace("(anonymous namespace)::", "").split("(")[0], us, ga, mb,
                                           ga / 8.0 / us / 1e3 if us else 0.0, mb / (1024.0 * ga / 8.0) if ga else 0.0))
PY
rm -rf "$out"/pmc
cat "$out"/mfma_busy_by_rows.txt
