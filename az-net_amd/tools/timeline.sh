#!/bin/bash
# Timeline of one image's kernels (start / duration / gap to the previous kernel, us) from a rocprofv3 kernel trace.
# usage (GPU box, repo root): bash az-net_amd/tools/timeline.sh [bench flags]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/timeline; rm -rf $out; mkdir -p $out
args="bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-one-pass --no-extras --no-rccl --event-every 1000 $*"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 $args > $out/bench.json 2> $out/kt.log
kt=$(find $out/kt -name '*kernel_trace.csv' | head -1)
python3 - "$kt" <<'PY'
import csv, re, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:28]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# an image from the timed loop (a third of the way through the trace: behind it the bench measures other forms)
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_final_select")]
m = len(idx) // 3
a, b = idx[m] + 1, idx[m + 1] + 1
t0 = rows[a][0]
prev = rows[a - 1][1]
for s, e, n in rows[a:b + 2]:
    print("%-28s start %9.2f  dur %8.2f  gap %7.2f" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
PY
rm -rf $out/kt
