#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (previous kernel ->
next kernel): where the per-image time outside kernels goes.  usage: trace_gaps.py kernel_trace.csv [first_kernel]"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))
    return n.split("<")[0][-40:]


def main():
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort()
    first = sys.argv[2] if len(sys.argv) > 2 else "k_rank_count"      # once per search (final selection)
    # keep the steady-state tail: last 60 % of the trace
    rows = rows[int(len(rows) * 0.4):]
    gaps = collections.defaultdict(list)
    busy = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gaps[(n0, n1)].append((s1 - e0) / 1e3)
        busy[n0].append((e0 - s0) / 1e3)
    steps = sum(1 for r in rows if r[2] == first)
    print("steps in window: %d" % steps)
    tot = 0.0
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        per = sum(v) / max(steps, 1)
        tot += per
        if per > 0.3:
            print("%-34s -> %-34s n=%4d avg %7.2f us  per-step %7.2f us" % (k[0], k[1], len(v), sum(v) / len(v), per))
    print("total gap per step: %.1f us" % tot)
    print("total busy per step: %.1f us" % (sum(sum(v) for v in busy.values()) / max(steps, 1)))


if __name__ == "__main__":
    main()
