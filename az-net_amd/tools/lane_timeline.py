#!/usr/bin/env python3
"""Steady-state timeline from a rocprofv3 --kernel-trace CSV: every kernel of a few consecutive searches from the middle
of the trace with start, end, duration (us), one column per HIP queue (= lane).  Shows which kernels of the two lanes
overlap and which wait for the other lane's GEMM.
usage: lane_timeline.py kernel_trace.csv [searches=4] [kernel-name prefixes that end a search, comma separated]"""
import csv
import re
import sys


def main():
    rd = list(csv.DictReader(open(sys.argv[1])))
    nsearch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ends = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("k_final_select", "k_static_select")
    qk = next((k for k in ("Queue_Id", "Stream_Id") if rd and k in rd[0]), None)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:24], r.get(qk, "?") if qk else "?")
            for r in rd]
    rows.sort()
    # a search ends with its selection kernel (k_final_select; the one-pass plan: k_static_select)
    idx = [i for i, r in enumerate(rows) if r[2].startswith(ends)]
    if len(idx) < nsearch + 2:
        print("too few searches in the trace (%d)" % len(idx))
        return
    m = len(idx) // 2
    a, b = idx[m] + 1, idx[m + nsearch] + 1
    t0 = rows[a][0]
    qs = sorted({r[3] for r in rows[a:b]})
    print("# %d searches from the middle of the trace; columns = queues %s; times in us from the first kernel shown" % (nsearch, qs))
    print("# (a kernel's start is its dispatch: a single-workgroup kernel that shows ~1000 us next to the other lane's")
    print("#  k_fc_splitk12 waited for a CU that GEMM held -- all VGPRs of every SIMD -- and ran when it ended)")
    print("#   start       end       dur   kernel")
    for s, e, n, q in rows[a:b]:
        print("%9.2f %9.2f  %8.2f  %s%-24s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, " " * (26 * qs.index(q)), n))
    gem = [(s, e) for s, e, n, q in rows[a:b] if n.startswith("k_fc_splitk12")]
    if len(gem) >= 2:
        per = (gem[-1][0] - gem[0][0]) / 1e3 / (len(gem) - 1)
        dur = sum(e - s for s, e in gem) / 1e3 / len(gem)
        print("# k_fc_splitk12: one every %.1f us, %.1f us long: %.1f us per search in which no int6 GEMM runs" % (per, dur, per - dur))


if __name__ == "__main__":
    main()
