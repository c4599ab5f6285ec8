#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one or more runs).
usage: summarize_pmc.py out.csv run1_counter_collection.csv [run2 ...]"""
import collections
import csv
import re
import sys


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))[:60]


def main():
    out = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sys.argv[2:]:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = sorted({c for k in agg for c in agg[k]})
    with open(out, "w") as o:
        o.write("kernel,launches,avg_us," + ",".join("avg_" + c for c in counters) + "\n")
        for k in sorted(agg, key=lambda k: -sum(dur[k])):
            if not (k.startswith("k_") or k.startswith("void k_")):
                continue
            n = max(len(v) for v in agg[k].values())
            o.write("%s,%d,%.2f," % (k, n, sum(dur[k]) / len(dur[k])) +
                    ",".join("%.6g" % (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else "" for c in counters) + "\n")


if __name__ == "__main__":
    main()
