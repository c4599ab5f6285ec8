#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a
small CSV for profiles/ (kernel names shortened).  usage: summarize_prof.py in.csv out.csv "header note" """
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    src, dst, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as o:
        if note:
            o.write("# %s\n" % note)
        o.write("name,calls,total_us,avg_us,pct,min_us,max_us\n")
        for r in rows:
            o.write("%s,%s,%.1f,%.2f,%s,%.2f,%.2f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"],
                                                      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))


if __name__ == "__main__":
    main()
