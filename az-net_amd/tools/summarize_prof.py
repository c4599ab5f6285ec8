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
        # this library's kernels first (pct = share of THEIR total), then everything else the process ran (torch / MIOpen:
        # the backbone that makes the synthetic conv5_3 maps, incl. MIOpen's one-time benchmark search)
        def ours(r):
            n = short(r["Name"])
            return n.startswith("k_") or n.startswith("void k_")
        mine = [r for r in rows if ours(r)]
        tot = sum(float(r["TotalDurationNs"]) for r in mine) or 1.0
        o.write("name,calls,total_us,avg_us,pct,min_us,max_us\n")
        for r in mine:
            o.write("%s,%s,%.1f,%.2f,%.2f,%.2f,%.2f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                        float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot,
                                                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
        rest = [r for r in rows if not ours(r)]
        if rest:
            o.write("# other kernels of the process (PyTorch / MIOpen backbone that produces the synthetic maps, set-up only): %d names, %.1f ms\n"
                    % (len(rest), sum(float(r["TotalDurationNs"]) for r in rest) / 1e6))
            for r in rest[:12]:
                o.write("#   %s,%s,%.1f,%.2f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                  float(r["AverageNs"]) / 1e3))


if __name__ == "__main__":
    main()
