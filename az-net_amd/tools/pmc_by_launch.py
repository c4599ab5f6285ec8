#!/usr/bin/env python3
"""Per launch-shape averages of rocprofv3 --pmc counters for one kernel whose launches repeat with a fixed
period (k_fc_splitk: six launches per image: int6 / int7 of the speculative pass, level 4, level 5).
usage: pmc_by_launch.py counter_collection.csv kernel_substring period"""
import collections
import csv
import sys


def main():
    f, name, period = sys.argv[1], sys.argv[2], int(sys.argv[3])
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if name not in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    ids = ids[len(ids) % period:]                      # drop leading partial period (head_forward warm-ups)
    agg = [collections.defaultdict(list) for _ in range(period)]
    for i, k in enumerate(ids):
        for c, v in disp[k].items():
            agg[i % period][c].append(v)
    cols = sorted({c for a in agg for c in a})
    print("slot," + ",".join(cols))
    for i, a in enumerate(agg):
        print("%d," % i + ",".join("%.6g" % (sum(a[c]) / len(a[c])) if a[c] else "" for c in cols))


if __name__ == "__main__":
    main()
