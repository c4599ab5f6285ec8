import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "nms" in n:
        print("   %-22s %8.1f us" % (n.split("::")[-1].split("(")[0], float(r["AverageNs"]) / 1e3))
