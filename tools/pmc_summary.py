"""Aggregate rocprofv3 --pmc counter CSVs per kernel (mean per dispatch)."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("ba::", "").replace("anonymous namespace)::", "")
            name = r["Kernel_Name"]
            name = name[name.find("k_"):].split("(")[0] if "k_" in name else name[:40]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-22s %-28s %8s %16s" % ("kernel", "counter", "calls", "mean/dispatch"))
for k in sorted(rows):
    for c in sorted(rows[k]):
        v = rows[k][c]
        print("%-22s %-28s %8d %16.1f" % (k, c, len(v), sum(v) / len(v)))
