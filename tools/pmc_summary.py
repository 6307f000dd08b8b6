#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections
import csv
import glob
import sys

for pat in sys.argv[1:]:
    for f in sorted(glob.glob(pat)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:26]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, v in agg.items():
            print("%-26s" % k, " ".join("%s=%d" % (c.replace("SQ_", ""), round(sum(x) / len(x))) for c, x in sorted(v.items())))
