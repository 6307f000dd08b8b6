#!/usr/bin/env python3
"""Per-kernel summary of tools/prof_quick.sh output: average duration and counters per launch."""
import collections
import csv
import glob
import sys

o = sys.argv[1]


def rows(pat):
    for f in glob.glob(pat, recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


dur = collections.defaultdict(list)
for r in rows(o + "/stats/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("sqa", "sqb"):
    for r in rows(o + "/" + d + "/**/*counter_collection.csv"):
        k = r["Kernel_Name"].split("(")[0][:60]
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        nl[k][r["Counter_Name"]] += 1
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 50:
        continue
    skip = 4 if len(v) > 8 else 0  # warm-up launches
    vv = v[skip:]
    print("%-62s n=%3d avg %.1f us" % (k, len(v), sum(vv) / len(vv)))
    c = {n: cnt[k][n] / max(nl[k][n], 1) for n in cnt[k]}
    if c:
        print("    " + "  ".join("%s=%.3g" % (n.replace("SQ_", ""), x) for n, x in sorted(c.items())))
